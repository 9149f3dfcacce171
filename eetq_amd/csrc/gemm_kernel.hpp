// Kernel template of the LDS-tiled MFMA dequant-GEMM (included by gemm.hip and tools/kbench.hip).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace eetq {
namespace gemm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_STAGE_BYTES = BM * BK * 2;  // 16 KiB of fp16 activations per K step
constexpr int B_STAGE_BYTES = BN * BK;      // 8 KiB of uint8 weights (8 native tiles) per K step
constexpr int STAGE_BYTES   = A_STAGE_BYTES + B_STAGE_BYTES;
constexpr int STAGES        = 6;
constexpr int SMEM_BYTES    = STAGES * STAGE_BYTES;  // 144 KiB (also covers the 64 KiB end-of-kernel reduction)
constexpr int kMinKSteps    = STAGES - 1;            // the statically unrolled drain needs K/64 >= 5

typedef __attribute__((address_space(3))) void lds_void;

// LDS-DMA: 16 B per lane from a buffer (base in the descriptor, per-lane byte offset in voff, wave-uniform byte
// offset in soff) straight into LDS at wave-uniform base + lane*16, no VGPR round trip.  The MUBUF form
// (buffer_load_dwordx4 ... lds) is used rather than global_load_lds: hipcc treats the latter as a FLAT access
// that may touch LDS and from then on degrades every counted "s_waitcnt lgkmcnt(N)" to lgkmcnt(0) while any DMA
// is in flight (SIInsertWaitcnts "pending flat"), which serialises the LDS fragment reads behind the MFMAs.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, uint8_t* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ f16x8 make_frag(f16x2 a, f16x2 b, f16x2 c, f16x2 d)
{
    return f16x8{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

// Workgroup tile 128 x 128 x 64.  The two 32-deep halves of every K step go to two wave groups; a wave owns 128 rows x
// (32*J) columns of one K half.
//   J = 1: 8 waves (2 groups x 4 column slices of 32), two waves per SIMD.
//   J = 2: 4 waves (2 groups x 2 column slices of 64), one wave per SIMD, 128 fp32 accumulators per lane: every
//          activation fragment read from LDS feeds two MFMAs, so the LDS read traffic of the A operand halves
//          (72 -> 40 KiB per K step).  LDS reads + DMA writes were ~63 % of the LDS peak with J = 1.
// The two groups' fp32 partial sums are added once at the end through LDS.  6-stage DMA ring, one barrier per K step.
// ABLATE (kbench only): 1 = no DMA in the loop, 2 = no dequant, 4 = no LDS fragment reads, 8 = no MFMA, 16 = no barrier
template <int ABLATE, int J>
__global__ __launch_bounds__(512 / J, J == 1 ? 2 : 1) void gemm_tile_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales, const f16* __restrict__ bias,
    f16* __restrict__ y, int M, int N, int K)
{
    constexpr int NWAVES  = 8 / J;            // 2 K groups x (4 / J) column slices
    constexpr int WN_COLS = 32 * J;           // columns per wave
    constexpr int PIECES  = 24 / NWAVES;      // 16 A + 8 B LDS-DMA pieces of 1 KiB per stage, split over the waves
    constexpr int NMFMA   = 8 * J;            // MFMAs per wave per K step
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp  = wave / (NWAVES / 2);  // which 32-deep half of each K step
    const int wn   = wave % (NWAVES / 2);  // which column slice of the tile
    const int KT   = K >> 6;

    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = (N + BN - 1) / BN;
    const int T       = tiles_m * tiles_n;
    int       tile;
    {
        const int b = blockIdx.x, q = T >> 3, r = T & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile % tiles_m) * BM;
    const int n0 = (tile / tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(w), 0, (int)((size_t)N * K), 0x00020000);
    // DMA pieces of this wave: global piece p = wave*PIECES + i; p < 16: A rows 8p..8p+7 (128 B each), else B tile p-16
    int  dma_voff[PIECES];
    const int n_tiles_total = N >> 4;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int p = wave * PIECES + i;
        if (p < 16) {
            const int row  = p * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            int       gm   = m0 + row;
            gm             = gm < M ? gm : M - 1;
            dma_voff[i]    = (gm * K + slot * 8) * 2;
        } else {
            int nt      = (n0 >> 4) + (p - 16);
            nt          = nt < n_tiles_total ? nt : n_tiles_total - 1;
            dma_voff[i] = nt * KT * kTileBytes + lane * 16;
        }
    }
    auto dma_piece = [&](int i, int stage, int kt) {
        const int p  = wave * PIECES + i;
        uint8_t*  sa = smem + stage * STAGE_BYTES;
        if (p < 16)
            dma16(x_rsrc, dma_voff[i], kt * BK * 2, sa + p * 1024);
        else
            dma16(w_rsrc, dma_voff[i], kt * kTileBytes, sa + A_STAGE_BYTES + (p - 16) * 1024);
    };

    const int fn = lane & 31, fh = lane >> 5;
    // weights of column block j: tile column wn*WN_COLS + 32j + fn -> chunk (col>>4), slot (2*grp+fh)*16 + (col&15)
    int b_off[J];
#pragma unroll
    for (int j = 0; j < J; ++j)
        b_off[j] = A_STAGE_BYTES + ((wn * WN_COLS + 32 * j + fn) >> 4) * 1024 + (fn & 15) * 16 + fh * 256 + grp * 512;
    const int a_key     = (fn >> 1) & 7;
    const int a_row_off = fn * 128;
    const int a_slot0   = ((4 * grp + 2 * fh + 0) ^ a_key) << 4;
    const int a_slot1   = ((4 * grp + 2 * fh + 1) ^ a_key) << 4;

    f16x2 scale2[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int ncol = n0 + wn * WN_COLS + 32 * j + fn;
        const f16 sc   = scales[ncol < N ? ncol : N - 1];
        scale2[j]      = f16x2{sc, sc};
    }

    f32x16 acc[4][J];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;

    struct Frags {
        u32x4 wq[J];
        f16x8 xa[2][4];
    };
    struct WFrag {
        f16x8 f[J][2];
    };

    // One K step of one wave: NMFMA MFMAs on (wcur, fcur); in their shadow the LDS reads of the next K step's fragments
    // (READ), the dequant of the freshly read weights, and this wave's DMA pieces of stage kt+STAGES-1 (DMA).
    // Issue order (sched_barrier-pinned): slot i = MFMA i, then
    //   i = 0: weight reads;  i < 4: two activation reads;  i < PIECES: one DMA piece;  4 <= i < 4+4J: one dword of dequant.
    auto step = [&](const WFrag& wcur, const Frags& fcur, auto read_tag, int nstage, Frags& fnext, WFrag& wnext,
                    auto dma_tag, int dma_stage, int dma_kt) {
        constexpr bool READ = decltype(read_tag)::value;
        constexpr bool DMA  = decltype(dma_tag)::value;
        const uint8_t* sa   = smem + nstage * STAGE_BYTES;
        f16x2          wd[J][8];
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
            const int e = i / (4 * J), j = (i / 4) % J, mt = i & 3;
            if constexpr (!(ABLATE & 8))
                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur.f[j][e], fcur.xa[e][mt], acc[mt][j], 0, 0, 0);
            else
                asm volatile("" ::"v"(wcur.f[j][e]), "v"(fcur.xa[e][mt]));
            if constexpr (READ && !(ABLATE & 4)) {
                if (i == 0) {
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) fnext.wq[jj] = *reinterpret_cast<const u32x4*>(sa + b_off[jj]);
                }
                if (i < 4) {
                    const int ne = i >> 1;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int nmt     = 2 * (i & 1) + t;
                        fnext.xa[ne][nmt] = *reinterpret_cast<const f16x8*>(sa + nmt * 32 * 128 + a_row_off +
                                                                            (ne ? a_slot1 : a_slot0));
                    }
                } else if (i < 4 + 4 * J) {
                    if constexpr (!(ABLATE & 2)) {
                        const int jj = (i - 4) >> 2, d = (i - 4) & 3;
                        const u32 wdw = d == 0   ? fnext.wq[jj].x
                                        : d == 1 ? fnext.wq[jj].y
                                        : d == 2 ? fnext.wq[jj].z
                                                 : fnext.wq[jj].w;
                        dequant_dword(wdw, scale2[jj], wd[jj][2 * d], wd[jj][2 * d + 1]);
                        // pure VALU ops float freely through instruction selection; the empty asm pins them to this slot
                        asm volatile("" : "+v"(wd[jj][2 * d]), "+v"(wd[jj][2 * d + 1]));
                    }
                }
            }
            if constexpr (DMA && !(ABLATE & 1)) {
                if (i < PIECES) dma_piece(i, dma_stage, dma_kt);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (READ && !(ABLATE & 4)) {
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                if constexpr (!(ABLATE & 2)) {
                    wnext.f[jj][0] = make_frag(wd[jj][0], wd[jj][1], wd[jj][2], wd[jj][3]);
                    wnext.f[jj][1] = make_frag(wd[jj][4], wd[jj][5], wd[jj][6], wd[jj][7]);
                } else {
                    wnext.f[jj][0] = __builtin_bit_cast(f16x8, fnext.wq[jj]);
                    wnext.f[jj][1] = __builtin_bit_cast(f16x8, fnext.wq[jj]);
                }
            }
            // the activation reads above are >= 4 MFMAs (128+ cycles) old: this wait is free, and it hands hipcc's
            // wait-count pass an empty LDS queue at every step boundary (keeps its counted lgkmcnt waits exact)
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt/expcnt untouched
        } else if constexpr (READ) {
            wnext = wcur;
            fnext = fcur;
        }
    };

    // ---- prologue: STAGES-1 stages in flight; stage 0 -> fragments ----
    asm volatile("" ::"v"(scale2[0]));  // the (tiny) scale loads retire before LDS-DMA is queued behind them
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {  // KT >= STAGES - 1 by launch contract
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dma_piece(i, s, s);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PIECES) : "memory");  // stage 0 landed
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    WFrag w0, w1;
    {
#pragma unroll
        for (int j = 0; j < J; ++j) f0.wq[j] = *reinterpret_cast<const u32x4*>(smem + b_off[j]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                f0.xa[e][mt] = *reinterpret_cast<const f16x8*>(smem + mt * 32 * 128 + a_row_off + (e ? a_slot1 : a_slot0));
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f16x2 wd[8];
            dequant_16(f0.wq[j], scale2[j], wd);
            w0.f[j][0] = make_frag(wd[0], wd[1], wd[2], wd[3]);
            w0.f[j][1] = make_frag(wd[4], wd[5], wd[6], wd[7]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }

    // REM = K steps after this one, clamped to STAGES-1.  REM >= STAGES-1: steady state (DMA for stage kt+STAGES-1).
    int  stage = 0;
    auto k_step = [&](int kt, auto rem_tag, const WFrag& wcur, const Frags& fcur, WFrag& wnext, Frags& fnext) {
        constexpr int REM  = decltype(rem_tag)::value;
        const int     next = stage + 1 == STAGES ? 0 : stage + 1;
        if constexpr (REM >= 1) {
            // stage kt+1 must have landed in LDS (all waves' pieces); younger stages stay in flight
            constexpr int younger = (REM - 1) < (STAGES - 3) ? (REM - 1) : (STAGES - 3);
            if constexpr (!(ABLATE & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger * PIECES) : "memory");
            if constexpr (!(ABLATE & 16)) __builtin_amdgcn_s_barrier();
            int dst = stage + STAGES - 1;
            dst     = dst >= STAGES ? dst - STAGES : dst;
            step(wcur, fcur, std::true_type{}, next, fnext, wnext, std::integral_constant<bool, (REM >= STAGES - 1)>{},
                 dst, kt + STAGES - 1);
        } else {
            step(wcur, fcur, std::false_type{}, 0, fnext, wnext, std::false_type{}, 0, 0);
        }
        stage = next;
    };
    using Steady = std::integral_constant<int, STAGES - 1>;
    // Main loop: two K steps per iteration so the two fragment sets alternate without register copies.  It stops
    // 6 (KT even) or 5 (KT odd) steps before the end, so the drain below is fully static: no run-time choice of
    // fragment set or of REM (both would push the fragment registers through scratch).  Needs KT >= 5.
    const int tail = (KT & 1) ? 5 : 6;
    int       kt   = 0;
    for (; kt < KT - tail; kt += 2) {
        k_step(kt, Steady{}, w0, f0, w1, f1);
        k_step(kt + 1, Steady{}, w1, f1, w0, f0);
    }
    if (tail == 6) {
        k_step(kt, std::integral_constant<int, 5>{}, w0, f0, w1, f1);
        k_step(kt + 1, std::integral_constant<int, 4>{}, w1, f1, w0, f0);
        k_step(kt + 2, std::integral_constant<int, 3>{}, w0, f0, w1, f1);
        k_step(kt + 3, std::integral_constant<int, 2>{}, w1, f1, w0, f0);
        k_step(kt + 4, std::integral_constant<int, 1>{}, w0, f0, w1, f1);
        k_step(kt + 5, std::integral_constant<int, 0>{}, w1, f1, w0, f0);
    } else {
        k_step(kt, std::integral_constant<int, 4>{}, w0, f0, w1, f1);
        k_step(kt + 1, std::integral_constant<int, 3>{}, w1, f1, w0, f0);
        k_step(kt + 2, std::integral_constant<int, 2>{}, w0, f0, w1, f1);
        k_step(kt + 3, std::integral_constant<int, 1>{}, w1, f1, w0, f0);
        k_step(kt + 4, std::integral_constant<int, 0>{}, w0, f0, w1, f1);
    }

    // ---- combine the two K halves: group 1 parks its accumulators in LDS, group 0 adds and stores ----
    __builtin_amdgcn_s_barrier();  // every wave is done with the stage ring
    float* red = reinterpret_cast<float*>(smem) + (size_t)wn * (64 * J) * 64;  // [reg][lane]
    if (grp == 1) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((mt * J + j) * 16 + r) * 64 + lane] = acc[mt][j][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + mt * 32 + fn;
#pragma unroll
            for (int j = 0; j < J; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][j][r] += red[((mt * J + j) * 16 + r) * 64 + lane];
                // acc[mt][j][r] = y[m0 + 32*mt + fn][n0 + wn*WN_COLS + 32*j + 8*(r>>2) + 4*fh + (r&3)]
                const int nbase = n0 + wn * WN_COLS + 32 * j + 4 * fh;
                if (m < M) {
                    f16* yrow = y + (size_t)m * N + nbase;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (nbase + 8 * q < N) {  // N % 16 == 0: a group of 4 columns is all-in or all-out
                            f16x2 lo = {(f16)acc[mt][j][4 * q + 0], (f16)acc[mt][j][4 * q + 1]};
                            f16x2 hi = {(f16)acc[mt][j][4 * q + 2], (f16)acc[mt][j][4 * q + 3]};
                            if (bias) {  // fp16 add after the fp16 rounding: bit-identical to a separate `+ bias`
                                const u32x2 b = *reinterpret_cast<const u32x2*>(bias + nbase + 8 * q);
                                lo            = lo + as_f16x2(b.x);
                                hi            = hi + as_f16x2(b.y);
                            }
                            *reinterpret_cast<u32x2*>(yrow + 8 * q) = u32x2{as_u32(lo), as_u32(hi)};
                        }
                    }
                }
            }
        }
    }
}

}  // namespace gemm
}  // namespace eetq
