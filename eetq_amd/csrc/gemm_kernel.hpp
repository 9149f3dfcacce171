// Kernel template of the LDS-tiled MFMA dequant-GEMM (included by gemm.hip and tools/kbench.hip).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace eetq {
namespace gemm {

constexpr int BM = 128, BN = 128, BK = 64, STAGES = 4, THREADS = 256;
constexpr int A_STAGE_BYTES = BM * BK * 2;  // 16 KiB
constexpr int B_STAGE_BYTES = BN * BK;      // 8 KiB
constexpr int STAGE_BYTES   = A_STAGE_BYTES + B_STAGE_BYTES;
constexpr int SMEM_BYTES    = STAGES * STAGE_BYTES;  // 96 KiB
constexpr int A_GLDS_PER_WAVE = A_STAGE_BYTES / 1024 / 4;  // 4
constexpr int B_GLDS_PER_WAVE = B_STAGE_BYTES / 1024 / 4;  // 2
constexpr int GLDS_PER_STAGE  = A_GLDS_PER_WAVE + B_GLDS_PER_WAVE;

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

template <int SCHED>
__global__ __launch_bounds__(THREADS) void gemm_mfma_kernel(const f16* __restrict__ x, const uint8_t* __restrict__ w,
                                                            const f16* __restrict__ scales, const f16* __restrict__ bias,
                                                            f16* __restrict__ y, int M, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int KT   = K >> 6;

    // ---- XCD-aware tile assignment (bijective for any tile count) ----
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

    // ---- buffer descriptors + per-lane byte offsets for the LDS-DMA stage copies ----
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(w), 0, (int)((size_t)N * K), 0x00020000);
    int a_voff[A_GLDS_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_GLDS_PER_WAVE; ++i) {
        const int row  = (wave * A_GLDS_PER_WAVE + i) * 8 + (lane >> 3);  // 8 rows x 128 B per instruction
        const int slot = (lane & 7) ^ ((row >> 1) & 7);                   // source slot for this LDS slot
        int       gm   = m0 + row;
        gm             = gm < M ? gm : M - 1;  // rows past M: read a valid row, results are never stored
        a_voff[i]      = (gm * K + slot * 8) * 2;
    }
    int       b_voff[B_GLDS_PER_WAVE];
    const int n_tiles_total = N >> 4;
#pragma unroll
    for (int i = 0; i < B_GLDS_PER_WAVE; ++i) {
        int nt    = (n0 >> 4) + wave * B_GLDS_PER_WAVE + i;
        nt        = nt < n_tiles_total ? nt : n_tiles_total - 1;
        b_voff[i] = nt * KT * kTileBytes + lane * 16;
    }

    auto issue_stage = [&](int stage, int kt) {
        uint8_t* sa = smem + stage * STAGE_BYTES;
        uint8_t* sb = sa + A_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_GLDS_PER_WAVE; ++i)
            dma16(x_rsrc, a_voff[i], kt * BK * 2, sa + (wave * A_GLDS_PER_WAVE + i) * 1024);
#pragma unroll
        for (int i = 0; i < B_GLDS_PER_WAVE; ++i)
            dma16(w_rsrc, b_voff[i], kt * kTileBytes, sb + (wave * B_GLDS_PER_WAVE + i) * 1024);
    };

    // ---- per-lane fragment addressing ----
    // Operand roles are swapped w.r.t. the textbook C = A*B: the dequantised weights are the MFMA "A" operand
    // (row i = output column n) and the activations the "B" operand (column j = token m), so a lane's 4
    // consecutive accumulator registers are 4 consecutive n of one token: 8-byte fp16 stores, no transpose.
    const int fn = lane & 31, fh = lane >> 5;
    // weights: column 32*wave + fn -> chunk 2*wave + (fn>>4), 16-B slot g*16 + (fn&15), g = 2s + fh
    const int b_off = (wave * 2 + (fn >> 4)) * 1024 + (fn & 15) * 16 + fh * 256;  // + s*512
    // activations: row 32*mt + fn, slot (4s + 2fh + e) ^ key(row); key depends on fn only (32*mt/2 = 0 mod 8)
    const int a_key     = (fn >> 1) & 7;
    const int a_row_off = fn * 128;

    // scale of this lane's weight column (clamped for a ragged last tile)
    const int   ncol_c = (n0 + wave * 32 + fn) < N ? (n0 + wave * 32 + fn) : N - 1;
    const f16   sc     = scales[ncol_c];
    const f16x2 scale2 = {sc, sc};

    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    struct Frags {
        u32x4 wq;        // 16 k of one weight column (raw uint8)
        f16x8 xa[2][4];  // activations [e][mt]
    };
    auto load_frags = [&](int stage, int s, Frags& f) {
        const uint8_t* sa = smem + stage * STAGE_BYTES;
        const uint8_t* sb = sa + A_STAGE_BYTES;
        f.wq              = *reinterpret_cast<const u32x4*>(sb + b_off + s * 512);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int slot = ((4 * s + 2 * fh + e) ^ a_key) << 4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                f.xa[e][mt] = *reinterpret_cast<const f16x8*>(sa + mt * 32 * 128 + a_row_off + slot);
        }
    };
    typedef f16x8 WFrag[2];
    auto dequant_frags = [&](const u32x4& wq, WFrag& wf) {
        f16x2 wd[8];
        dequant_16(wq, scale2, wd);
        wf[0] = make_frag(wd[0], wd[1], wd[2], wd[3]);
        wf[1] = make_frag(wd[4], wd[5], wd[6], wd[7]);
    };
    auto mma_half = [&](const WFrag& wf, const Frags& f) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[e], f.xa[e][mt], acc[mt], 0, 0, 0);
    };
    // Issue-order hint for one half K step (8 MFMA, 9 LDS fragment reads of the *next* half, its 24-op dequant):
    // reads ride behind the first MFMAs, the dequant of the freshly read weights behind the last ones, so the
    // matrix pipe never waits for VALU or LDS (one wave per SIMD: nothing else would hide them).
    auto sched_half = [&](bool with_dma) {
        if constexpr (SCHED == 1) {
            if (with_dma) {
                __builtin_amdgcn_sched_group_barrier(0x006, 16, 0);  // VALU|SALU: DMA address arithmetic
                __builtin_amdgcn_sched_group_barrier(0x010, GLDS_PER_STAGE, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // the weight read first: its dequant is the long pole
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 4)
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
        }
    };

    // SCHED == 2: hand-placed half step.  Program order inside a half (sched_barrier(0) pins every line):
    //   MFMA0 R(wq_next) R R [DMA] | MFMA1 R R [DMA] | MFMA2 R R | MFMA3 R R | MFMA4 dq0 | MFMA5 dq1 | MFMA6 dq2 | MFMA7 dq3
    // i.e. <= 2 LDS reads or 6 packed-f16 VALU ops in the shadow of each 32-cycle MFMA.
    auto half_manual = [&](const WFrag& wcur, const Frags& fcur, int nstage, int ns, Frags& fnext, WFrag& wnext,
                           auto dma_tag, int dma_stage, int dma_kt) {
        constexpr bool  DMA = decltype(dma_tag)::value;
        const uint8_t*  sa  = smem + nstage * STAGE_BYTES;
        const uint8_t*  sb  = sa + A_STAGE_BYTES;
        uint8_t*        da  = smem + dma_stage * STAGE_BYTES;
        uint8_t*        db  = da + A_STAGE_BYTES;
        f16x2 wd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = i >> 2, mt = i & 3;
            acc[mt]     = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[e], fcur.xa[e][mt], acc[mt], 0, 0, 0);
            if (i == 0) fnext.wq = *reinterpret_cast<const u32x4*>(sb + b_off + ns * 512);
            if (i < 4) {
                // two activation fragments of the next half: (e', mt') = (i>>1, 2*(i&1)) and (i>>1, 2*(i&1)+1)
                const int ne   = i >> 1;
                const int slot = ((4 * ns + 2 * fh + ne) ^ a_key) << 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nmt     = 2 * (i & 1) + j;
                    fnext.xa[ne][nmt] = *reinterpret_cast<const f16x8*>(sa + nmt * 32 * 128 + a_row_off + slot);
                }
                if constexpr (DMA) {
                    dma16(x_rsrc, a_voff[i], dma_kt * BK * 2, da + (wave * A_GLDS_PER_WAVE + i) * 1024);
                }
            } else {
                const int d = i - 4;
                const u32 wdw = d == 0 ? fnext.wq.x : d == 1 ? fnext.wq.y : d == 2 ? fnext.wq.z : fnext.wq.w;
                dequant_dword(wdw, scale2, wd[2 * d], wd[2 * d + 1]);
                // pure VALU ops float freely through instruction selection; the empty asm pins them to this slot
                asm volatile("" : "+v"(wd[2 * d]), "+v"(wd[2 * d + 1]));
                if constexpr (DMA) {
                    if (d < B_GLDS_PER_WAVE)
                        dma16(w_rsrc, b_voff[d], dma_kt * kTileBytes, db + (wave * B_GLDS_PER_WAVE + d) * 1024);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wnext[0] = make_frag(wd[0], wd[1], wd[2], wd[3]);
        wnext[1] = make_frag(wd[4], wd[5], wd[6], wd[7]);
        // The 8 activation reads above were issued >= 4 MFMAs (128+ cycles) ago: this wait is free, and it hands
        // hipcc's wait-count pass an empty LDS queue at every half-step boundary (otherwise the loop-header
        // merge degrades the next half's first counted wait to lgkmcnt(0), stalling on the read just issued).
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt/expcnt untouched
    };

    // ---- prologue: 3 stages in flight, wait for stage 0 ----
    asm volatile("" ::"v"(scale2));  // force the (tiny) scale load to retire before LDS-DMA is queued behind it
    issue_stage(0, 0);
    if (KT > 1) issue_stage(1, 1);
    if (KT > 2) issue_stage(2, 2);
    if (KT > 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GLDS_PER_STAGE) : "memory");
    else if (KT > 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GLDS_PER_STAGE) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    Frags f0, f1;
    WFrag w0, w1;
    load_frags(0, 0, f0);
    dequant_frags(f0.wq, w0);
    int stage = 0;
    // One K step = two half steps.  While the MFMAs of one half run, the fragments of the next half are read
    // from LDS and its weights dequantised.  AHEAD = how many further K steps exist (clamped to 3): compile-time,
    // so the steady-state loop body is branch-free and the compiler can emit counted lgkmcnt waits.
    auto k_step = [&](int kt, auto ahead_tag) {
        constexpr int AHEAD = decltype(ahead_tag)::value;
        const int     next  = stage + 1 == STAGES ? 0 : stage + 1;
        if constexpr (SCHED == 2) {
            half_manual(w0, f0, stage, 1, f1, w1, std::false_type{}, 0, 0);
            if constexpr (AHEAD >= 1) {
                if constexpr (AHEAD >= 2)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GLDS_PER_STAGE) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                int st3 = stage + 3;
                st3     = st3 >= STAGES ? st3 - STAGES : st3;
                half_manual(w1, f1, next, 0, f0, w0, std::integral_constant<bool, (AHEAD >= 3)>{}, st3, kt + 3);
            } else {
                mma_half(w1, f1);
            }
        } else {
            load_frags(stage, 1, f1);
            mma_half(w0, f0);
            dequant_frags(f1.wq, w1);
            sched_half(false);
            if constexpr (SCHED == 1) __builtin_amdgcn_sched_barrier(0);  // half steps are separate scheduling regions
            if constexpr (AHEAD >= 1) {
                // stage kt+1 must have landed (every wave's pieces); stage kt+2 may stay in flight
                if constexpr (AHEAD >= 2)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GLDS_PER_STAGE) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // the ring slot of stage kt-1 is free: every wave finished reading it before the barrier above
                if constexpr (AHEAD >= 3) {
                    int st3 = stage + 3;
                    st3     = st3 >= STAGES ? st3 - STAGES : st3;
                    issue_stage(st3, kt + 3);
                }
                load_frags(next, 0, f0);  // first half of the next K step, overlapping the MFMAs below
                mma_half(w1, f1);
                dequant_frags(f0.wq, w0);
                sched_half(AHEAD >= 3);
                if constexpr (SCHED == 1) __builtin_amdgcn_sched_barrier(0);
            } else {
                mma_half(w1, f1);
            }
        }
        stage = next;
    };
    int kt = 0;
    // first steady-state step peeled: the loop header then merges two identical wait-counter states (prologue
    // state == latch state), which keeps hipcc's counted lgkmcnt waits exact inside the loop
    if (kt + 3 < KT) k_step(kt++, std::integral_constant<int, 3>{});
    for (; kt + 3 < KT; ++kt) k_step(kt, std::integral_constant<int, 3>{});
    if (kt + 2 < KT) k_step(kt++, std::integral_constant<int, 2>{});
    if (kt + 1 < KT) k_step(kt++, std::integral_constant<int, 1>{});
    k_step(kt, std::integral_constant<int, 0>{});

    // ---- epilogue: acc[mt][r] = y[m0 + 32*mt + fn][n0 + 32*wave + 8*(r>>2) + 4*fh + (r&3)], fp32 -> fp16 ----
    const int nbase = n0 + wave * 32 + 4 * fh;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + mt * 32 + fn;
        if (m < M) {
            f16* yrow = y + (size_t)m * N + nbase;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (nbase + 8 * q < N) {  // N % 16 == 0: a group of 4 columns is all-in or all-out
                    f16x2 lo = {(f16)acc[mt][4 * q + 0], (f16)acc[mt][4 * q + 1]};
                    f16x2 hi = {(f16)acc[mt][4 * q + 2], (f16)acc[mt][4 * q + 3]};
                    if (bias) {  // fp16 add after the fp16 rounding: bit-identical to the reference's separate `+ bias`
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


// =====================================================================================================================
// 8-wave variant: same 128 x 128 x 64 workgroup tile, but the two 32-deep halves of every K step go to two wave
// groups (waves 0-3: k-locals 0..31, waves 4-7: 32..63; each wave still owns 128 rows x 32 columns).  Two waves per
// SIMD: while one wave's MFMAs occupy the matrix pipe the other issues its LDS reads / dequant VALU / DMA, which a
// single in-order wave per SIMD cannot hide (PMC on the 4-wave kernel: 22 % issue stalls + 24 % s_waitcnt/barrier).
// The two groups' fp32 partial sums are added once at the end through LDS.  S8-stage DMA ring, one barrier per K step.
constexpr int THREADS8 = 512;
constexpr int STAGES8  = 6;
constexpr int SMEM8_BYTES = STAGES8 * STAGE_BYTES;  // 144 KiB (also covers the 64 KiB end-of-kernel reduction)
constexpr int DMA8_PER_WAVE = 3;                    // 16 A + 8 B pieces of 1 KiB per stage over 8 waves

// ABLATE (kbench only): 1 = no DMA in the loop, 2 = no dequant, 4 = no LDS fragment reads, 8 = no MFMA, 16 = no barrier,
// 32 = static s_setprio 1 for the second wave group, 64 = s_setprio 1 around every MFMA
template <int ABLATE = 0>
__global__ __launch_bounds__(THREADS8, 2) void gemm_mfma8_kernel(const f16* __restrict__ x, const uint8_t* __restrict__ w,
                                                                 const f16* __restrict__ scales, const f16* __restrict__ bias,
                                                                 f16* __restrict__ y, int M, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp  = wave >> 2;  // which 32-deep half of each K step
    const int wn   = wave & 3;   // which 32-column slice of the tile
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
    // DMA pieces of this wave: A pieces 2*wave, 2*wave+1 (8 rows x 128 B each), B piece wave (one 16-column tile)
    int a_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row  = (wave * 2 + i) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        int       gm   = m0 + row;
        gm             = gm < M ? gm : M - 1;
        a_voff[i]      = (gm * K + slot * 8) * 2;
    }
    int b_voff;
    {
        const int n_tiles_total = N >> 4;
        int       nt            = (n0 >> 4) + wave;
        nt                      = nt < n_tiles_total ? nt : n_tiles_total - 1;
        b_voff                  = nt * KT * kTileBytes + lane * 16;
    }
    auto dma_piece = [&](int i, int stage, int kt) {  // i = 0, 1: A pieces; 2: B piece
        uint8_t* sa = smem + stage * STAGE_BYTES;
        if (i < 2)
            dma16(x_rsrc, a_voff[i], kt * BK * 2, sa + (wave * 2 + i) * 1024);
        else
            dma16(w_rsrc, b_voff, kt * kTileBytes, sa + A_STAGE_BYTES + wave * 1024);
    };

    const int fn = lane & 31, fh = lane >> 5;
    const int b_off     = A_STAGE_BYTES + (wn * 2 + (fn >> 4)) * 1024 + (fn & 15) * 16 + fh * 256 + grp * 512;
    const int a_key     = (fn >> 1) & 7;
    const int a_row_off = fn * 128;
    const int a_slot0   = ((4 * grp + 2 * fh + 0) ^ a_key) << 4;
    const int a_slot1   = ((4 * grp + 2 * fh + 1) ^ a_key) << 4;

    const int   ncol_c = (n0 + wn * 32 + fn) < N ? (n0 + wn * 32 + fn) : N - 1;
    const f16   sc     = scales[ncol_c];
    const f16x2 scale2 = {sc, sc};

    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    struct Frags {
        u32x4 wq;
        f16x8 xa[2][4];
    };
    typedef f16x8 WFrag[2];

    // One K step of one wave: 8 MFMAs on (wcur, fcur); in their shadow the LDS reads of the next K step's fragments
    // (READ), the dequant of the freshly read weights, and this wave's 3 DMA pieces of stage kt+STAGES8-1 (DMA).
    auto step = [&](const WFrag& wcur, const Frags& fcur, auto read_tag, int nstage, Frags& fnext, WFrag& wnext,
                    auto dma_tag, int dma_stage, int dma_kt) {
        constexpr bool READ = decltype(read_tag)::value;
        constexpr bool DMA  = decltype(dma_tag)::value;
        const uint8_t* sa   = smem + nstage * STAGE_BYTES;
        f16x2          wd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = i >> 2, mt = i & 3;
            if constexpr (ABLATE & 64) __builtin_amdgcn_s_setprio(1);
            if constexpr (!(ABLATE & 8))
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[e], fcur.xa[e][mt], acc[mt], 0, 0, 0);
            else
                asm volatile("" ::"v"(wcur[e]), "v"(fcur.xa[e][mt]));
            if constexpr (ABLATE & 64) __builtin_amdgcn_s_setprio(0);
            if constexpr (READ && !(ABLATE & 4)) {
                if (i == 0) fnext.wq = *reinterpret_cast<const u32x4*>(sa + b_off);
                if (i < 4) {
                    const int ne = i >> 1;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int nmt     = 2 * (i & 1) + j;
                        fnext.xa[ne][nmt] = *reinterpret_cast<const f16x8*>(sa + nmt * 32 * 128 + a_row_off +
                                                                            (ne ? a_slot1 : a_slot0));
                    }
                } else if constexpr (!(ABLATE & 2)) {
                    const int d   = i - 4;
                    const u32 wdw = d == 0 ? fnext.wq.x : d == 1 ? fnext.wq.y : d == 2 ? fnext.wq.z : fnext.wq.w;
                    dequant_dword(wdw, scale2, wd[2 * d], wd[2 * d + 1]);
                    asm volatile("" : "+v"(wd[2 * d]), "+v"(wd[2 * d + 1]));  // pin the VALU ops to this MFMA's shadow
                }
            }
            if constexpr (DMA && !(ABLATE & 1)) {
                if (i < DMA8_PER_WAVE) dma_piece(i, dma_stage, dma_kt);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (READ && !(ABLATE & 4)) {
            if constexpr (!(ABLATE & 2)) {
                wnext[0] = make_frag(wd[0], wd[1], wd[2], wd[3]);
                wnext[1] = make_frag(wd[4], wd[5], wd[6], wd[7]);
            } else {
                wnext[0] = __builtin_bit_cast(f16x8, fnext.wq);
                wnext[1] = __builtin_bit_cast(f16x8, fnext.wq);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): free (reads are >= 4 MFMAs old), keeps hipcc's counts exact
        } else if constexpr (READ) {
            wnext[0] = wcur[0];
            wnext[1] = wcur[1];
            fnext    = fcur;
        }
    };

    if constexpr (ABLATE & 32) {
        if (grp == 1) __builtin_amdgcn_s_setprio(1);
    }
    // ---- prologue: STAGES8-1 stages in flight; stage 0 -> fragments ----
    asm volatile("" ::"v"(scale2));
#pragma unroll
    for (int s = 0; s < STAGES8 - 1; ++s) {  // KT >= STAGES8 - 1 by launch contract
#pragma unroll
        for (int i = 0; i < DMA8_PER_WAVE; ++i) dma_piece(i, s, s);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES8 - 2) * DMA8_PER_WAVE) : "memory");  // stage 0 landed
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    WFrag w0, w1;
    {
        f0.wq = *reinterpret_cast<const u32x4*>(smem + b_off);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                f0.xa[e][mt] = *reinterpret_cast<const f16x8*>(smem + mt * 32 * 128 + a_row_off + (e ? a_slot1 : a_slot0));
        f16x2 wd[8];
        dequant_16(f0.wq, scale2, wd);
        w0[0] = make_frag(wd[0], wd[1], wd[2], wd[3]);
        w0[1] = make_frag(wd[4], wd[5], wd[6], wd[7]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }

    // REM = K steps after this one, clamped to STAGES8-1.  REM >= STAGES8-1: steady state (DMA for stage kt+STAGES8-1).
    int  stage = 0;
    auto k_step = [&](int kt, auto rem_tag, const WFrag& wcur, const Frags& fcur, WFrag& wnext, Frags& fnext) {
        constexpr int REM  = decltype(rem_tag)::value;
        const int     next = stage + 1 == STAGES8 ? 0 : stage + 1;
        if constexpr (REM >= 1) {
            // stage kt+1 must have landed in LDS (all waves' pieces); younger stages stay in flight
            constexpr int younger = (REM - 1) < (STAGES8 - 3) ? (REM - 1) : (STAGES8 - 3);
            if constexpr (!(ABLATE & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger * DMA8_PER_WAVE) : "memory");
            if constexpr (!(ABLATE & 16)) __builtin_amdgcn_s_barrier();
            int dst = stage + STAGES8 - 1;
            dst     = dst >= STAGES8 ? dst - STAGES8 : dst;
            step(wcur, fcur, std::true_type{}, next, fnext, wnext, std::integral_constant<bool, (REM >= STAGES8 - 1)>{},
                 dst, kt + STAGES8 - 1);
        } else {
            step(wcur, fcur, std::false_type{}, 0, fnext, wnext, std::false_type{}, 0, 0);
        }
        stage = next;
    };
    using Steady = std::integral_constant<int, STAGES8 - 1>;
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
    float* red = reinterpret_cast<float*>(smem) + (size_t)wn * 64 * 64;  // [reg 0..63][lane]
    if (grp == 1) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(mt * 16 + r) * 64 + lane] = acc[mt][r];
    }
    __syncthreads();
    if (grp == 0) {
        const int nbase = n0 + wn * 32 + 4 * fh;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] += red[(mt * 16 + r) * 64 + lane];
            const int m = m0 + mt * 32 + fn;
            if (m < M) {
                f16* yrow = y + (size_t)m * N + nbase;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (nbase + 8 * q < N) {
                        f16x2 lo = {(f16)acc[mt][4 * q + 0], (f16)acc[mt][4 * q + 1]};
                        f16x2 hi = {(f16)acc[mt][4 * q + 2], (f16)acc[mt][4 * q + 3]};
                        if (bias) {  // fp16 add after the fp16 rounding: bit-identical to the reference's separate `+ bias`
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

}  // namespace gemm
}  // namespace eetq
