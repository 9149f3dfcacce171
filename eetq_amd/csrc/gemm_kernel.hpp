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
                                                            const f16* __restrict__ scales, f16* __restrict__ y,
                                                            int M, int N, int K)
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
                    const f16x2 lo = {(f16)acc[mt][4 * q + 0], (f16)acc[mt][4 * q + 1]};
                    const f16x2 hi = {(f16)acc[mt][4 * q + 2], (f16)acc[mt][4 * q + 3]};
                    *reinterpret_cast<u32x2*>(yrow + 8 * q) = u32x2{as_u32(lo), as_u32(hi)};
                }
            }
        }
    }
}

}  // namespace gemm
}  // namespace eetq
