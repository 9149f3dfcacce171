// Kernel template of the "wide tile" medium-batch (M <= 64) MFMA dequant-GEMM (included by gemm_splitk.hip and tools/kbench.hip).
//
// Why another tiling for 17 <= M <= 64.  The split-K kernel (gemm_splitk_kernel.hpp) re-reads the M x K activations once per 32
// or 64 output columns, through a ring whose stages are mostly activation bytes with two steps of lookahead; its step time
// (~0.8 us at M = 64) survived every change of ring depth, wave count and occupancy (round 3, profiles/r03_kbench_*.txt): at
// those activation bytes the weight-stream rate would need more bytes in flight per CU than the LDS holds.  This kernel cuts
// the activation bytes instead: the workgroup tile is (32*MT rows) x 128 columns, so x is re-read once per 128 columns, and a K
// step is only 64 (or 128) deep, so a stage is 12-32 KiB and the 160 KiB of LDS hold a 4- to 12-deep ring -- 100+ KiB in
// flight per CU with both streams in it.
//
// Structure (4 waves): wave w owns the 32 output columns [32w, 32w + 32) of the tile for ALL of the workgroup's K range: its
// accumulators (MT blocks of v_mfma_f32_32x32x16_f16, weights as the A operand like the other kernels, scale applied
// before the MFMA) never meet another wave's, so there is no cross-wave reduction at all.  x tiles (rows x 128 B per k tile,
// 16-byte slots XOR-swizzled by row through the DMA source address) are shared by the four waves; weight tiles are the native
// 1 KiB tiles, DMA'd as they are.  One barrier per K step.  K slices across workgroups (S = 1, 2, 4) with the split-K kernel's
// in-launch deterministic reduction (write-through slabs, drained, one ticket per tile; the last arriver adds the slabs in slice
// order), sharing its scratch regions.  Like every split launch: one workgroup per CU (the ring takes the whole LDS anyway).
#pragma once
#include "common.hpp"
#include "gemm_kernel.hpp"

namespace eetq {
namespace gemm_wide {

constexpr int kBN        = 128;
constexpr int kThreads   = 256;
constexpr int kMaxSlices = 4;

template <int MT, int KS>
struct Cfg {
    static_assert(MT == 1 || MT == 2, "32 or 64 rows");
    static_assert(KS == 1 || KS == 2, "64- or 128-deep K steps");
    static constexpr int kRows   = 32 * MT;
    static constexpr int kABlock = kRows * 128;          // one k tile of x: rows x 64 halfs
    static constexpr int kABytes = kABlock * KS;
    static constexpr int kBBytes = kBN * 64 * KS;        // 8 native tiles per k tile
    static constexpr int kStage  = kABytes + kBBytes;
    static constexpr int kStagesMax = (156 * 1024) / kStage;
    static constexpr int kStages = kStagesMax > 10 ? 10 : kStagesMax;
    static constexpr int kSmem   = kStages * kStage + 16;  // + the ticket word
    static constexpr int kAP     = kABytes / 1024;       // DMA pieces per stage
    static constexpr int kBP     = kBBytes / 1024;
    static constexpr int kPW     = (kAP + kBP) / 4;      // per wave
    static_assert((kAP + kBP) % 4 == 0, "pieces divide among the four waves");
    static_assert((kStages - 1) * kPW <= 63, "vmcnt is a 6-bit counter");
    static constexpr int kSlabFloats = kRows * kBN;
};

// grid = tiles_n * S workgroups; slabs [tiles_n][S][kSlabFloats] floats and counters [tiles_n] (unused when S == 1)
template <int MT, int KS>
__global__ __launch_bounds__(kThreads, 1) void gemm_wide_kernel(const f16* __restrict__ x, const uint8_t* __restrict__ w,
                                                                const f16* __restrict__ scales, f16* __restrict__ y, int M, int N,
                                                                int K, int S, float* __restrict__ slabs,
                                                                unsigned* __restrict__ counters, Epilogue ep)
{
    using C = Cfg<MT, KS>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lds0 = (int)(uint32_t)(uintptr_t)(gemm::lds_void*)smem;
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int fn = lane & 31, fh = lane >> 5;
    const int KT = K >> 6;                            // 64-deep k tiles
    const int steps_total = (KT + KS - 1) / KS;

    const int tiles_n = (N + kBN - 1) / kBN;
    int       tile, slice;
    if ((tiles_n & 7) == 0) {  // a tile's slices on one XCD when the dispatcher places block b on XCD b % 8 (speed only)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = j % S;
        tile  = (j / S) * 8 + xcd;
    } else {
        slice = blockIdx.x % S;
        tile  = blockIdx.x / S;
    }
    const int n0 = tile * kBN;
    const int s0 = (int)(((long)steps_total * slice) / S), s1 = (int)(((long)steps_total * (slice + 1)) / S);
    const int n_tiles_total = N >> 4;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(w), 0, (int)((size_t)N * K), 0x00020000);

    // ---- this wave's DMA pieces of a stage: piece p = wave * kPW + i; p < kAP: x piece (8 rows x 128 B of k tile p / (rows/8)),
    // else weight tile (p - kAP) % 8 of k tile (p - kAP) / 8 ----
    int  dma_voff[C::kPW];
    int  dma_kt[C::kPW];   // k tile of the step the piece belongs to
#pragma unroll
    for (int i = 0; i < C::kPW; ++i) {
        const int p = wave * C::kPW + i;
        if (p < C::kAP) {
            constexpr int kRB = C::kRows / 8;  // x pieces per k tile
            const int kt = p / kRB, rblk = p % kRB;
            const int row = 8 * rblk + (lane >> 3), slot = lane & 7;
            const int gm = row < M ? row : M - 1;
            dma_voff[i] = (gm * K + ((slot ^ (row & 7)) << 3)) * 2;  // + (step*KS + kt) * 128 bytes
            dma_kt[i]   = kt;
        } else {
            const int q = p - C::kAP, kt = q >> 3, t = q & 7;
            int       nt = (n0 >> 4) + t;
            nt           = nt < n_tiles_total ? nt : n_tiles_total - 1;
            dma_voff[i] = nt * KT * kTileBytes + lane * 16;          // + (step*KS + kt) * 1024 bytes
            dma_kt[i]   = kt;
        }
    }
    auto issue_stage = [&](int buf, int step) {
        uint8_t* sb = smem + buf * C::kStage;
#pragma unroll
        for (int i = 0; i < C::kPW; ++i) {
            const int p = wave * C::kPW + i;
            int       kt = step * KS + dma_kt[i];
            kt           = kt < KT ? kt : KT - 1;  // a k tile beyond K (last step of an odd K / 64): re-read the last one, never used
            if (p < C::kAP)
                gemm::dma16(x_rsrc, dma_voff[i], kt * 128, sb + p * 1024);
            else
                gemm::dma16(w_rsrc, dma_voff[i], kt * kTileBytes, sb + p * 1024);
        }
    };
    // s_waitcnt vmcnt(n * kPW), n = stages that may stay in flight (wave-uniform run-time value, immediate operand)
    auto wait_stages = [&](int n) {
#define EETQ_WIDE_WAIT(NN)                                                              \
    case NN:                                                                            \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NN) * C::kPW > 63 ? 63 : (NN) * C::kPW) : "memory"); \
        break;
        switch (n) {
            EETQ_WIDE_WAIT(1)
            EETQ_WIDE_WAIT(2)
            EETQ_WIDE_WAIT(3)
            EETQ_WIDE_WAIT(4)
            EETQ_WIDE_WAIT(5)
            EETQ_WIDE_WAIT(6)
            EETQ_WIDE_WAIT(7)
            EETQ_WIDE_WAIT(8)
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
#undef EETQ_WIDE_WAIT
    };

    // ---- fragment addresses inside a stage (k tile kt adds kt * kABlock resp. kt * 8 KiB) ----
    // weights: native tile 2*wave + (fn >> 4) of the k tile, lane ((fh + 2 s) * 16 + (fn & 15)) of it: k-locals 16 (fh + 2 s) ..
    const int b_off = C::kABytes + (2 * wave + (fn >> 4)) * 1024 + (fh * 16 + (fn & 15)) * 16;  // + kt * 8192 + s * 512
    // x: row 32 mt + fn, slot (4 s + 2 fh + e) ^ (row & 7)
    int a_off[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) a_off[s][e] = fn * 128 + (((4 * s + 2 * fh + e) ^ (fn & 7)) << 4);  // + mt * 4096 + kt * kABlock

    const int   ncol_s = n0 + 32 * wave + fn;
    const f16   sc     = scales[ncol_s < N ? ncol_s : N - 1];
    const f16x2 scale2 = f16x2{sc, sc};

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    asm volatile("" ::"v"(scale2));
    constexpr int D = C::kStages;
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
        if (s0 + j < s1) issue_stage(j, s0 + j);
    int buf = 0;
    for (int step = s0; step < s1; ++step) {
        {
            const int R = s1 - 1 - step;  // stages after this one; up to D - 2 of them are in flight
            wait_stages(R < D - 2 ? R : D - 2);
        }
        __builtin_amdgcn_s_barrier();  // everyone's pieces of this stage have landed; everyone is done with the stage refilled below
        const int sa = lds0 + buf * C::kStage;
        u32x4     wq[KS][2];
        f16x8     xa[KS][2][2][MT];
#pragma unroll
        for (int kt = 0; kt < KS; ++kt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) wq[kt][s] = gemm::lds_read16(sa + b_off + kt * 8192 + s * 512);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        xa[kt][s][e][mt] =
                            __builtin_bit_cast(f16x8, gemm::lds_read16(sa + kt * C::kABlock + mt * 4096 + a_off[s][e]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (step + D - 1 < s1) issue_stage(buf == 0 ? D - 1 : buf - 1, step + D - 1);  // into the buffer of step - 1
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < KS; ++kt) {
            const bool active = step * KS + kt < KT;  // wave-uniform
            if (active) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x2 wd[8];
                    dequant_16(wq[kt][s], scale2, wd);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f16x8 wf = gemm::make_frag(wd[4 * e], wd[4 * e + 1], wd[4 * e + 2], wd[4 * e + 3]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[kt][s][e][mt], acc[mt], 0, 0, 0);
                    }
                }
            }
        }
        buf = buf + 1 == D ? 0 : buf + 1;
    }

    // acc[mt][4 q + i] = partial y[32 mt + fn][n0 + 32 wave + 8 q + 4 fh + i]
    if (S > 1) {
        const size_t                 tile_floats = (size_t)S * C::kSlabFloats;
        const __amdgpu_buffer_rsrc_t s_rsrc      = __builtin_amdgcn_make_buffer_rsrc(
            slabs + (size_t)tile * tile_floats, 0, (int)(tile_floats * 4), 0x00020000);
        // float4 index inside a slab: ((wave * MT + mt) * 4 + q) * 64 + lane
        const int lane_off = lane * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = {__builtin_bit_cast(u32, acc[mt][4 * q]), __builtin_bit_cast(u32, acc[mt][4 * q + 1]),
                                 __builtin_bit_cast(u32, acc[mt][4 * q + 2]), __builtin_bit_cast(u32, acc[mt][4 * q + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, s_rsrc, ((wave * MT + mt) * 4 + q) * 1024 + lane_off,
                                                       slice * C::kSlabFloats * 4, /*sc1*/ 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem + C::kSmem - 16);
        if (tid == 0) *flag = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *flag;
        if ((ticket & (unsigned)(S - 1)) != (unsigned)(S - 1)) return;  // not the last slice of this tile
        // last arriver: all S slabs (its own read back like the others), added in slice order
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 part[kMaxSlices];
#pragma unroll
                for (int s = 0; s < kMaxSlices; ++s) {
                    const int ss = s < S ? s : S - 1;  // clamped, predicated use: no load behind a branch
                    part[s]      = __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, ((wave * MT + mt) * 4 + q) * 1024 + lane_off,
                                                                         ss * C::kSlabFloats * 4, /*sc1*/ 16);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = __builtin_bit_cast(float, (u32)part[0][i]);
#pragma unroll
                    for (int s = 1; s < kMaxSlices; ++s) {
                        const float v = __builtin_bit_cast(float, (u32)part[s][i]);
                        t             = s < S ? t + v : t;
                    }
                    acc[mt][4 * q + i] = t;
                }
            }
    }

    // ---- epilogue: four consecutive columns per lane and register quad ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 32 * mt + fn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ncol = n0 + 32 * wave + 8 * q + 4 * fh;
            if (m < M && ncol < N) {
                const float a4[4] = {acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]};
                f16x2       lo, hi;
                finish_quad(a4, ep, ncol, lo, hi);
                if (ep.residual) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(ep.residual + (size_t)m * N + ncol);
                    lo            = lo + as_f16x2(r.x);
                    hi            = hi + as_f16x2(r.y);
                }
                *reinterpret_cast<u32x2*>(y + (size_t)m * N + ncol) = u32x2{as_u32(lo), as_u32(hi)};
            }
        }
    }
}

}  // namespace gemm_wide
}  // namespace eetq
