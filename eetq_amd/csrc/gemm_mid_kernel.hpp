// Kernel template of the medium-batch (33 <= M <= 128) LDS-tiled MFMA dequant-GEMM (included by gemm_mid.hip and
// tools/kbench.hip).
//
// At these M the work is dominated by streaming the weights once and by how often the activations are re-read from
// L2; there are too few 128 x 128 tiles to fill the chip and one K step of 64 per barrier is latency-bound.  So:
//   * workgroup tile (32*MT rows = all of M) x 32 columns x 256-deep K step: N/32 workgroups, 16 barriers at K = 4096;
//   * 4 waves split every K step four ways (wave j owns k tile j of the step): 8 MFMAs (MT = 2) per wave per step;
//   * A (fp16, 32*MT x 256) and B (8 native 1 KiB tiles) go through a 3-deep (MT <= 2) or 2-deep LDS-DMA ring;
//   * A's LDS image (512-byte rows) is XOR-swizzled through the source address: 16-byte slot ^= row & 15;
//   * the four K quarters are added through LDS at the end; each wave then stores 8 of the 32 columns.
#pragma once
#include "common.hpp"
#include "gemm_kernel.hpp"

namespace eetq {
namespace gemm_mid {

constexpr int kBN      = 32;
constexpr int kBK      = 256;
constexpr int kThreads = 256;
constexpr int kBBytes  = kBN * kBK;  // 8 KiB: 2 column tiles x 4 k tiles

template <int MT, int STAGES>
struct Cfg {
    static constexpr int kRows     = 32 * MT;
    static constexpr int kABytes   = kRows * kBK * 2;
    static constexpr int kStage    = kABytes + kBBytes;
    static constexpr int kStages   = STAGES;                    // DMA ring depth (2 or 3)
    static constexpr int kSmem     = kStages * kStage;
    static constexpr int kAPW      = kABytes / 1024 / 4;        // A pieces (2 rows of 512 B) per wave and stage
    static constexpr int kBPW      = 2;                         // B pieces (native 1 KiB tiles) per wave and stage
    static constexpr int kPieces   = kAPW + kBPW;
};

// STAGES = 3: deeper prefetch, one workgroup per CU (when there are no more tiles than CUs anyway);
// STAGES = 2: <= 80 KiB of LDS at MT <= 2, two workgroups per CU cover each other's barrier and DMA latency.
// KFULL: K is a multiple of the 256-deep step (all Llama shapes): no wave ever skips a step, so the loop body has no
// branch around the MFMAs -- with one, hipcc keeps the accumulators in VGPRs and copies all of them to AGPRs and back
// around the MFMAs of every step (64 x MT extra VALU instructions per step).
template <int MT, int STAGES, bool KFULL>
__global__ __launch_bounds__(kThreads, (STAGES == 2 && MT <= 2) ? 2 : 1) void gemm_mid_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int M, int N, int K, Epilogue ep)
{
    using C = Cfg<MT, STAGES>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int KT    = K >> 6;
    const int steps = (KT + 3) >> 2;

    const int tiles_n = (N + kBN - 1) / kBN;
    // block b runs on XCD b % 8 (when the dispatcher deals them round-robin): every XCD takes one contiguous eighth of the column
    // tiles, as in gemm_splitk_kernel.hpp (round 5; together with the scale wait below -0.4 .. -2.9 % on the shapes AUTO runs
    // here, tools/experiments/ab_lib.py on one box: 4096 x 6144 M = 24 9.13 -> 8.86 us, 2048 x 8192 M = 64 7.95 -> 7.78)
    int ct = blockIdx.x % tiles_n;
    if ((tiles_n & 7) == 0) ct = (ct & 7) * (tiles_n >> 3) + (ct >> 3);
    const int n0      = ct * kBN;
    const int m0      = (blockIdx.x / tiles_n) * C::kRows;
    const int n_tiles_total = N >> 4;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(w), 0, (int)((size_t)N * K), 0x00020000);

    // ---- DMA pieces of this wave (roles fixed per index: no run-time branch per piece) ----
    //   i < kAPW : A piece p = wave*kAPW + i  -> rows 2p, 2p+1 (512 B each)
    //   else     : B piece b = wave*2 + (i - kAPW) -> column tile b>>2, k tile b&3 of the step
    int dma_voff[C::kPieces];
#pragma unroll
    for (int i = 0; i < C::kPieces; ++i) {
        if (i < C::kAPW) {
            const int p    = wave * C::kAPW + i;
            const int row  = 2 * p + (lane >> 5);
            const int slot = (lane & 31) ^ (row & 15);  // source slot for this LDS slot
            int       gm   = m0 + row;
            gm             = gm < M ? gm : M - 1;
            dma_voff[i]    = (gm * K + slot * 8) * 2;
        } else {
            const int b  = wave * C::kBPW + (i - C::kAPW);
            int       nt = (n0 >> 4) + (b >> 2);
            nt           = nt < n_tiles_total ? nt : n_tiles_total - 1;
            dma_voff[i]  = (nt * KT + (b & 3)) * kTileBytes + lane * 16;  // + step*4 tiles
        }
    }
    auto issue_stage = [&](int buf, int step) {
        uint8_t* sa = smem + buf * C::kStage;
#pragma unroll
        for (int i = 0; i < C::kPieces; ++i) {
            if (i < C::kAPW) {
                // the last step of a K that is not a multiple of 256 reads past the row end into the next row (or is
                // zero-filled by the descriptor bounds check at the very end): those k tiles are never multiplied
                gemm::dma16(x_rsrc, dma_voff[i], step * kBK * 2, sa + (wave * C::kAPW + i) * 1024);
            } else {
                const int b    = wave * C::kBPW + (i - C::kAPW);
                const int kt   = step * 4 + (b & 3);
                const int back = kt < KT ? 0 : (kt - (KT - 1)) * kTileBytes;  // clamp to the last valid k tile
                gemm::dma16(w_rsrc, dma_voff[i] - back, step * 4 * kTileBytes, sa + C::kABytes + b * 1024);
            }
        }
    };

    // ---- fragment addressing: lane (fn, fh); this wave owns k tile `wave` of every step ----
    const int fn = lane & 31, fh = lane >> 5;
    const int b_off = C::kABytes + ((fn >> 4) * 4 + wave) * 1024 + (fn & 15) * 16 + fh * 256;  // + s*512
    const int a_key = fn & 15;
    int       a_slot[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) a_slot[s][e] = ((8 * wave + 4 * s + 2 * fh + e) ^ a_key) << 4;
    const int a_row_off = fn * 512;

    const int   ncol   = n0 + fn;
    const f16   sc     = scales[ncol < N ? ncol : N - 1];
    const f16x2 scale2 = {sc, sc};

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    issue_stage(0, 0);
    // (the scales were asked for ahead of the first stage and return ahead of it: the wait for them -- the K loop below must not
    // contain one the compiler counts -- stands BEHIND the stage's requests, so a cold scale read runs under the stage's latency
    // instead of in front of it; gemm_splitk_kernel.hpp has the measurement)
    asm volatile("" ::"v"(scale2));
    if (C::kStages == 3 && steps > 1) issue_stage(1, 1);
    int buf = 0;
    for (int step = 0; step < steps; ++step) {
        // this wave's pieces of the current stage have landed (a younger stage may stay in flight with a 3-deep ring)
        if (C::kStages == 3 && step + 1 < steps)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::kPieces) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... everyone's have; everyone is done with the buffer refilled below
        const bool     active = KFULL || step * 4 + wave < KT;  // wave-uniform: k tile beyond K on the last step
        const uint8_t* sa     = smem + buf * C::kStage;
        // Order inside a step: (1) all fragment reads of the step are issued, (2) then this wave's DMA pieces of the
        // stage two steps ahead -- an LDS-DMA instruction holds the wave's issue slot for 60+ cycles, ten of them right
        // after the barrier used to delay the step's math by ~800 cycles; now they run under the LDS read latency --
        // (3) then dequant + MFMA.
        u32x4 wq[2];
        f16x8 xa[2][2][MT];
        if (active) {
#pragma unroll
            for (int s = 0; s < 2; ++s) wq[s] = *reinterpret_cast<const u32x4*>(sa + b_off + s * 512);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        xa[s][e][mt] = *reinterpret_cast<const f16x8*>(sa + mt * 32 * 512 + a_row_off + a_slot[s][e]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (step + C::kStages - 1 < steps) {
            int nb = buf + C::kStages - 1;
            nb     = nb >= C::kStages ? nb - C::kStages : nb;
            issue_stage(nb, step + C::kStages - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            asm volatile("" : "+v"(wq[0]), "+v"(wq[1]));  // keep the reads above the dequant below in program order
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x2 wd[8];
                dequant_16(wq[s], scale2, wd);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f16x8 wf = gemm::make_frag(wd[4 * e], wd[4 * e + 1], wd[4 * e + 2], wd[4 * e + 3]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[s][e][mt], acc[mt], 0, 0, 0);
                }
            }
        }
        buf = buf + 1 == C::kStages ? 0 : buf + 1;
    }

    // ---- add the four K quarters through LDS; wave q then owns accumulator registers 4q..4q+3 (8 columns) ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][mt][reg][lane]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    const int nb = n0 + 8 * wave + 4 * fh;  // acc[mt][4*wave + i] = y[m0 + 32*mt + fn][n0 + 8*wave + 4*fh + i]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float s4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) s += red[((q * MT + mt) * 16 + 4 * wave + i) * 64 + lane];
            s4[i] = s;
        }
        const int m = m0 + 32 * mt + fn;
        if (m < M && nb < N) {
            f16x2 lo, hi;
            finish_quad(s4, ep, nb, lo, hi);
            if (ep.residual) {
                const u32x2 r = *reinterpret_cast<const u32x2*>(ep.residual + (size_t)m * N + nb);
                lo            = lo + as_f16x2(r.x);
                hi            = hi + as_f16x2(r.y);
            }
            *reinterpret_cast<u32x2*>(y + (size_t)m * N + nb) = u32x2{as_u32(lo), as_u32(hi)};
        }
    }
}

}  // namespace gemm_mid
}  // namespace eetq
