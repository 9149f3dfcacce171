// "Block-owner" form of the split-K medium-batch tile (round 4; included by gemm_splitk.hip).
//
// gemm_splitk_kernel.hpp splits the 256-deep K step among the four waves (wave j multiplies k tile j for every row and column
// block), so after the loop every wave holds a FULL-SIZE partial tile and the four are added through LDS: 4 * MT * NB * 16 KiB
// written and read, ~1.1 us at MT * NB = 2 and ~2.2 us at 4 -- which is what makes the plans with 64-column blocks and four K
// slices (the ones that halve what a CU must ingest: x slice M*K*2/S + 64 KiB of weights) lose to 32-column blocks and two
// slices.  Here the MT * NB blocks of 32 x 32 outputs are distributed over the waves instead and every wave walks ALL the k of a
// step for its own block(s): nothing to add across waves.  The price is dequantisation: the waves that share a column block
// each rebuild the same weights (2x the VALU at MT = 2, NB = 2: 192 operations per wave and step for 16 MFMAs; none at MT = 4,
// NB = 2 where a wave owns the two row blocks (2p, 2p + 1) of one column block and uses every fragment twice), and 2x the LDS
// fragment reads.  Same ring, same LDS-DMA pieces, same waits, same slabs / tickets / sum order as the k-split kernel; slabs
// are laid out so that a block's float4 index is the same in both kernels.
#pragma once
#include "gemm_splitk_kernel.hpp"

namespace eetq {
namespace gemm_splitk {

// blocks per wave: MT * NB / 4 in {1, 2}; BPW = 2 pairs row blocks of one column block
template <int MT, int NB, int SA, int SB, bool KFULL>
__global__ __launch_bounds__(256, (Cfg<MT, NB, SA, SB, 4>::kSmem <= 80 * 1024) ? 2 : 1) void gemm_splitk_own_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales, f16* __restrict__ y, int M,
    int N, int K, int S, float* __restrict__ slabs, unsigned* __restrict__ counters, Epilogue ep)
{
    using C = Cfg<MT, NB, SA, SB, 4>;
    static_assert((MT * NB) % 4 == 0 && MT * NB <= 8, "4 or 8 blocks for 4 waves");
    constexpr int BPW = MT * NB / 4;                 // blocks per wave
    static_assert(BPW == 1 || (BPW == 2 && MT % 2 == 0), "two blocks per wave = two row blocks of one column block");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lds0 = (int)(uint32_t)(uintptr_t)(gemm::lds_void*)smem;
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int KT   = K >> 6;
    const int steps_total = (KT + 3) >> 2;
    // this wave's blocks: column block onb, row blocks omt0 .. omt0 + BPW - 1
    const int onb  = wave % NB;
    const int omt0 = (wave / NB) * BPW;

    const int tiles_n = (N + C::kBN - 1) / C::kBN;
    int       tile, slice;
    if ((tiles_n & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = j % S;
        tile  = (j / S) * 8 + xcd;
    } else {
        slice = blockIdx.x % S;
        tile  = blockIdx.x / S;
    }
    const int n0 = tile * C::kBN;
    const int s0 = (int)(((long)steps_total * slice) / S), s1 = (int)(((long)steps_total * (slice + 1)) / S);
    const int n_tiles_total = N >> 4;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(w), 0, (int)((size_t)N * K), 0x00020000);

    // ---- DMA pieces of this wave: identical to gemm_splitk_kernel (A: rows 2p, 2p+1; B: 16-column tile b>>2, k tile b&3) ----
    int dma_voff[C::kPieces];
#pragma unroll
    for (int i = 0; i < C::kPieces; ++i) {
        if (i < C::kAPW) {
            const int p    = wave * C::kAPW + i;
            const int row  = 2 * p + (lane >> 5);
            const int slot = (lane & 31) ^ (row & 15);
            int       gm   = row;
            gm             = gm < M ? gm : M - 1;
            dma_voff[i]    = (gm * K + slot * 8) * 2;
        } else {
            const int b  = wave * C::kBPW + (i - C::kAPW);
            int       nt = (n0 >> 4) + (b >> 2);
            nt           = nt < n_tiles_total ? nt : n_tiles_total - 1;
            dma_voff[i]  = (nt * KT + (b & 3)) * kTileBytes + lane * 16;
        }
    }
    auto issue_a = [&](int buf, int step) {
        uint8_t* sa = smem + buf * C::kABytes;
#pragma unroll
        for (int i = 0; i < C::kAPW; ++i)
            gemm::dma16(x_rsrc, dma_voff[i], step * kBK * 2, sa + (wave * C::kAPW + i) * 1024);
    };
    auto issue_b = [&](int buf, int step) {
        uint8_t* sb = smem + C::kARing + buf * C::kBBytes;
#pragma unroll
        for (int i = C::kAPW; i < C::kPieces; ++i) {
            const int b    = wave * C::kBPW + (i - C::kAPW);
            const int kt   = step * 4 + (b & 3);
            const int back = kt < KT ? 0 : (kt - (KT - 1)) * kTileBytes;
            gemm::dma16(w_rsrc, dma_voff[i] - back, step * 4 * kTileBytes, sb + b * 1024);
        }
    };
    auto wait_younger = [&](int ya, int yb) {
#define EETQ_SPLITK_WAIT(A, B)                                                                     \
    case (A) * 4 + (B):                                                                            \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * C::kAPW + (B) * C::kBPW) : "memory");      \
        break;
        switch (ya * 4 + yb) {
            EETQ_SPLITK_WAIT(0, 0)
            EETQ_SPLITK_WAIT(1, 1)
            EETQ_SPLITK_WAIT(2, 2)
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
#undef EETQ_SPLITK_WAIT
    };
    static_assert(SA == SB, "the block-owner form uses the shared ring");

    // ---- fragment addressing: lane (fn, fh) ----
    const int fn = lane & 31, fh = lane >> 5;
    // weights of column block onb: 16-column tile (fn >> 4) of its two, k tile kt, k group fh + 2s  (+ kt*1024 + s*512)
    const int b_off = onb * 8192 + (fn >> 4) * 4096 + (fn & 15) * 16 + fh * 256;
    const int a_key = fn & 15;
    const int a_row_off = fn * 512;

    const int   ncol0 = n0 + 32 * onb + fn;
    const f16   sc    = scales[ncol0 < N ? ncol0 : N - 1];
    const f16x2 scale2 = f16x2{sc, sc};

    f32x16 acc[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

    asm volatile("" ::"v"(scale2));
#pragma unroll
    for (int v = -(SB - 1); v < 0; ++v) {
        if (s0 + v + SA - 1 < s1) {
            issue_a((v + SA - 1) % SA, s0 + v + SA - 1);
            issue_b((v + SB - 1) % SB, s0 + v + SB - 1);
        }
    }
    int buf = 0;
    for (int step = s0; step < s1; ++step) {
        {
            const int R = s1 - 1 - step;
            const int y = R < SA - 2 ? R : SA - 2;
            wait_younger(y, y);
        }
        __builtin_amdgcn_s_barrier();
        const int sa = lds0 + buf * C::kABytes;
        const int sb = lds0 + C::kARing + buf * C::kBBytes;
        // order inside a step as in gemm_splitk_kernel: all fragment reads (this wave's block(s), all four k tiles), then its DMA
        // pieces of the stage SA - 1 steps ahead (into the buffer the barrier has just freed; they run under the LDS read
        // latency), then dequant + MFMA
        const int buf_next = buf == 0 ? SA - 1 : buf - 1;
        u32x4 wq[4][2];
        f16x8 xa[4][BPW][2][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) wq[kt][s] = gemm::lds_read16(sb + b_off + kt * 1024 + s * 512);
#pragma unroll
            for (int b = 0; b < BPW; ++b)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        xa[kt][b][s][e] = __builtin_bit_cast(
                            f16x8, gemm::lds_read16(sa + (omt0 + b) * 32 * 512 + a_row_off + (((8 * kt + 4 * s + 2 * fh + e) ^ a_key) << 4)));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (step + SA - 1 < s1) {
            issue_a(buf_next, step + SA - 1);
            issue_b(buf_next, step + SB - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (KFULL || step * 4 + kt < KT) {  // wave-uniform: k tiles beyond K on the last step
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x2 wd[8];
                    dequant_16(wq[kt][s], scale2, wd);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f16x8 wf = gemm::make_frag(wd[4 * e], wd[4 * e + 1], wd[4 * e + 2], wd[4 * e + 3]);
#pragma unroll
                        for (int b = 0; b < BPW; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[kt][b][s][e], acc[b], 0, 0, 0);
                    }
                }
            }
        }
        buf = buf + 1 == SA ? 0 : buf + 1;
    }

    // ---- no cross-wave reduction: r4[b][q][i] = partial y[32*(omt0+b) + fn][n0 + 32*onb + 8*q + 4*fh + i] ----
    float r4[BPW][4][4];
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f = acc[b][4 * q + i];  // (through a named float: see gemm_kernel.hpp on bit casts of vector elements)
                r4[b][q][i]   = f;
            }

    if (S > 1) {
        const size_t tile_floats = (size_t)S * C::kSlabFloats;
        const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            slabs + (size_t)tile * tile_floats, 0, (int)(tile_floats * 4), 0x00020000);
        // float4 index inside a slab (shared with gemm_splitk_kernel): ((mt*NB + nb)*4 + q)*64 + lane
        u32x4 pub[BPW][4];
#pragma unroll
        for (int b = 0; b < BPW; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pub[b][q] = u32x4{__builtin_bit_cast(u32, r4[b][q][0]), __builtin_bit_cast(u32, r4[b][q][1]),
                                  __builtin_bit_cast(u32, r4[b][q][2]), __builtin_bit_cast(u32, r4[b][q][3])};
#pragma unroll
        for (int b = 0; b < BPW; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(pub[b][q], s_rsrc, ((((omt0 + b) * NB + onb) * 4 + q) * 64 + lane) * 16,
                                                       slice * C::kSlabFloats * 4, /*sc1*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY wave drains its write-through stores
        // the stores' DATA registers stay live until here (gemm_splitk_kernel.hpp has the story; the build checks the machine code)
#pragma unroll
        for (int b = 0; b < BPW; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(pub[b][q]));
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem + C::kSmem - 16);
        if (tid == 0) *flag = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *flag;
        if ((ticket & (unsigned)(S - 1)) != (unsigned)(S - 1)) return;  // not the last slice of this tile
        // last arriver: every wave reads ITS blocks from all S slabs and adds them in slice order (its own like the others)
        u32x4 part[kMaxSlices][BPW][4];
#pragma unroll
        for (int s = 0; s < kMaxSlices; ++s) {
            const int ss = s < S ? s : S - 1;
#pragma unroll
            for (int b = 0; b < BPW; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    part[s][b][q] = __builtin_amdgcn_raw_buffer_load_b128(
                        s_rsrc, ((((omt0 + b) * NB + onb) * 4 + q) * 64 + lane) * 16, ss * C::kSlabFloats * 4, /*sc1*/ 16);
        }
#pragma unroll
        for (int b = 0; b < BPW; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = __builtin_bit_cast(float, (u32)part[0][b][q][i]);
#pragma unroll
                    for (int s = 1; s < kMaxSlices; ++s) {
                        const float v = __builtin_bit_cast(float, (u32)part[s][b][q][i]);
                        t             = s < S ? t + v : t;
                    }
                    r4[b][q][i] = t;
                }
    }

    // ---- epilogue: 8-byte stores, row 32*(omt0+b) + fn, columns 8q + 4fh .. +3 of the block ----
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
        const int m = 32 * (omt0 + b) + fn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ncol = n0 + 32 * onb + 8 * q + 4 * fh;
            if (m < M && ncol < N) {
                f16x2 lo, hi;
                finish_quad(r4[b][q], ep, ncol, lo, hi);
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

}  // namespace gemm_splitk
}  // namespace eetq
