// Kernel template of the split-K medium-batch (17 <= M <= 128) MFMA dequant-GEMM (included by gemm_splitk.hip and
// tools/kbench.hip).
//
// Why split K.  At these M the time of the round-1 medium-batch tile (gemm_mid_kernel.hpp) is neither the weight stream nor
// the matrix cores: a workgroup that owns BN columns re-reads all of x (M x K fp16) next to its BN x K weight bytes, only
// N / BN workgroups exist, and each of them walks K in barrier-separated steps with one or two stages in flight -- a chain
// of L2/HBM latencies (N = K = 4096, M = 64: 128 workgroups, 16 steps, 10.9 us; the weights alone stream in ~3 us).
// Two levers, used together:
//   * wider column blocks + K slices: (N / BN) * S workgroups, each ingesting (M * 2 + BN) * K / S bytes -- BN = 64, S = 4
//     gives 256 workgroups of 192 KiB for the shape above instead of 128 of 640 KiB;
//   * no barriers and everything in flight: the four waves of a workgroup are INDEPENDENT streams.  Wave j owns k tile
//     4*step + j of every 256-deep step for all rows and columns, so it needs only that k tile of x: it fetches those
//     128-byte row segments itself by LDS-DMA into a wave-private LDS ring (eight rows x 128 B per 1 KiB piece, full
//     lines; 16-byte slots XOR-swizzled by (row >> 1) & 7 through the source address so the MFMA operand reads are
//     conflict-free) and loads its weight tiles straight into registers in fragment order (the native layout makes a
//     wave-wide 16 B/lane load two contiguous 512-byte runs).  D steps are in flight per wave, counted with s_waitcnt
//     vmcnt; nothing but the issuing wave's own vmcnt orders its ds_reads behind its DMA, so the main loop has no
//     s_barrier at all.  With K / S <= D * 256 the whole slice is requested at kernel entry: one memory latency, like the GEMV.
// MFMA: v_mfma_f32_32x32x16_f16, weights as the A operand (4*MT*NB MFMAs per wave per step; every dequantised fragment
// feeds MT MFMAs, every activation fragment NB).  The four k quarters are added through LDS at the end.
//
// The cross-workgroup reduction is in-launch and deterministic (no float atomics; replicas stay bit-identical):
//   * every slice writes its fp32 partial tile (a "slab", 4*MT*NB KiB) with WRITE-THROUGH 16-byte stores
//     (buffer_store_dwordx4 ... sc1), every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier,
//     one lane takes a ticket with a relaxed agent-scope atomic add on the tile's counter;
//   * the slice that draws the last ticket reads ALL S slabs of the tile back (sc1 loads: served below the per-CU L1,
//     and the XCD's L2 cannot hold an older copy -- slab lines are only ever written in a launch before they are read, and
//     kernel boundaries invalidate) and adds them in slice order 0..S-1, whichever slice it is itself and whatever the
//     arrival order was: the sum is a function of the data alone.  Then the usual epilogue.
//   * counters are monotonic: a tile's counter grows by exactly S per launch, "last" is (old & (S-1)) == S-1; they are
//     zeroed once when the scratch buffer is created and never reset.
// This is the hand-off of cdna_hip_programming.md (section 5, "in-launch split-K reduction", write-through form): no
// placement or dispatch-order assumption; the block-id -> (tile, slice) map below only makes a tile's slices neighbours
// on one XCD when the dispatcher places block b on XCD b % 8 (a speed matter).
#pragma once
#include "common.hpp"
#include "gemm_kernel.hpp"

namespace eetq {
namespace gemm_splitk {

constexpr int kThreads   = 256;
constexpr int kMaxSlices = 4;

// 16-byte non-temporal global load that hipcc does NOT see as a memory operation.  Needed because hipcc's s_waitcnt
// bookkeeping does not count LDS-DMA operations (buffer_load ... lds) as vmcnt events next to ordinary loads: for a register
// load issued between DMA pieces it emits "vmcnt(number of younger REGISTER loads)", which on the hardware counter -- that
// counts the DMA pieces too -- drains every DMA in flight.  With the weight loads hidden, all waiting in the main loop is
// the hand-counted "s_waitcnt vmcnt((D-1) * group)" at the top of a step, followed by a sched_barrier so that no consumer of
// `dst` is scheduled above it (cdna_hip_programming.md 5.7, item 1, form iii).
__device__ __forceinline__ void load16_nt_uncounted(u32x4& dst, const void* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}

template <int MT, int NB, int D>
struct Cfg {
    static constexpr int kRows   = 32 * MT;
    static constexpr int kBN     = 32 * NB;
    static constexpr int kXPW    = 4 * MT;                    // x pieces (8 rows x 128 B) per wave and step
    static constexpr int kWPW    = 2 * NB;                    // weight loads (16 B/lane) per wave and step
    static constexpr int kGroup  = kXPW + kWPW;               // vector memory operations per wave and step
    static constexpr int kStageW = kXPW * 1024;               // bytes of one wave's ring slot
    static constexpr int kRing   = 4 * D * kStageW;           // all four waves
    static constexpr int kRed    = 4 * MT * NB * 16 * 64 * 4; // end-of-kernel cross-wave reduction area
    static constexpr int kSmem   = (kRing > kRed ? kRing : kRed) + 16;
    static constexpr int kSlabFloats = kRows * kBN;           // fp32 partial tile of one slice
    static_assert((D - 1) * kGroup <= 63, "vmcnt is a 6-bit counter");
};

// grid = tiles_n * S workgroups (all of M in one row tile: M <= 32*MT).  slabs: [tiles_n][S][kSlabFloats] floats,
// counters: [tiles_n] unsigned (both unused when S == 1).
template <int MT, int NB, int D, bool KFULL>
__global__ __launch_bounds__(kThreads, (Cfg<MT, NB, D>::kSmem <= 80 * 1024 && MT * NB <= 2) ? 2 : 1) void gemm_splitk_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales, f16* __restrict__ y, int M,
    int N, int K, int S, float* __restrict__ slabs, unsigned* __restrict__ counters, Epilogue ep)
{
    using C = Cfg<MT, NB, D>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int KT   = K >> 6;
    const int steps_total = (KT + 3) >> 2;

    // ---- block id -> (column tile, K slice): a tile's slices are consecutive ids on one XCD when tiles_n % 8 == 0 ----
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
    // K steps of this slice: [s0, s1); slices differ by at most one step
    const int s0 = (int)(((long)steps_total * slice) / S), s1 = (int)(((long)steps_total * (slice + 1)) / S);
    const int n_tiles_total = N >> 4;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);

    // ---- this wave's x pieces: piece i = rows 8i .. 8i+7, the 128-byte segment of the wave's k tile ----
    // LDS image of a piece is lane-linear (row 8i + (lane >> 3), physical slot lane & 7); physical slot p of row r holds
    // logical slot p ^ ((r >> 1) & 7): the permutation goes on the source address, the same XOR on the fragment read.
    int x_voff[C::kXPW];
#pragma unroll
    for (int i = 0; i < C::kXPW; ++i) {
        const int row     = 8 * i + (lane >> 3);
        const int logical = (lane & 7) ^ ((row >> 1) & 7);
        const int gm      = row < M ? row : M - 1;
        x_voff[i]         = (gm * K + logical * 8) * 2;  // + k tile * 128 bytes
    }
    uint8_t* ring = smem + wave * (D * C::kStageW);  // this wave's ring: D slots of kStageW bytes
    // fragment reads go through integer LDS addresses (gemm::lds_read16): with a pointer hipcc can trace back to the LDS
    // array it orders every ds_read behind the most recent LDS-DMA (s_waitcnt vmcnt of nearly everything in flight)
    const int ring_addr = (int)(uint32_t)(uintptr_t)(gemm::lds_void*)smem + wave * (D * C::kStageW);

    // ---- this wave's weight fragments: lane (fn, fh), column block nb: 16-column tile 2*nb + (fn >> 4), k tile of the
    // wave; two 16-byte loads (s = 0, 1) at ((2s + fh) * 16 + (fn & 15)) * 16 inside the tile ----
    const int      fn = lane & 31, fh = lane >> 5;
    const uint8_t* w_lane[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int nt = (n0 >> 4) + 2 * nb + (fn >> 4);
        nt     = nt < n_tiles_total ? nt : n_tiles_total - 1;
        w_lane[nb] = w + (size_t)nt * KT * kTileBytes + (fh * 16 + (fn & 15)) * 16;  // + kt * 1024 + s * 512
    }

    u32x4 wreg[D][NB][2];
    // one group = the wave's vector memory traffic of one step: kXPW DMA pieces, then kWPW register loads.  Steps beyond
    // the slice are clamped to its last step (redundant, cache-resident, never used): no load sits behind a branch and the
    // number of operations in flight is a compile-time constant.
    auto issue_group = [&](int slot, int step) {
        const int st = step < s1 ? step : s1 - 1;
        int       kt = 4 * st + wave;
        kt           = kt < KT ? kt : KT - 1;  // the k tile beyond a ragged K is fetched from the last valid one, never used
        const int x_soff = kt * 128;  // byte offset of k tile kt inside a row
#pragma unroll
        for (int i = 0; i < C::kXPW; ++i) gemm::dma16(x_rsrc, x_voff[i], x_soff, ring + slot * C::kStageW + i * 1024);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                load16_nt_uncounted(wreg[slot][nb][s], w_lane[nb] + (size_t)kt * kTileBytes + s * 512);
    };

    // fragment reads: row 32*mt + fn, logical slot 4s + 2fh + e
    int a_off[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) a_off[s][e] = fn * 128 + (((4 * s + 2 * fh + e) ^ ((fn >> 1) & 7)) << 4);
    // (rows 32*mt + fn: (row >> 1) & 7 == (fn >> 1) & 7 because 32*mt is a multiple of 16)

    f16x2 scale2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ncol = n0 + 32 * nb + fn;
        const f16 sc   = scales[ncol < N ? ncol : N - 1];
        scale2[nb]     = f16x2{sc, sc};
    }

    f32x16 acc[MT][NB];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nb][i] = 0.f;

#pragma unroll
    for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(scale2[nb]));
#pragma unroll
    for (int d = 0; d < D; ++d) issue_group(d, s0 + d);

    for (int base = s0; base < s1; base += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int step = base + d;
            // the group of `step` has landed: exactly D-1 younger groups are in flight behind it
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * C::kGroup) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const bool active = step < s1 && (KFULL || 4 * step + wave < KT);  // wave-uniform
            if (active) {
                const int sa = ring_addr + d * C::kStageW;
                f16x8     xa[2][2][MT];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            xa[s][e][mt] = __builtin_bit_cast(f16x8, gemm::lds_read16(sa + mt * 4096 + a_off[s][e]));
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        f16x2 wd[8];
                        dequant_16(wreg[d][nb][s], scale2[nb], wd);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f16x8 wf = gemm::make_frag(wd[4 * e], wd[4 * e + 1], wd[4 * e + 2], wd[4 * e + 3]);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[s][e][mt], acc[mt][nb], 0, 0, 0);
                        }
                    }
                }
            }
            // the fragment reads of this slot have returned (their MFMAs consumed them): refill it
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_group(d, step + D);
        }
    }

    // ---- add the four k quarters through LDS; wave q then owns accumulator registers 4q..4q+3 of every block ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail groups still target the ring
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][mt][nb][reg][lane]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave * MT + mt) * NB + nb) * 16 + r) * 64 + lane] = acc[mt][nb][r];
    __syncthreads();
    // s4[mt][nb][i] = partial y[32*mt + fn][n0 + 32*nb + 8*wave + 4*fh + i]
    float s4[MT][NB][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) s += red[(((q * MT + mt) * NB + nb) * 16 + 4 * wave + i) * 64 + lane];
                s4[mt][nb][i] = s;
            }

    if (S > 1) {
        // ---- publish this slice's partial tile (write-through), take a ticket ----
        const size_t tile_floats = (size_t)S * C::kSlabFloats;
        const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            slabs + (size_t)tile * tile_floats, 0, (int)(tile_floats * 4), 0x00020000);
        // float4 index inside a slab: ((mt*NB + nb)*4 + wave)*64 + lane
        const int lane_off = (wave * 64 + lane) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const u32x4 v = {__builtin_bit_cast(u32, s4[mt][nb][0]), __builtin_bit_cast(u32, s4[mt][nb][1]),
                                 __builtin_bit_cast(u32, s4[mt][nb][2]), __builtin_bit_cast(u32, s4[mt][nb][3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, s_rsrc, (mt * NB + nb) * 4096 + lane_off,
                                                       slice * C::kSlabFloats * 4, /*sc1*/ 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem + C::kSmem - 16);  // inside the one dynamic LDS array
        if (tid == 0)
            *flag = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *flag;
        if ((ticket & (unsigned)(S - 1)) != (unsigned)(S - 1)) return;  // not the last slice of this tile
        // ---- last arriver: all S slabs, summed in slice order (its own one read back like the others) ----
        u32x4 part[kMaxSlices][MT][NB];
#pragma unroll
        for (int s = 0; s < kMaxSlices; ++s) {
            const int ss = s < S ? s : S - 1;  // clamped, predicated use: no load behind a branch
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    part[s][mt][nb] = __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, (mt * NB + nb) * 4096 + lane_off,
                                                                            ss * C::kSlabFloats * 4, /*sc1*/ 16);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = __builtin_bit_cast(float, (u32)part[0][mt][nb][i]);
#pragma unroll
                    for (int s = 1; s < kMaxSlices; ++s) {
                        const float v = __builtin_bit_cast(float, (u32)part[s][mt][nb][i]);
                        t             = s < S ? t + v : t;
                    }
                    s4[mt][nb][i] = t;
                }
    }

    // ---- epilogue ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 32 * mt + fn;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int ncol = n0 + 32 * nb + 8 * wave + 4 * fh;
            if (m < M && ncol < N) {
                f16x2 lo, hi;
                finish_quad(s4[mt][nb], ep, ncol, lo, hi);
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
