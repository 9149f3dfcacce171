// Fused int8-dequant + fp16 MFMA GEMM for prefill (M >= 5), gfx950.
//
// Replaces the reference's CUTLASS path: ft::gemm_fp16_int (csrc/cutlass_kernels/fpA_intB_gemm.cu:21-33) ->
// GemmFpAIntB / DqMmaMultistage (csrc/cutlass_extensions/.../gemm/kernel/fpA_intB_gemm.h:296-462,
// gemm/threadblock/dq_mma_multistage.h:310-590).  Numerics contract kept from that code: int8 -> fp16 exact,
// multiplied by the per-column fp16 scale in registers *before* the matrix instruction
// (mma_tensorop_dequantizer.h:259-274), fp32 accumulation (default_fpA_intB_traits.h:110), fp16 store with
// alpha = 1, beta = 0 (epilogue_helpers.h:73-80).  Nothing else is shared with it.
//
// Structure (DESIGN.md "MFMA GEMM"):
//   * workgroup tile 128(M) x 128(N), K step 64, 4 waves; wave w owns all 128 rows x columns [32w, 32w+32)
//     so each dequantised B fragment is reused by 4 MFMAs (M-repeat) -- the int8->fp16 VALU work is the
//     scarce resource, not LDS bandwidth;
//   * v_mfma_f32_32x32x16_f16, 64 fp32 accumulators per lane;
//   * A (fp16 activations) and B (uint8 weights, native tile layout) are staged HBM/L2 -> LDS with
//     global_load_lds_dwordx4 (no VGPR round trip), 4-stage ring, one s_barrier per K step, counted vmcnt
//     so a stage stays in flight across the barrier; LDS->register fragment reads run half a K step
//     ahead of the MFMAs that consume them (one wave per SIMD: nothing else hides LDS latency);
//   * A's LDS image is XOR-swizzled through the *source* address (LDS-DMA writes lane-linear), key
//     (row>>1)&7 on the 16-byte slot: ds_read_b128 of 32 rows at one k offset is conflict-free;
//   * B's LDS image is the global tile image: lane-linear 16-byte slots, conflict-free by construction;
//   * blockIdx -> tile mapping gives each XCD (own L2) a contiguous run of tiles, M fastest, so a weight
//     panel is fetched from HBM by one XCD only.
#include <type_traits>

#include "common.hpp"

namespace eetq {

namespace {

constexpr int BM = 128, BN = 128, BK = 64, STAGES = 4, THREADS = 256;
constexpr int A_STAGE_BYTES = BM * BK * 2;  // 16 KiB
constexpr int B_STAGE_BYTES = BN * BK;      // 8 KiB
constexpr int STAGE_BYTES   = A_STAGE_BYTES + B_STAGE_BYTES;
constexpr int SMEM_BYTES    = STAGES * STAGE_BYTES;  // 96 KiB
constexpr int A_GLDS_PER_WAVE = A_STAGE_BYTES / 1024 / 4;  // 4
constexpr int B_GLDS_PER_WAVE = B_STAGE_BYTES / 1024 / 4;  // 2
constexpr int GLDS_PER_STAGE  = A_GLDS_PER_WAVE + B_GLDS_PER_WAVE;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* gptr, uint8_t* lds_wave_base)
{
    // LDS destination = wave-uniform base + lane*16 (hardware adds the lane offset)
    __builtin_amdgcn_global_load_lds((gbl_void*)gptr, (lds_void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ f16x8 make_frag(f16x2 a, f16x2 b, f16x2 c, f16x2 d)
{
    return f16x8{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

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

    // ---- per-lane source pointers for the LDS-DMA stage copies ----
    const f16* a_src[A_GLDS_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_GLDS_PER_WAVE; ++i) {
        const int row  = (wave * A_GLDS_PER_WAVE + i) * 8 + (lane >> 3);  // 8 rows x 128 B per instruction
        const int slot = (lane & 7) ^ ((row >> 1) & 7);                   // source slot for this LDS slot
        int       gm   = m0 + row;
        gm             = gm < M ? gm : M - 1;  // rows past M: read a valid row, results are never stored
        a_src[i]       = x + (size_t)gm * K + slot * 8;
    }
    const uint8_t* b_src[B_GLDS_PER_WAVE];
    const int      n_tiles_total = N >> 4;
#pragma unroll
    for (int i = 0; i < B_GLDS_PER_WAVE; ++i) {
        int nt   = (n0 >> 4) + wave * B_GLDS_PER_WAVE + i;
        nt       = nt < n_tiles_total ? nt : n_tiles_total - 1;
        b_src[i] = w + (size_t)nt * KT * kTileBytes + lane * 16;
    }

    auto issue_stage = [&](int stage, int kt) {
        uint8_t* sa = smem + stage * STAGE_BYTES;
        uint8_t* sb = sa + A_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_GLDS_PER_WAVE; ++i) glds16(a_src[i] + (size_t)kt * BK, sa + (wave * A_GLDS_PER_WAVE + i) * 1024);
#pragma unroll
        for (int i = 0; i < B_GLDS_PER_WAVE; ++i)
            glds16(b_src[i] + (size_t)kt * kTileBytes, sb + (wave * B_GLDS_PER_WAVE + i) * 1024);
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
    auto mma_half = [&](const Frags& f) {
        f16x2 wd[8];
        dequant_16(f.wq, scale2, wd);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f16x8 wfrag = make_frag(wd[4 * e], wd[4 * e + 1], wd[4 * e + 2], wd[4 * e + 3]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, f.xa[e][mt], acc[mt], 0, 0, 0);
        }
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
    load_frags(0, 0, f0);
    int stage = 0;
    // One K step.  AHEAD = how many further K steps exist (clamped to 3): compile-time so that the loop body
    // is branch-free and the compiler can emit counted lgkmcnt waits (reads for the next half stay in flight).
    auto k_step = [&](int kt, auto ahead_tag) {
        constexpr int AHEAD = decltype(ahead_tag)::value;
        // second half of this K step is fetched from LDS while the first half's MFMAs run
        load_frags(stage, 1, f1);
        mma_half(f0);
        const int next = stage + 1 == STAGES ? 0 : stage + 1;
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
        }
        mma_half(f1);
        stage = next;
    };
    int kt = 0;
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

}  // namespace

int launch_gemm_mfma(const f16* x, const uint8_t* w, const f16* scales, f16* y, int M, int N, int K,
                     hipStream_t stream)
{
    // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device
    static unsigned long long attr_set_mask = 0;
    int                       dev           = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    if (!(attr_set_mask >> (dev & 63) & 1ull)) {
        EETQ_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mfma_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        attr_set_mask |= 1ull << (dev & 63);
    }
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    launch_kernel(gemm_mfma_kernel, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, x, w, scales, y, M, N, K);
    return check_hip(hipGetLastError(), "gemm_mfma_kernel launch");
}

}  // namespace eetq
