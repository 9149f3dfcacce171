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
#include "gemm_kernel.hpp"

namespace eetq {

using namespace gemm;

int launch_gemm_mfma(const f16* x, const uint8_t* w, const f16* scales, const f16* bias, f16* y, int M, int N, int K,
                     hipStream_t stream)
{
    // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device
    static unsigned long long attr_set_mask = 0;
    int                       dev           = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    if (!(attr_set_mask >> (dev & 63) & 1ull)) {
        EETQ_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mfma_kernel<2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        EETQ_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mfma8_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM8_BYTES));
        attr_set_mask |= 1ull << (dev & 63);
    }
    EETQ_REQUIRE((size_t)M * K * 2 < (1ull << 31) && (size_t)N * K < (1ull << 31),
                 "operand larger than 2 GiB is not supported by the buffer-addressed DMA path");
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (K / BK >= STAGES8 - 1)  // 8 waves (two K halves per step, 6-stage DMA ring): needs >= 5 K steps
        launch_kernel(gemm_mfma8_kernel<0>, dim3(tiles), dim3(THREADS8), SMEM8_BYTES, stream, x, w, scales, bias, y, M, N, K);
    else
        launch_kernel(gemm_mfma_kernel<2>, dim3(tiles), dim3(THREADS), SMEM_BYTES, stream, x, w, scales, bias, y, M, N, K);
    return check_hip(hipGetLastError(), "gemm_mfma_kernel launch");
}

}  // namespace eetq
