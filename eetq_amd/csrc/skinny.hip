// Small-batch (2 <= M <= 16) W8A16 stream GEMM launcher; kernel in skinny_kernel.hpp.
// Covers the reference's batched-GEMV range (m <= 4, kernelLauncher.cu:165-192) and the small-M end of its
// CUTLASS range where the weight stream, not the tensor cores, bounds the time.
#include "skinny_kernel.hpp"

namespace eetq {

namespace {

template <int WAVES, int D, bool EXACT, int XV, int OCC>
int launch_inst(const f16* x, const uint8_t* w, const f16* scales, f16* y, int M, int N, int K, hipStream_t stream)
{
    auto         kern = skinny::skinny_kernel<WAVES, D, EXACT, XV, OCC>;
    const size_t smem = skinny::skinny_smem_bytes(M, K, WAVES);
    if (smem > 64 * 1024) {
        EETQ_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)smem));
    }
    launch_kernel(kern, dim3(N / kTileN), dim3(WAVES * 64), smem, stream, x, w, scales, y, M, N, K);
    return check_hip(hipGetLastError(), "skinny_kernel launch");
}

template <int WAVES, int D, bool EXACT, int OCC>
int launch_xv(const f16* x, const uint8_t* w, const f16* scales, f16* y, int M, int N, int K, hipStream_t stream)
{
    const int need = (M * K / 8 + WAVES * 64 - 1) / (WAVES * 64);
    if (need <= 1) return launch_inst<WAVES, D, EXACT, 1, OCC>(x, w, scales, y, M, N, K, stream);
    if (need <= 2) return launch_inst<WAVES, D, EXACT, 2, OCC>(x, w, scales, y, M, N, K, stream);
    if (need <= 4) return launch_inst<WAVES, D, EXACT, 4, OCC>(x, w, scales, y, M, N, K, stream);
    if (need <= 8) return launch_inst<WAVES, D, EXACT, 8, OCC>(x, w, scales, y, M, N, K, stream);
    return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] skinny path: M*K too large for LDS staging");
}

}  // namespace

bool skinny_supported(int M, int N, int K)
{
    if (M < 1 || M > kSkinnyMaxM) return false;
    const int KT    = K / kTileK;
    const int waves = KT >= 32 ? 16 : KT >= 16 ? 8 : KT >= 4 ? 4 : 1;
    const int need  = (M * K / 8 + waves * 64 - 1) / (waves * 64);
    return skinny::skinny_smem_bytes(M, K, waves) <= 144 * 1024 && need <= 8;
}

int launch_skinny(const f16* x, const uint8_t* w, const f16* scales, f16* y, int M, int N, int K, hipStream_t stream)
{
    if (!skinny_supported(M, N, K))
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] skinny path needs 1 <= M <= 16 and M*K activations that fit LDS");
    const int KT = K / kTileK;
    if (KT == 64) return launch_xv<16, 4, true, 4>(x, w, scales, y, M, N, K, stream);
    if (KT >= 32) return launch_xv<16, 2, false, 4>(x, w, scales, y, M, N, K, stream);
    if (KT >= 16) return launch_xv<8, 2, false, 2>(x, w, scales, y, M, N, K, stream);
    if (KT >= 4) return launch_xv<4, 1, false, 1>(x, w, scales, y, M, N, K, stream);
    return launch_xv<1, 1, false, 1>(x, w, scales, y, M, N, K, stream);
}

}  // namespace eetq
