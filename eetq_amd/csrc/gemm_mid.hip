// Medium-batch (M <= 128) LDS-tiled MFMA dequant-GEMM launcher; kernel in gemm_mid_kernel.hpp.
#include "gemm_mid_kernel.hpp"

namespace eetq {

namespace {

template <int MT, int STAGES, bool KFULL>
int launch_full(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream)
{
    using C   = gemm_mid::Cfg<MT, STAGES>;
    auto kern = gemm_mid::gemm_mid_kernel<MT, STAGES, KFULL>;
    if (C::kSmem > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        int st = opt_in_large_lds(kern, opted);
        if (st != EETQ_OK) return st;
    }
    const int tiles = ((N + gemm_mid::kBN - 1) / gemm_mid::kBN) * ((M + C::kRows - 1) / C::kRows);
    launch_kernel(kern, dim3(tiles), dim3(gemm_mid::kThreads), C::kSmem, stream, x, w, scales, y, M, N, K, ep);
    return check_hip(hipGetLastError(), "gemm_mid_kernel launch");
}

template <int MT, int STAGES>
int launch_inst(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream)
{
    return K % gemm_mid::kBK == 0 ? launch_full<MT, STAGES, true>(x, w, scales, ep, y, M, N, K, stream)
                                  : launch_full<MT, STAGES, false>(x, w, scales, ep, y, M, N, K, stream);
}

}  // namespace

int launch_gemm_mid(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                    hipStream_t stream)
{
    if (M < 1 || M > kMidMaxM) return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] medium-batch tile path supports 1 <= M <= 128");
    EETQ_REQUIRE((size_t)M * K * 2 < (1ull << 31) && (size_t)N * K < (1ull << 31),
                 "operand larger than 2 GiB is not supported by the buffer-addressed DMA path");
    // ring depth: with at most one tile per CU a 3-deep ring hides more DMA latency; with more tiles than CUs the
    // 2-deep ring (<= 80 KiB) lets two workgroups share a CU instead (measured: profiles/r01_kbench_mid.txt)
    const bool deep = (N + gemm_mid::kBN - 1) / gemm_mid::kBN <= 256;
    switch ((M + 31) / 32) {
        case 1: return deep ? launch_inst<1, 3>(x, w, scales, ep, y, M, N, K, stream)
                            : launch_inst<1, 2>(x, w, scales, ep, y, M, N, K, stream);
        case 2: return deep ? launch_inst<2, 3>(x, w, scales, ep, y, M, N, K, stream)
                            : launch_inst<2, 2>(x, w, scales, ep, y, M, N, K, stream);
        case 3: return launch_inst<3, 2>(x, w, scales, ep, y, M, N, K, stream);
        default: return launch_inst<4, 2>(x, w, scales, ep, y, M, N, K, stream);
    }
}

}  // namespace eetq
