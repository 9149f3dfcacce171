// W4A16 (int4 weight-only) path: quantise, pack / unpack and the fused dequant GEMM.  Placeholder until the kernels land:
// every entry point reports EETQ_ERR_UNSUPPORTED.
#include "common.hpp"

namespace eetq {

int launch_quantize_i4(const void*, int, size_t, size_t, int8_t*, int8_t*, int, void*, float*, hipStream_t)
{
    return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] int4 (quint4x2) weight-only quantization is not implemented in this build");
}
int launch_pack_i4(const int8_t*, size_t, size_t, int8_t*, int, hipStream_t)
{
    return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] int4 weights are not implemented in this build");
}
int launch_unpack_i4(const int8_t*, size_t, size_t, int8_t*, int, hipStream_t)
{
    return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] int4 weights are not implemented in this build");
}
int launch_w4a16(const f16*, const uint8_t*, const f16*, Epilogue, f16*, int, int, int, hipStream_t)
{
    return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] the W4A16 GEMM is not implemented in this build");
}

}  // namespace eetq
