// Debug launcher for the split-K co-residency fault: the library's kernel, caller-owned slabs / counters and a
// caller-chosen dynamic-LDS size (kSmem = 80 KiB -> two workgroups per CU, 84 KiB -> one), so that a failing launch's per-slice
// partial tiles can be read back and compared with a reference (tools/experiments/sk_debug.py).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -Ieetq_amd/csrc tools/experiments/sk_debug.hip -o tools/experiments/libsk_debug.so
#include <hip/hip_runtime.h>

#include "gemm_splitk_kernel.hpp"

using namespace eetq;

template <bool KFULL>
static int go(const f16* x, const uint8_t* w, const f16* scales, f16* y, int M, int N, int K, int S, float* slabs, unsigned* counters,
              int lds, hipStream_t st)
{
    using C   = gemm_splitk::Cfg<2, 1, 2, 2, 4>;
    auto kern = gemm_splitk::gemm_splitk_kernel<2, 1, 2, 2, KFULL, 4, false>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    const int tiles = (N + C::kBN - 1) / C::kBN;
    hipLaunchKernelGGL(kern, dim3(tiles * S), dim3(C::kThreads), lds, st, x, w, scales, y, M, N, K, S, slabs, counters, Epilogue{});
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int sk_launch(const void* x, const void* w, const void* scales, void* y, int M, int N, int K, int S, float* slabs,
                         unsigned* counters, int lds_bytes, void* stream)
{
    auto st = static_cast<hipStream_t>(stream);
    return K % gemm_splitk::kBK == 0
               ? go<true>((const f16*)x, (const uint8_t*)w, (const f16*)scales, (f16*)y, M, N, K, S, slabs, counters, lds_bytes, st)
               : go<false>((const f16*)x, (const uint8_t*)w, (const f16*)scales, (f16*)y, M, N, K, S, slabs, counters, lds_bytes, st);
}
extern "C" int sk_smem() { return gemm_splitk::Cfg<2, 1, 2, 2, 4>::kSmem; }
extern "C" int sk_slab_floats() { return gemm_splitk::Cfg<2, 1, 2, 2, 4>::kSlabFloats; }
