// W8A16 batched GEMV for decode (M = 1..kGemvMaxM), wave64 reduction, no MFMA, no LDS staging.
//
// Replaces the reference's weight_only_batched_gemv (csrc/weightOnlyBatchedGemv/kernel.h:294-468, launched
// from kernelLauncher.cu:165-192 for m <= 4).  Same math up to summation order, except that partial sums are
// kept in fp32 throughout (the CUDA kernel accumulates 32 products per thread in fp16, kernel.h:325-329):
//   y[m][n] = fp16( sum_k fp32(x[m][k]) * fp32(fp16(q[k][n] * s[n])) )
//
// HBM-bound: K*N weight bytes are read exactly once; everything else is KBs.  Design (DESIGN.md "GEMV"):
//   * one workgroup owns one 16-column tile row of the native layout, which is K/64 contiguous 1 KiB tiles
//     = one contiguous K*16-byte stream; its waves take tiles round-robin;
//   * every load is a whole-tile 16 B/lane global_load_dwordx4 with the non-temporal hint (weights are
//     streamed once); UNROLL tiles are issued before the first is consumed so a 16-wave workgroup has
//     64 KiB in flight per CU;
//   * lane (g = lane>>4, c = lane&15) gets 16 consecutive k of column c: dequantised in registers (v_perm +
//     v_pk_add_f16 + v_pk_mul_f16, exact q then fp16(q*s)) and accumulated with v_dot2c_f32_f16;
//   * reduction: DPP/bpermute across the 4 k-groups of a wave, then across waves through 1 KiB of LDS.
#include "common.hpp"

namespace eetq {

namespace {

template <int M>
__device__ __forceinline__ void gemv_consume(const u32x4& wv, f16x2 scale2, const f16* __restrict__ xrow, int K,
                                             float (&acc)[M])
{
    f16x2 wq[8];
    dequant_16(wv, scale2, wq);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const u32x4 xa   = *reinterpret_cast<const u32x4*>(xrow + (size_t)m * K);
        const u32x4 xb   = *reinterpret_cast<const u32x4*>(xrow + (size_t)m * K + 8);
        const u32   xd[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[m] = __builtin_amdgcn_fdot2(wq[i], as_f16x2(xd[i]), acc[m], false);
    }
}

template <int M, int WAVES, int UNROLL>
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(const f16* __restrict__ x, const uint8_t* __restrict__ w,
                                                          const f16* __restrict__ scales, f16* __restrict__ y,
                                                          int N, int K)
{
    __shared__ float red[WAVES][M][16];
    const int ntile = blockIdx.x;
    const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane  = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const int KT = K >> 6;

    const u32x4* wp = reinterpret_cast<const u32x4*>(w + (size_t)ntile * KT * kTileBytes) + lane;
    const f16*   xg = x + 16 * g;  // this lane's 16-k window inside a 64-k tile

    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.f;

    // The scale is needed before the first dot product (contract: fp16(q*s) first) but must not delay the
    // weight stream: its load is queued here, its first use is pinned *after* the weight loads are issued.
    u32 sraw = reinterpret_cast<const uint16_t*>(scales)[ntile * 16 + c];

    int kt = wave;
    // main loop: UNROLL tiles in flight per wave
    for (; kt + (UNROLL - 1) * WAVES < KT; kt += UNROLL * WAVES) {
        u32x4 wv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) wv[u] = __builtin_nontemporal_load(wp + (size_t)(kt + u * WAVES) * 64);
        asm volatile("" : "+v"(sraw)::"memory");  // weight loads stay at the head of the memory queue
        const f16x2 scale2 = as_f16x2(sraw | (sraw << 16));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) gemv_consume<M>(wv[u], scale2, xg + (size_t)(kt + u * WAVES) * 64, K, acc);
    }
    // tail: one tile at a time
    for (; kt < KT; kt += WAVES) {
        const u32x4 wv = __builtin_nontemporal_load(wp + (size_t)kt * 64);
        asm volatile("" : "+v"(sraw)::"memory");
        const f16x2 scale2 = as_f16x2(sraw | (sraw << 16));
        gemv_consume<M>(wv, scale2, xg + (size_t)kt * 64, K, acc);
    }

    // reduce the 4 k-groups (lanes c, c+16, c+32, c+48)
#pragma unroll
    for (int m = 0; m < M; ++m) {
        acc[m] = wave_xor_add(acc[m], 16);
        acc[m] = wave_xor_add(acc[m], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int m = 0; m < M; ++m) red[wave][m][lane] = acc[m];
    }
    __syncthreads();
    if (threadIdx.x < M * 16) {
        const int m = threadIdx.x >> 4, cc = threadIdx.x & 15;
        float     s = 0.f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) s += red[wv][m][cc];
        y[(size_t)m * N + ntile * 16 + cc] = (f16)s;
    }
}

template <int M>
int launch_m(const f16* x, const uint8_t* w, const f16* scales, f16* y, int N, int K, hipStream_t stream)
{
    const int ntiles = N / kTileN;
    const int KT     = K / kTileK;
    // 16 waves x 4 tiles in flight = 64 KiB per workgroup; shorter K uses fewer waves so every wave has work
    if (KT >= 64) {
        gemv_kernel<M, 16, 4><<<ntiles, 16 * 64, 0, stream>>>(x, w, scales, y, N, K);
    } else if (KT >= 16) {
        gemv_kernel<M, 8, 2><<<ntiles, 8 * 64, 0, stream>>>(x, w, scales, y, N, K);
    } else {
        gemv_kernel<M, 4, 1><<<ntiles, 4 * 64, 0, stream>>>(x, w, scales, y, N, K);
    }
    return check_hip(hipGetLastError(), "gemv_kernel launch");
}

}  // namespace

int launch_gemv(const f16* x, const uint8_t* w, const f16* scales, f16* y, int M, int N, int K, hipStream_t stream)
{
    switch (M) {
        case 1: return launch_m<1>(x, w, scales, y, N, K, stream);
        case 2: return launch_m<2>(x, w, scales, y, N, K, stream);
        case 3: return launch_m<3>(x, w, scales, y, N, K, stream);
        case 4: return launch_m<4>(x, w, scales, y, N, K, stream);
        default: return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] GEMV path only supports M <= 4");
    }
}

}  // namespace eetq
