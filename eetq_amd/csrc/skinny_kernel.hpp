// Kernel template of the small-batch (2 <= M <= 16) W8A16 "stream" GEMM: same HBM-bound weight stream as the
// decode GEMV (one workgroup per 16-column tile row, waves split K, 16 B/lane loads straight to registers),
// but the multiply-accumulate of up to 16 batch rows rides on v_mfma_f32_16x16x32_f16 instead of M x dot2:
// per 1 KiB tile 24 VALU (dequant) + 2 MFMA, independent of M.  Included by skinny.hip and tools/kbench.hip.
#pragma once
#include "common.hpp"
#include "gemv_kernel.hpp"

namespace eetq {
namespace skinny {

constexpr int kRowPad = 16;  // bytes added to every LDS row of x: breaks the 256 B-multiple row stride

// Native tile lane (g = lane>>4, c = lane&15) already *is* the MFMA B-operand lane (column c, k group g), so a
// tile load feeds two MFMAs (k-locals 16g..16g+7 and 16g+8..16g+15) without any shuffle.  The A operand is the
// activation fragment x[r][64*kt + 16g + 8e + j] of row r = lane&15 (rows >= M read row M-1; never stored).
template <int WAVES, int D, bool EXACT, int XV, int MIN_WAVES_PER_SIMD>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void skinny_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales, f16* __restrict__ y, int M,
    int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int pitch = K * 2 + kRowPad;  // bytes
    float*    red   = reinterpret_cast<float*>(smem + (size_t)M * pitch);

    const int tid   = threadIdx.x;
    const int ntile = blockIdx.x;
    const int wave  = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane  = tid & 63;
    const int g = lane >> 4, c = lane & 15;
    const int KT = K >> 6;

    u32 sraw = reinterpret_cast<const uint16_t*>(scales)[ntile * 16 + c];
    // activations first (they retire at L2 latency), weight stream right behind
    const int    kvec  = K >> 3;        // 16-byte vectors per row
    const int    xvecs = M * kvec;
    const u32x4* xg    = reinterpret_cast<const u32x4*>(x);
    u32x4        xv[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int v = tid + i * WAVES * 64;
        xv[i]       = xg[v < xvecs ? v : xvecs - 1];
    }
    const u32x4* wp     = reinterpret_cast<const u32x4*>(w + (size_t)ntile * KT * kTileBytes) + wave * 64 + lane;
    const size_t stride = (size_t)WAVES * 64;
    u32x4        buf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) buf[d] = gemv::load_w<true>(wp + d * stride);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int v = tid + i * WAVES * 64;
        if (v < xvecs) {
            const int row = v / kvec, col = v - row * kvec;
            *reinterpret_cast<u32x4*>(smem + (size_t)row * pitch + col * 16) = xv[i];
        }
    }
    asm volatile("" : "+v"(sraw));
    const f16x2 scale2 = as_f16x2(sraw | (sraw << 16));
    __syncthreads();

    const int      arow = c < M ? c : M - 1;
    const uint8_t* xl   = smem + (size_t)arow * pitch + (wave * 64 + 16 * g) * 2;  // + 128*WAVES bytes per tile step
    f32x4          acc  = {0.f, 0.f, 0.f, 0.f};

    auto consume = [&](const u32x4& wv, int t) {
        f16x2 wq[8];
        dequant_16(wv, scale2, wq);
        const uint8_t* xp = xl + (size_t)t * 128 * WAVES;
        const f16x8    a0 = *reinterpret_cast<const f16x8*>(xp);
        const f16x8    a1 = *reinterpret_cast<const f16x8*>(xp + 16);
        const f16x8    b0 = {wq[0].x, wq[0].y, wq[1].x, wq[1].y, wq[2].x, wq[2].y, wq[3].x, wq[3].y};
        const f16x8    b1 = {wq[4].x, wq[4].y, wq[5].x, wq[5].y, wq[6].x, wq[6].y, wq[7].x, wq[7].y};
        acc               = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
        acc               = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc, 0, 0, 0);
    };

    if constexpr (EXACT) {
#pragma unroll
        for (int d = 0; d < D; ++d) consume(buf[d], d);
    } else {
        const int n = (KT - wave + WAVES - 1) / WAVES;  // >= D by launch contract
        int       i = 0;
        for (; i + 2 * D <= n; i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                consume(buf[d], i + d);
                buf[d] = gemv::load_w<true>(wp + (size_t)(i + d + D) * stride);
            }
        }
        const int r = n - (i + D);
        u32x4     tail[D > 1 ? D - 1 : 1];
#pragma unroll
        for (int d = 0; d < D - 1; ++d) {
            const int t = i + D + d;
            tail[d]     = gemv::load_w<true>(wp + (size_t)(t < n ? t : n - 1) * stride);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) consume(buf[d], i + d);
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (d < r) consume(tail[d], i + D + d);
    }

    // ---- cross-wave reduction: acc[i] = partial y[row 4g+i][col c] ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * g + i;
        if (row < M) red[(wave * 16 + row) * 16 + c] = acc[i];
    }
    __syncthreads();
    for (int t = tid; t < M * 16; t += WAVES * 64) {  // one pass unless the workgroup is a single wave
        const int m = t >> 4, cc = t & 15;
        float     s = 0.f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) s += red[(wv * 16 + m) * 16 + cc];
        y[(size_t)m * N + ntile * 16 + cc] = (f16)s;
    }
}

inline size_t skinny_smem_bytes(int M, int K, int waves)
{
    return (size_t)M * (K * 2 + kRowPad) + (size_t)waves * 256 * 4;
}

}  // namespace skinny
}  // namespace eetq
