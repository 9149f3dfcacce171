// Kernel templates of the W8A16 decode GEMV (included by gemv.hip and by tools/kbench.hip).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace eetq {
namespace gemv {

// tools/kbench_stamps only (never defined in the library build): wave 0 and the last wave of every workgroup record the
// 100 MHz device clock at kernel entry (0), after their last dot product (1) and after the final store (2), for the
// launch-ramp / stream / tail decomposition under profiles/.
#ifdef EETQ_KBENCH_STAMPS
__device__ unsigned long long* g_gemv_stamps = nullptr;
#define EETQ_STAMP(i)                                                                                          \
    do {                                                                                                       \
        if (g_gemv_stamps && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == (blockDim.x >> 6) - 1)) \
            g_gemv_stamps[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) != 0)) * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// same, issued only once `val` (a VGPR) has been produced: orders the stamp behind the math that waited for the loads
#define EETQ_STAMP_AFTER(i, val)                                                                               \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(val) : "memory");              \
        if (g_gemv_stamps && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == (blockDim.x >> 6) - 1)) \
            g_gemv_stamps[((size_t)blockIdx.x * 2 + ((threadIdx.x >> 6) != 0)) * 4 + (i)] = t_;                \
    } while (0)
#else
#define EETQ_STAMP(i) do { } while (0)
#define EETQ_STAMP_AFTER(i, val) do { } while (0)
#endif

template <bool NT>
__device__ __forceinline__ u32x4 load_w(const u32x4* p)
{
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}

// xor-16 / xor-32 butterfly sums with the gfx950 lane-swap instructions (VALU, no LDS round trip).
// RMS-norm of the activation vector while it sits in the staging registers (M = 1): every thread holds XV 16-byte
// vectors xv[i] of x (vector index tid + i*THREADS, valid below xvecs) and the matching gamma vectors.  Same arithmetic as
// rmsnorm_kernel (norm_rope.hip): fp32 sum of squares, rsqrtf(mean + eps), ((x * s) * gamma) clamped to the fp16 range.
// `red` is the (not yet used) cross-wave reduction area; one extra barrier.
template <int XV, int WAVES>
__device__ __forceinline__ void rmsnorm_staged(u32x4 (&xv)[XV], const u32x4 (&gv)[XV], int tid, int xvecs, int K, float eps,
                                               float* red)
{
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const bool  valid = tid + i * WAVES * 64 < xvecs;
        const f16x8 v     = __builtin_bit_cast(f16x8, xv[i]);
        float       p     = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) p += (float)v[j] * (float)v[j];
        ss += valid ? p : 0.f;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) tot += red[wv];
    const float s = rsqrtf(tot / (float)K + eps);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const f16x8 v = __builtin_bit_cast(f16x8, xv[i]);
        const f16x8 g = __builtin_bit_cast(f16x8, gv[i]);
        f16x8       o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = ((float)v[j] * s) * (float)g[j];
            f       = f > 0.f ? fminf(f, 65504.f - 1000.f) : fmaxf(f, -(65504.f - 1000.f));
            o[j]    = (f16)f;
        }
        xv[i] = __builtin_bit_cast(u32x4, o);
    }
}

// Gated-MLP activation on the staged vectors: xv = gate, uv = up -> xv = fp16(silu(gate)) * up (no reduction, no barrier).
template <int XV>
__device__ __forceinline__ void silu_mul_staged(u32x4 (&xv)[XV], const u32x4 (&uv)[XV])
{
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const f16x8 g = __builtin_bit_cast(f16x8, xv[i]);
        const f16x8 u = __builtin_bit_cast(f16x8, uv[i]);
        f16x8       o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (float)g[j];
            o[j]          = (f16)(x / (1.0f + expf(-x))) * u[j];
        }
        xv[i] = __builtin_bit_cast(u32x4, o);
    }
}

// int4 tiles (DESIGN.md "int4 layout"): one dword = 8 consecutive k as unsigned nibbles q + 8 at nibble positions
// [0, 4, 1, 5, 2, 6, 3, 7]; the mask / shift extractions of the reference's converter
// (cutlass_extensions/.../interleaved_numeric_conversion.h:215-280: (w & 0x000f000f) | 0x64006400, - 1032; (w & 0x00f000f0) |
// 0x64006400, * 1/16 - 72) give the exact integers as fp16 pairs (k0,k1) (k2,k3) (k4,k5) (k6,k7); one rounding by the scale.
__device__ __forceinline__ void dequant_dword_i4(u32 w, f16x2 scale2, f16x2 (&out)[4])
{
    const f16x2 c1032 = {(f16)1032.0f, (f16)1032.0f};
    const f16x2 c16th = {(f16)0.0625f, (f16)0.0625f};
    const f16x2 c72   = {(f16)72.0f, (f16)72.0f};
    const u32   top   = w >> 8;
    const f16x2 p0 = as_f16x2((w & 0x000f000fu) | 0x64006400u);
    const f16x2 p1 = as_f16x2((w & 0x00f000f0u) | 0x64006400u);
    const f16x2 p2 = as_f16x2((top & 0x000f000fu) | 0x64006400u);
    const f16x2 p3 = as_f16x2((top & 0x00f000f0u) | 0x64006400u);
    out[0] = (p0 - c1032) * scale2;
    out[1] = (p1 * c16th - c72) * scale2;
    out[2] = (p2 - c1032) * scale2;
    out[3] = (p3 * c16th - c72) * scale2;
}

// weight bits -> k values per 1 KiB tile (16 columns x kTileK<BITS>) and per lane (16 bytes)
template <int BITS>
struct Codec {
    static constexpr int kTileK = BITS == 8 ? 64 : 128;
    static constexpr int kLaneK = BITS == 8 ? 16 : 32;
    static constexpr int kXQ    = kLaneK / 8;  // 16-byte activation vectors per lane and tile
};

// One lane's 16 weight bytes against the matching activations of one batch row (xq: kXQ 16-byte vectors of x).
template <int BITS>
__device__ __forceinline__ float dot_lane(const u32x4& wv, f16x2 scale2, const u32x4* xq, float acc)
{
    if constexpr (BITS == 8) {
        f16x2 wq[8];
        dequant_16(wv, scale2, wq);
        const u32 xd[8] = {xq[0].x, xq[0].y, xq[0].z, xq[0].w, xq[1].x, xq[1].y, xq[1].z, xq[1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_fdot2(wq[i], as_f16x2(xd[i]), acc, false);
    } else {
        const u32 wd[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f16x2 wq[4];
            dequant_dword_i4(wd[d], scale2, wq);
            acc = __builtin_amdgcn_fdot2(wq[0], as_f16x2(xq[d].x), acc, false);
            acc = __builtin_amdgcn_fdot2(wq[1], as_f16x2(xq[d].y), acc, false);
            acc = __builtin_amdgcn_fdot2(wq[2], as_f16x2(xq[d].z), acc, false);
            acc = __builtin_amdgcn_fdot2(wq[3], as_f16x2(xq[d].w), acc, false);
        }
    }
    return acc;
}

// acc += wq . (one dword of x read from lane SRC of this lane's 16-lane row): DPP row broadcast on the dot product's own operand
// -- no LDS, no extra instruction.  In the native tile a 16-lane row is one k-group: its lanes all need the same 16 k of x.
template <int SRC>
__device__ __forceinline__ float dot2_row_bcast(f16x2 wq, u32 xreg, float acc)
{
    asm("v_dot2c_f32_f16_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xreg), "v"(wq), "n"(SRC));
    return acc;
}
// Tile d (0..3) of a wave's four: the row's 32 dwords of x ([tile][dword]) sit two per lane, lane c holding 2c and 2c + 1
// (xp0, xp1); pair i of the tile is dword 8d + i.  Same products, same order as the sixteen-bytes-per-lane form: same bits.
template <int DT>
__device__ __forceinline__ float dot_tile_row_bcast(const f16x2 (&wq)[8], u32 xp0, u32 xp1, float acc)
{
    acc = dot2_row_bcast<DT * 4 + 0>(wq[0], xp0, acc);
    acc = dot2_row_bcast<DT * 4 + 0>(wq[1], xp1, acc);
    acc = dot2_row_bcast<DT * 4 + 1>(wq[2], xp0, acc);
    acc = dot2_row_bcast<DT * 4 + 1>(wq[3], xp1, acc);
    acc = dot2_row_bcast<DT * 4 + 2>(wq[4], xp0, acc);
    acc = dot2_row_bcast<DT * 4 + 2>(wq[5], xp1, acc);
    acc = dot2_row_bcast<DT * 4 + 3>(wq[6], xp0, acc);
    acc = dot2_row_bcast<DT * 4 + 3>(wq[7], xp1, acc);
    return acc;
}

// int4 tiles (128 k per tile, 32 k = 16 dwords of x per lane and tile): tile DT of the wave's D, the row's 16 D dwords of x spread
// over its sixteen lanes, DPL = D per lane (lane c holds dwords DPL c ... DPL c + DPL - 1 of the list [tile][dword]).  Dword d of
// the lane's weights meets x dwords 4 d ... 4 d + 3 of the tile: same products, same order as dot_lane<4>.
template <int DT, int DPL>
__device__ __forceinline__ float dot_tile_row_bcast_i4(const u32x4& wv, f16x2 scale2, const u32 (&xp)[DPL], float acc)
{
    static_assert(DPL == 2 || DPL == 4, "two tiles (K = 4096) or four (K = 8192) per wave");
    const u32 wd[4] = {wv.x, wv.y, wv.z, wv.w};
    auto one = [&](auto d_tag) {
        constexpr int d = decltype(d_tag)::value;
        f16x2         wq[4];
        dequant_dword_i4(wd[d], scale2, wq);
        if constexpr (DPL == 2) {
            constexpr int L = DT * 8 + 2 * d;
            acc = dot2_row_bcast<L>(wq[0], xp[0], acc);
            acc = dot2_row_bcast<L>(wq[1], xp[1], acc);
            acc = dot2_row_bcast<L + 1>(wq[2], xp[0], acc);
            acc = dot2_row_bcast<L + 1>(wq[3], xp[1], acc);
        } else {
            constexpr int L = DT * 4 + d;
            acc = dot2_row_bcast<L>(wq[0], xp[0], acc);
            acc = dot2_row_bcast<L>(wq[1], xp[1], acc);
            acc = dot2_row_bcast<L>(wq[2], xp[2], acc);
            acc = dot2_row_bcast<L>(wq[3], xp[3], acc);
        }
    };
    one(std::integral_constant<int, 0>{});
    one(std::integral_constant<int, 1>{});
    one(std::integral_constant<int, 2>{});
    one(std::integral_constant<int, 3>{});
    return acc;
}

// One 1 KiB tile: this lane's k values of column c against the matching activations of every batch row.
// xs = this lane's window of the LDS copy of x (row m at xs + m*K halfs).
template <int M, int BITS = 8>
__device__ __forceinline__ void consume_tile(const u32x4& wv, f16x2 scale2, const f16* xs, int K, float (&acc)[M])
{
    if constexpr (BITS == 8) {
        f16x2 wq[8];
        dequant_16(wv, scale2, wq);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const u32x4 xa    = *reinterpret_cast<const u32x4*>(xs + m * K);
            const u32x4 xb    = *reinterpret_cast<const u32x4*>(xs + m * K + 8);
            const u32   xd[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[m] = __builtin_amdgcn_fdot2(wq[i], as_f16x2(xd[i]), acc[m], false);
        }
    } else {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            u32x4 xq[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) xq[d] = *reinterpret_cast<const u32x4*>(xs + m * K + 8 * d);
            acc[m] = dot_lane<4>(wv, scale2, xq, acc[m]);
        }
    }
}

// grid.x = N/16 (one workgroup per 16-column tile row = one contiguous K*16-byte stream), block = WAVES*64.
// Wave w takes tiles w, w+WAVES, ...; D tiles (16 B/lane each) are kept in flight per wave.
//
// No load in this kernel sits behind a branch: hipcc answers a conditional load with s_waitcnt vmcnt(0) at
// every join, which serialises the stream.  Instead the launch picks a shape-specialised instantiation:
//   EXACT : K/64 == WAVES*D, every wave owns exactly D tiles -> straight-line code, no loop.
//   else  : every wave owns >= D tiles; software-pipelined loop with unconditional refills, then a clamped
//           (possibly redundant, L2-resident) tail batch whose *use* is predicated.
//   XREG  : (EXACT only) activations go straight to registers, issued ahead of the weight stream so they
//           retire at L2 latency; no LDS, no barrier before the math.  M = 1, int8, D = 4 (the K = 4096 form): ONE 8-byte load
//           per lane -- the 16 lanes of a k-group hold the group's 32 dwords of x between them, two each -- and the dot
//           products take their operand from the owning lane by DPP row broadcast (round 6: 128 KiB of lane data per
//           workgroup became 8; profiles/r06_gemv_ladder3.txt rung 11, -0.07 us, same bits).
//   else  : activations are staged once per workgroup in LDS (XV 16-byte loads per thread, clamped).
// Dynamic LDS: [M*K fp16 activations unless XREG] + WAVES*M*16 floats (cross-wave reduction).
// The body is a device function of the tile row `ntile` so that the grouped launch (one dispatch over the tile rows of
// several problems, gemv_grouped_kernel below) runs exactly the same code as the single-problem kernel.
// PLAIN: y = fp16(acc), no bias / residual / activation -- the reference's own GEMV instantiates its no-bias form at compile time
// too (weightOnlyBatchedGemv/kernelLauncher.cu:165-192: Zero = 0, Bias = 0); the run-time epilogue (argument fetch, three
// branches) is 0.1 us of the 4.6 us launch (profiles/r06_gemv_ladder.txt: ladder rung 5 vs the library kernel).
template <int M, int WAVES, int D, bool EXACT, bool XREG, int XV, int NORM = 0, int BITS = 8, bool PLAIN = false>
__device__ __forceinline__ void gemv_body(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int N, int K, const Epilogue& ep_arg, const Prologue& pro, const int ntile)
{
    static_assert(!XREG || EXACT, "register-resident activations need the exact-fit shape");
    // NORM: 0 = none, 1 = RMS-norm prologue, 2 = gated-MLP activation prologue
    static_assert(!NORM || (!XREG && M == 1), "the activation prologues live in the LDS-staged M = 1 form");
    EETQ_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    f16*   xs  = reinterpret_cast<f16*>(smem);
    float* red = reinterpret_cast<float*>(smem + (XREG ? 0 : (size_t)M * K * 2));

    const int tid   = threadIdx.x;
    const int wave  = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane  = tid & 63;
    const int g = lane >> 4, c = lane & 15;
    using CD = Codec<BITS>;
    constexpr int TK = CD::kTileK, LK = CD::kLaneK, XQ = CD::kXQ;
    const int KT = K / TK;

    // Memory queue order matters (returns are in order): the tiny scale + activation loads go first so they
    // retire at L2 latency while the weight stream is already queued right behind them.
    u32 sraw = reinterpret_cast<const uint16_t*>(scales)[ntile * 16 + c];

    constexpr bool XBC  = XREG && M == 1 && BITS == 8 && D == 4;  // activations by row broadcast (see XREG above)
    constexpr bool XBC4 = XREG && M == 1 && BITS == 4 && (D == 2 || D == 4);  // the same on int4 tiles: 16 D dwords per row
    constexpr int  DPL4 = XBC4 ? D : 2;
    u32x4 xr[XREG && !XBC && !XBC4 ? M * D * XQ : 1];  // XREG: this lane's activations for each of its D tiles, per batch row
    u32x4 xv[XREG ? 1 : XV];
    u32   xp0 = 0, xp1 = 0;
    u32   xp4[DPL4] = {};
    if constexpr (XBC4) {
        // lane (g, c): dwords DPL c ... of row g's list [tile][16 dwords]: tile (DPL c) / 16, dword (DPL c) % 16
        const f16* px = x + (wave + ((DPL4 * c) >> 4) * WAVES) * TK + LK * g + 2 * ((DPL4 * c) & 15);
        if constexpr (DPL4 == 2) {
            using u32x2v = __attribute__((ext_vector_type(2))) u32;
            const u32x2v v = *reinterpret_cast<const u32x2v*>(px);
            xp4[0] = v.x, xp4[1] = v.y;
        } else {
            const u32x4 v = *reinterpret_cast<const u32x4*>(px);
            xp4[0] = v.x, xp4[1] = v.y, xp4[2] = v.z, xp4[3] = v.w;
        }
    } else if constexpr (XBC) {
        // lane (g, c): dwords 2c, 2c + 1 of row g's list [tile d][dword i] -> tile c >> 2, dwords 2 (c & 3), + 1
        using u32x2 = __attribute__((ext_vector_type(2))) u32;
        const u32x2 v = *reinterpret_cast<const u32x2*>(x + (wave + (c >> 2) * WAVES) * TK + LK * g + 4 * (c & 3));
        xp0 = v.x, xp1 = v.y;
    } else if constexpr (XREG) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const u32x4* p = reinterpret_cast<const u32x4*>(x + (size_t)m * K + (wave + d * WAVES) * TK + LK * g);
#pragma unroll
                for (int q = 0; q < XQ; ++q) xr[(m * D + d) * XQ + q] = p[q];
            }
    } else {
        const int    xvecs = (M * K) >> 3;  // 16-byte vectors of x
        const u32x4* xg    = reinterpret_cast<const u32x4*>(x);
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * WAVES * 64;
            xv[i]       = xg[v < xvecs ? v : xvecs - 1];
        }
    }
    u32x4 gv[NORM ? XV : 1];
    if constexpr (NORM) {
        const int    xvecs = K >> 3;
        const u32x4* gg    = reinterpret_cast<const u32x4*>(NORM == 1 ? pro.gamma : pro.up);
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * WAVES * 64;
            gv[i]       = gg[v < xvecs ? v : xvecs - 1];
        }
    }

    const u32x4* wp     = reinterpret_cast<const u32x4*>(w + (size_t)ntile * KT * kTileBytes) + wave * 64 + lane;
    const size_t stride = (size_t)WAVES * 64;  // u32x4 elements between consecutive tiles of this wave
    u32x4        buf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) buf[d] = load_w<true>(wp + d * stride);
    const Epilogue ep = PLAIN ? Epilogue{} : pin_epilogue(ep_arg);

    if constexpr (!XREG) {
        const int xvecs = (M * K) >> 3;
        if constexpr (NORM == 1) rmsnorm_staged<XV, WAVES>(xv, gv, tid, xvecs, K, pro.eps, red);
        if constexpr (NORM == 2) silu_mul_staged<XV>(xv, gv);
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * WAVES * 64;
            if (v < xvecs) reinterpret_cast<u32x4*>(xs)[v] = xv[i];  // store (not load) behind the branch
        }
    }
    asm volatile("" : "+v"(sraw));
    const f16x2 scale2 = as_f16x2(sraw | (sraw << 16));
    if constexpr (!XREG) __syncthreads();

    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.f;

    if constexpr (XBC4) {
        acc[0] = dot_tile_row_bcast_i4<0, DPL4>(buf[0], scale2, xp4, acc[0]);
        acc[0] = dot_tile_row_bcast_i4<1, DPL4>(buf[1], scale2, xp4, acc[0]);
        if constexpr (D == 4) {
            acc[0] = dot_tile_row_bcast_i4<2, DPL4>(buf[2], scale2, xp4, acc[0]);
            acc[0] = dot_tile_row_bcast_i4<3, DPL4>(buf[3], scale2, xp4, acc[0]);
        }
    } else if constexpr (XBC) {
        f16x2 wq[8];
        dequant_16(buf[0], scale2, wq);
        acc[0] = dot_tile_row_bcast<0>(wq, xp0, xp1, acc[0]);
        dequant_16(buf[1], scale2, wq);
        acc[0] = dot_tile_row_bcast<1>(wq, xp0, xp1, acc[0]);
        dequant_16(buf[2], scale2, wq);
        acc[0] = dot_tile_row_bcast<2>(wq, xp0, xp1, acc[0]);
        dequant_16(buf[3], scale2, wq);
        acc[0] = dot_tile_row_bcast<3>(wq, xp0, xp1, acc[0]);
    } else if constexpr (XREG) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if constexpr (BITS == 8) {
                f16x2 wq[8];
                dequant_16(buf[d], scale2, wq);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const u32x4 xa = xr[(m * D + d) * 2], xb = xr[(m * D + d) * 2 + 1];
                    const u32   xd[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[m] = __builtin_amdgcn_fdot2(wq[i], as_f16x2(xd[i]), acc[m], false);
                }
            } else {
#pragma unroll
                for (int m = 0; m < M; ++m) acc[m] = dot_lane<BITS>(buf[d], scale2, &xr[(m * D + d) * XQ], acc[m]);
            }
        }
    } else {
        const f16* xl = xs + wave * TK + LK * g;  // + TK*WAVES halfs per tile step
        if constexpr (EXACT) {
#pragma unroll
            for (int d = 0; d < D; ++d) consume_tile<M, BITS>(buf[d], scale2, xl + (size_t)d * TK * WAVES, K, acc);
        } else {
            const int n = (KT - wave + WAVES - 1) / WAVES;  // tiles of this wave (>= D by launch contract)
            int       i = 0;
            for (; i + 2 * D <= n; i += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    consume_tile<M, BITS>(buf[d], scale2, xl + (size_t)(i + d) * TK * WAVES, K, acc);
                    buf[d] = load_w<true>(wp + (size_t)(i + d + D) * stride);
                }
            }
            // buf holds tiles i..i+D-1; r = n-(i+D) in [0, D) tiles remain: fetch them with clamped indices
            const int r = n - (i + D);
            u32x4     tail[D > 1 ? D - 1 : 1];
#pragma unroll
            for (int d = 0; d < D - 1; ++d) {
                const int t = i + D + d;
                tail[d]     = load_w<true>(wp + (size_t)(t < n ? t : n - 1) * stride);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) consume_tile<M, BITS>(buf[d], scale2, xl + (size_t)(i + d) * TK * WAVES, K, acc);
#pragma unroll
            for (int d = 0; d < D - 1; ++d)
                if (d < r) consume_tile<M, BITS>(tail[d], scale2, xl + (size_t)(i + D + d) * TK * WAVES, K, acc);
        }
    }

    // ---- reduction: 4 k-groups of the wave (lanes c, c+16, c+32, c+48), then across waves via LDS ----
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = sum_xor32(sum_xor16(acc[m]));
    EETQ_STAMP_AFTER(1, acc[0]);
    if (lane < 16) {
#pragma unroll
        for (int m = 0; m < M; ++m) red[(wave * M + m) * 16 + lane] = acc[m];
    }
    __syncthreads();
    if (wave == 0) {
        // lane (g, c) sums waves g, g+4, ...; the butterfly adds the 4 groups
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < (WAVES + 3) / 4; ++wv) {
                const int ww = g + 4 * wv;
                if (ww < WAVES) s += red[(ww * M + m) * 16 + c];
            }
            s = sum_xor32(sum_xor16(s));
            if constexpr (PLAIN) {
                if (lane < 16) y[(size_t)m * N + ntile * 16 + c] = (f16)s;
            } else if (lane < 16) {
                if (ep.act == kActGlu8) {
                    // columns 0..7 of the tile are gate, 8..15 the matching up columns: N/2 outputs per row
                    Epilogue lin = ep;
                    lin.act      = 0;
                    const f16 v  = finish_element(s, lin, ntile * 16 + c);
                    const f16 up = __builtin_bit_cast(f16, (unsigned short)__shfl_xor((int)__builtin_bit_cast(unsigned short, v), 8, 64));
                    if (c < 8) y[(size_t)m * (N >> 1) + ntile * 8 + c] = silu_mul_f16(v, up);
                } else {
                    f16 v = finish_element(s, ep, ntile * 16 + c);  // identity: fp16 add after the fp16 rounding == the reference's separate `+ bias`
                    if (ep.residual) v = v + ep.residual[(size_t)m * N + ntile * 16 + c];
                    y[(size_t)m * N + ntile * 16 + c] = v;
                }
            }
        }
    }
    EETQ_STAMP(2);
}

template <int M, int WAVES, int D, bool EXACT, bool XREG, int XV, int MIN_WAVES_PER_SIMD, int NORM = 0, int BITS = 8, bool PLAIN = false>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void gemv_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int N, int K, const f16* aux, const f16* bias, const f16* residual, int act, float eps)
{
    // the epilogue operands are scalar kernel arguments, not an Epilogue by value: the first 16 argument dwords arrive in SGPRs
    // with the wave (kernarg preload), an aggregate does not -- its fields were fetched by an s_load where they are first used,
    // i.e. between the last weight tile and the store, 0.1 us of every launch (profiles/r06_gemv_ladder.txt).  The prologue's
    // pointer (aux: gamma of the RMS-norm or the `up` half of the gated activation) sits inside those 16 dwords too: behind them
    // (round 5: a Prologue by value at dword 16) its s_load and the wait for it stood in front of the first weight load.
    Epilogue ep;
    ep.bias = bias, ep.residual = residual, ep.act = act;
    Prologue pro;
    pro.gamma = NORM == 1 ? aux : nullptr, pro.up = NORM == 2 ? aux : nullptr, pro.eps = eps;
    gemv_body<M, WAVES, D, EXACT, XREG, XV, NORM, BITS, PLAIN>(x, w, scales, y, N, K, ep, pro, blockIdx.x);
}

// ---- grouped launch: ONE dispatch over the tile rows of up to kMaxGroup independent M = 1 problems of equal K ----------
// A single 16 MiB GEMV dispatch spends ~1.8 us of its ~4.8 on launch ramp, first-byte latency and tail; problems that do not
// depend on each other (the q / k / v or gate / up projections of an unfused layer, the experts of one token, ...) pay that
// once when their tile rows share a grid: workgroup b finds its problem from the first_row table (a scalar scan of the
// kernel arguments) and runs the single-problem body on it.  Same arithmetic, same summation order as gemv_kernel of the
// same instantiation: results are bit-identical to the separate launches.
constexpr int kMaxGroup = 32;
struct GroupedProblem {
    const f16*     x;
    const uint8_t* w;
    const f16*     scales;
    f16*           y;
    const f16*     bias;
    const f16*     residual;
    int            N;
    int            first_row;  // tile rows of the problems before this one
};
struct GroupedArgs {
    GroupedProblem p[kMaxGroup];
    int            count;
};

template <int WAVES, int D, bool EXACT, bool XREG, int XV, int MIN_WAVES_PER_SIMD>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void gemv_grouped_kernel(GroupedArgs g, int K)
{
    const int b = blockIdx.x;
    int       i = 0;
    while (i + 1 < g.count && b >= g.p[i + 1].first_row) ++i;  // wave-uniform
    const GroupedProblem& pr = g.p[i];
    Epilogue              ep;
    ep.bias     = pr.bias;
    ep.residual = pr.residual;
    gemv_body<1, WAVES, D, EXACT, XREG, XV>(pr.x, pr.w, pr.scales, pr.y, pr.N, K, ep, Prologue{}, b - pr.first_row);
}

// ---------------------------------------------------------------------------------------------------------------
// Column-unit forms, M = 1: one workgroup per COLS output columns, COLS = 8 (gemv_half_kernel, N/8 workgroups) or a mix of 8-
// and 4-column units (gemv_mixed_kernel).  For N/16 a little above a multiple of the CU count (N = 5120: 320 tile rows on 256
// CUs) whole tile rows leave a quarter of the CUs with twice the bytes of the others; in 8-column units the busiest CU gets 3
// units of 8 (24 columns) instead of 2 units of 16, and with 8 + 8 + 4 every CU gets its exact share of 20 (round 4).
// A wave instruction covers the COLS-column slices of 64 / (4 * COLS) k tiles: COLS = 8: lane = sub*32 + kg*8 + c reads the 16
// bytes of column col0 + c, k-group kg of k tile 2p + sub -- eight full 128-byte lines; COLS = 4: lane = sub*16 + kg*4 + c, k
// tile 4p + sub -- sixteen 64-byte half lines (the other half belongs to the neighbouring 4-column unit, which the launcher
// places on the same XCD so that the line is fetched into one L2 only).  Wave w owns groups w, w+WAVES, ...; D groups in
// flight; K/64 must be a multiple of the group size (launcher contract).  Activations are staged in LDS like the generic form.
// BITS = 4: the same units on int4 tiles (16 columns x 128 k, a lane's 16 bytes = 32 k of its column).
template <int COLS, int WAVES, int D, int XV, int NORM, int BITS = 8>
__device__ __forceinline__ void gemv_unit_body(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int N, int K, const Epilogue& ep_arg, const Prologue& pro, const int col0)
{
    static_assert(COLS == 8 || COLS == 4, "8- or 4-column units");
    constexpr int G = 64 / (4 * COLS);  // k tiles per wave instruction: 2 or 4
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    f16*   xs  = reinterpret_cast<f16*>(smem);
    float* red = reinterpret_cast<float*>(smem + (size_t)K * 2);

    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int sub = lane / (4 * COLS), kg = (lane / COLS) & 3, c = lane & (COLS - 1);
    constexpr int TK = Codec<BITS>::kTileK, LK = Codec<BITS>::kLaneK;  // k per tile / per lane: 64 / 16 (int8), 128 / 32 (int4)
    const int KT = K / TK, NP = KT / G;  // groups of G k tiles

    u32 sraw = reinterpret_cast<const uint16_t*>(scales)[col0 + c];

    u32x4        xv[XV];
    const int    xvecs = K >> 3;
    const u32x4* xg    = reinterpret_cast<const u32x4*>(x);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int v = tid + i * WAVES * 64;
        xv[i]       = xg[v < xvecs ? v : xvecs - 1];
    }
    u32x4 gv[NORM ? XV : 1];
    if constexpr (NORM) {
        const u32x4* gg = reinterpret_cast<const u32x4*>(NORM == 1 ? pro.gamma : pro.up);
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * WAVES * 64;
            gv[i]       = gg[v < xvecs ? v : xvecs - 1];
        }
    }

    // byte offset of this lane inside its k tile: lane index kg*16 + (col0 % 16 + c) of the native tile
    const uint8_t* wbase = w + (size_t)(col0 >> 4) * KT * kTileBytes + (size_t)sub * kTileBytes +
                           (kg * 16 + (col0 & 15) + c) * 16;
    auto wptr = [&](int grp) { return reinterpret_cast<const u32x4*>(wbase + (size_t)grp * G * kTileBytes); };
    u32x4 buf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) buf[d] = load_w<true>(wptr(wave + d * WAVES));
    const Epilogue ep = pin_epilogue(ep_arg);

    if constexpr (NORM == 1) rmsnorm_staged<XV, WAVES>(xv, gv, tid, xvecs, K, pro.eps, red);
    if constexpr (NORM == 2) silu_mul_staged<XV>(xv, gv);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int v = tid + i * WAVES * 64;
        if (v < xvecs) reinterpret_cast<u32x4*>(xs)[v] = xv[i];  // store (not load) behind the branch
    }
    asm volatile("" : "+v"(sraw));
    const f16x2 scale2 = as_f16x2(sraw | (sraw << 16));
    __syncthreads();

    float      acc[1] = {0.f};
    const f16* xl     = xs + sub * TK + LK * kg;  // + TK * G halfs per group
    const int  n      = (NP - wave + WAVES - 1) / WAVES;  // groups of this wave (>= D by launch contract)
    int        i      = 0;
    for (; i + 2 * D <= n; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            consume_tile<1, BITS>(buf[d], scale2, xl + (size_t)(wave + (i + d) * WAVES) * (TK * G), K, acc);
            buf[d] = load_w<true>(wptr(wave + (i + d + D) * WAVES));
        }
    }
    const int r = n - (i + D);
    u32x4     tail[D > 1 ? D - 1 : 1];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
        const int t = i + D + d;
        tail[d]     = load_w<true>(wptr(wave + (t < n ? t : n - 1) * WAVES));
    }
#pragma unroll
    for (int d = 0; d < D; ++d) consume_tile<1, BITS>(buf[d], scale2, xl + (size_t)(wave + (i + d) * WAVES) * (TK * G), K, acc);
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < r) consume_tile<1, BITS>(tail[d], scale2, xl + (size_t)(wave + (i + D + d) * WAVES) * (TK * G), K, acc);

    // lanes with the same c: 4 k-groups and G k tiles of the group, then across waves via LDS
    float a = acc[0];
    if constexpr (COLS == 4) a += __shfl_xor(a, 4, 64);
    a += __shfl_xor(a, 8, 64);
    a = sum_xor32(sum_xor16(a));
    if (lane < COLS) red[wave * COLS + lane] = a;
    __syncthreads();
    if (tid < COLS) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) s += red[wv * COLS + tid];
        f16 v = finish_element(s, ep, col0 + tid);
        if (ep.residual) v = v + ep.residual[col0 + tid];
        y[col0 + tid] = v;
    }
}

template <int WAVES, int D, int XV, int MIN_WAVES_PER_SIMD, int NORM = 0, int BITS = 8>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void gemv_half_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int N, int K, const f16* aux, const f16* bias, const f16* residual, int act, float eps)
{
    Epilogue ep;  // (scalar arguments: see gemv_kernel)
    ep.bias = bias, ep.residual = residual, ep.act = act;
    Prologue pro;
    pro.gamma = NORM == 1 ? aux : nullptr, pro.up = NORM == 2 ? aux : nullptr, pro.eps = eps;
    gemv_unit_body<8, WAVES, D, XV, NORM, BITS>(x, w, scales, y, N, K, ep, pro, blockIdx.x * 8);
}

// Workgroups [0, n8) take the 8-column units of columns [0, 8 * n8); the rest take 4-column units of the remaining columns,
// ordered so that the two units sharing 128-byte lines (2p, 2p + 1) are 8 block ids apart -- the same XCD when the dispatcher
// places block b on XCD b % 8 (a speed matter only).  With n8 = 2 * CUs and n4 = CUs all workgroups are resident at once and
// every CU streams 20 columns.
template <int WAVES, int D, int XV, int MIN_WAVES_PER_SIMD, int NORM = 0>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void gemv_mixed_kernel(
    const f16* __restrict__ x, const uint8_t* __restrict__ w, const f16* __restrict__ scales,
    f16* __restrict__ y, int N, int K, const f16* aux, const f16* bias, const f16* residual, int act, float eps)
{
    Epilogue ep;  // (scalar arguments: see gemv_kernel)
    ep.bias = bias, ep.residual = residual, ep.act = act;
    Prologue pro;
    pro.gamma = NORM == 1 ? aux : nullptr, pro.up = NORM == 2 ? aux : nullptr, pro.eps = eps;
    const int n8 = N / 4 - (int)gridDim.x;  // 8 n8 + 4 n4 = N and n8 + n4 = gridDim.x (no argument: it would be the seventeenth dword)
    const int b = blockIdx.x;
    if (b < n8) {
        gemv_unit_body<8, WAVES, D, XV, NORM>(x, w, scales, y, N, K, ep, pro, b * 8);
    } else {
        const int q  = b - n8, n4 = (int)gridDim.x - n8;
        int       u4 = q;
        if ((n4 & 15) == 0) u4 = 2 * ((q >> 4) * 8 + (q & 7)) + ((q >> 3) & 1);
        gemv_unit_body<4, WAVES, D, XV, NORM>(x, w, scales, y, N, K, ep, pro, n8 * 8 + u4 * 4);
    }
}

inline size_t gemv_half_smem_bytes(int K, int waves) { return (size_t)K * 2 + (size_t)waves * 8 * 4; }

inline size_t gemv_smem_bytes(int M, int K, int waves, bool xreg)
{
    return (xreg ? 0 : (size_t)M * K * 2) + (size_t)waves * M * 16 * 4;
}

}  // namespace gemv
}  // namespace eetq
