// Single-query ("decode") attention over a KV cache, fp16 in / fp16 out, fp32 softmax and accumulation.
//
// No counterpart in the reference's csrc: its EETLlamaAttention delegates the attention product to flash-attn
// (python/eetq/modules/llama_modules.py:131-143).  The stock library kernel this box offers for a single query token runs
// one workgroup per head (40 workgroups streaming a 25 MB cache: 68 us per layer at Llama-13B shapes), so the decode step
// of eet_accelerator's attention block uses this split-KV form instead: HBM/L2-bound byte work, no MFMA.
//   phase 1  grid (splits, heads, batch), 256 threads: a workgroup owns a contiguous chunk of cache positions; a group of
//            D/8 lanes owns one position at a time (16-byte loads of its k and v rows, fully coalesced across the wave),
//            keeps an online-softmax state (m, l) and 8 output channels per lane; groups and waves are merged through LDS
//            and the chunk's (m, l, o[D]) goes to an fp32 workspace;
//   phase 2  grid (heads, batch), D threads: merges the chunks, normalises, writes fp16.
#include "common.hpp"

namespace eetq {

namespace {

constexpr int kAttnThreads = 256;

template <int D>
__global__ __launch_bounds__(kAttnThreads) void attn_decode_partial_kernel(
    const f16* __restrict__ q, const f16* __restrict__ kc, const f16* __restrict__ vc, const f16* __restrict__ mask,
    float* __restrict__ ws, float scaling, int S, int chunk, int groups, long q_sb, long q_sh, long k_sb, long k_sh,
    long k_ss, long v_sb, long v_sh, long v_ss, long m_sb, const int64_t* __restrict__ kv_len, int kv_len_bias)
{
    constexpr int LPP  = D / 8;               // lanes per position
    constexpr int PPW  = 64 / LPP;            // positions per wave instruction
    constexpr int SETS = (kAttnThreads / 64) * PPW;
    __shared__ float sm_m[SETS], sm_l[SETS];
    __shared__ float sm_o[SETS][D];

    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z, hk = h / groups;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = lane / LPP, li = lane % LPP, d0 = li * 8;
    // rows at and beyond the valid length of a pre-allocated (static) cache hold zeros or stale tokens: never attended
    const int Sv = kv_len ? max(0, min(S, (int)min((int64_t)S, *kv_len + kv_len_bias))) : S;
    const int j0 = split * chunk, j1 = min(Sv, j0 + chunk);

    float qf[8];
    {
        const f16x8 qv = *reinterpret_cast<const f16x8*>(q + b * q_sb + h * q_sh + d0);
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = (float)qv[i] * scaling;
    }
    const f16* kbase = kc + b * k_sb + hk * k_sh + d0;
    const f16* vbase = vc + b * v_sb + hk * v_sh + d0;
    const f16* mrow  = mask ? mask + b * m_sb : nullptr;

    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;

    // U position groups per trip: 2U independent 16-byte loads per lane in flight (the loop is latency-bound otherwise)
    constexpr int U = 4, STEP = (kAttnThreads / 64) * PPW;
    for (int jb = j0 + wave * PPW; jb < j1; jb += U * STEP) {
        f16x8 kv[U], vv[U];
        int   jj[U];
        bool  valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * STEP + grp;
            valid[u]    = j < j1;
            jj[u]       = valid[u] ? j : j1 - 1;  // clamped, predicated use: no load behind a branch
            kv[u]       = *reinterpret_cast<const f16x8*>(kbase + (long)jj[u] * k_ss);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = *reinterpret_cast<const f16x8*>(vbase + (long)jj[u] * v_ss);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += qf[i] * (float)kv[u][i];
#pragma unroll
            for (int off = 1; off < LPP; off <<= 1) s += __shfl_xor(s, off, 64);
            if (mrow) s += (float)mrow[jj[u]];
            if (!valid[u]) s = -INFINITY;
            const float mn = fmaxf(m, s);
            if (mn > -INFINITY) {  // group-uniform
                const float sc = __expf(m - mn), p = __expf(s - mn);
                l = l * sc + p;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = o[i] * sc + p * (float)vv[u][i];
                m = mn;
            }
        }
    }
    const int set = wave * PPW + grp;
    if (li == 0) {
        sm_m[set] = m;
        sm_l[set] = l;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sm_o[set][d0 + i] = o[i];
    __syncthreads();
    if (tid < D) {
        float M = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < SETS; ++s2) M = fmaxf(M, sm_m[s2]);
        float L = 0.f, O = 0.f;
        if (M > -INFINITY) {
#pragma unroll
            for (int s2 = 0; s2 < SETS; ++s2) {
                const float w = __expf(sm_m[s2] - M);  // exp(-inf) = 0 for empty sets
                L += sm_l[s2] * w;
                O += sm_o[s2][tid] * w;
            }
        }
        float* out = ws + (((size_t)b * gridDim.y + h) * gridDim.x + split) * (D + 2);
        out[2 + tid] = O;
        if (tid == 0) {
            out[0] = M;
            out[1] = L;
        }
    }
}

// one workgroup per (head, batch): thread t < splits fetches that chunk's (m, l) in parallel; D threads then sum the chunk
// outputs with the loads of up to 8 chunks in flight
template <int D>
__global__ __launch_bounds__(D) void attn_decode_merge_kernel(const float* __restrict__ ws, f16* __restrict__ out,
                                                             int splits, long o_sb, long o_sh, int64_t* advance)
{
    extern __shared__ float sm_w[];  // [splits] weights exp(m_s - M), then sm_w[splits] = 1 / L
    const int    h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* p = ws + ((size_t)b * gridDim.x + h) * splits * (D + 2);
    float        M = -INFINITY;
    for (int s = d; s < splits; s += D) M = fmaxf(M, p[s * (D + 2)]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    if (D > 64) {
        __shared__ float sm_part[D / 64];
        if ((d & 63) == 0) sm_part[d >> 6] = M;
        __syncthreads();
        M = sm_part[0];
#pragma unroll
        for (int i = 1; i < D / 64; ++i) M = fmaxf(M, sm_part[i]);
    }
    for (int s = d; s < splits; s += D) sm_w[s] = M > -INFINITY ? __expf(p[s * (D + 2)] - M) : 0.f;
    __syncthreads();
    float L = 0.f, O = 0.f;
#pragma unroll 8
    for (int s = 0; s < splits; ++s) {
        const float w = sm_w[s];
        L += p[s * (D + 2) + 1] * w;
        O += p[s * (D + 2) + 2 + d] * w;
    }
    out[b * o_sb + h * o_sh + d] = (f16)(L > 0.f ? O / L : 0.f);  // a fully masked row yields zeros, not NaN
    // the cache's token counter (every reader of it in this step -- the cache-write launch and the partial kernel -- has
    // completed: they are earlier launches on the stream)
    if (advance && h == 0 && b == 0 && d == 0) *advance += 1;
}

template <int D>
int launch_d(const f16* q, const f16* k, const f16* v, const f16* mask, f16* out, float* ws, int B, int H, int Hkv, int S,
             int splits, float scaling, const long* st, const int64_t* kv_len, int kv_len_bias, int64_t* advance,
             hipStream_t stream)
{
    const int chunk = (S + splits - 1) / splits;
    attn_decode_partial_kernel<D><<<dim3(splits, H, B), kAttnThreads, 0, stream>>>(
        q, k, v, mask, ws, scaling, S, chunk, H / Hkv, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8],
        kv_len, kv_len_bias);
    EETQ_TRY_HIP(hipGetLastError());
    attn_decode_merge_kernel<D><<<dim3(H, B), D, splits * sizeof(float), stream>>>(ws, out, splits, st[9], st[10], advance);
    return check_hip(hipGetLastError(), "attn_decode kernels launch");
}

}  // namespace

int launch_attn_decode(const f16* q, const f16* k, const f16* v, const f16* mask, f16* out, float* ws, int B, int H, int Hkv,
                       int S, int D, int splits, float scaling, const long* strides, const int64_t* kv_len, int kv_len_bias,
                       int64_t* advance, hipStream_t stream)
{
    EETQ_REQUIRE(q && k && v && out && ws && strides, "null pointer");
    EETQ_REQUIRE(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && S > 0 && splits > 0 && splits <= S, "invalid attention shape");
    for (int i = 0; i < 9; ++i)
        if (i != 8) EETQ_REQUIRE(strides[i] % 8 == 0, "q / k / v strides must be multiples of 8 elements (16-byte loads)");
    EETQ_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0, "q, k, v must be 16-byte aligned");
    if (D == 128)
        return launch_d<128>(q, k, v, mask, out, ws, B, H, Hkv, S, splits, scaling, strides, kv_len, kv_len_bias, advance, stream);
    if (D == 64)
        return launch_d<64>(q, k, v, mask, out, ws, B, H, Hkv, S, splits, scaling, strides, kv_len, kv_len_bias, advance, stream);
    return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] decode attention supports head_dim 64 and 128");
}

}  // namespace eetq
