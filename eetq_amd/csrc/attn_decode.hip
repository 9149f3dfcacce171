// Single-query ("decode") attention over a KV cache, fp16 in / fp16 out, fp32 softmax and accumulation.
//
// No counterpart in the reference's csrc: its EETLlamaAttention delegates the attention product to flash-attn
// (python/eetq/modules/llama_modules.py:131-143).  The stock library kernel this box offers for a single query token runs
// one workgroup per head (40 workgroups streaming a 25 MB cache: 68 us per layer at Llama-13B shapes), so the decode step
// of eet_accelerator's attention block uses this split-KV form instead: HBM/L2-bound byte work, no MFMA.
//
// Two-launch form (eetq_decode_attention_f16):
//   phase 1  grid (splits, heads, batch), 256 threads: a workgroup owns a contiguous chunk of the VALID cache positions; a
//            group of D/8 lanes owns one position at a time (16-byte loads of its k and v rows, fully coalesced across the
//            wave), keeps an online-softmax state (m, l) and 8 output channels per lane; groups and waves are merged
//            through LDS and the chunk's (m, l, o[D]) goes to an fp32 workspace;
//   phase 2  grid (heads, batch), 256 threads: merges the chunks (attn_merge), normalises, writes fp16.
// One-launch form (eetq_rope_decode_attention_f16), the decode step of a static cache: the same phase 1 with the NeoX
// rotation of the new token's q and k done in registers on the way in (the arithmetic of rotary_neox_kvcache_kernel, fp16
// with a rounding after every multiply and add), the rotated k and the v written to their cache row by one workgroup per
// kv head, every workgroup taking the new row from registers rather than from the cache (so nothing in the launch reads
// what the launch writes), and phase 2 done by whichever workgroup of a head finishes last: partials are published with
// write-through stores, a ticket per head decides "last" (the in-launch hand-off of gemm_splitk_kernel.hpp), and the last
// head to finish advances the cache's token counter.  Three launches (15.0 us per layer at 13B shapes, 1 k rows) become one
// (10.9 us).  Both forms run the
// same chunk code and the same merge arithmetic: their outputs are bit-identical.
#include <type_traits>

#include "common.hpp"

namespace eetq {
// decode steps whose new token was NOT written because its cache row was outside the cache (full static cache, negative
// position): stock StaticLayer.update raises there, these kernels skip the write -- and count it here
__device__ unsigned g_attn_dropped = 0;
}  // namespace eetq

namespace eetq {

namespace {

constexpr int kAttnThreads = 256;
// A chunk record: kRecPad + D floats = [m, l, -, -, o[D]] (o on a 16-byte boundary).
constexpr int kRecPad = 4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One "trip" of a wave through the cache: U position groups, i.e. 2U independent 16-byte loads per lane in flight (the
// loop is latency-bound otherwise).  The FIRST trip of a workgroup is loaded by the caller before anything else it has to
// do (scalar setup, the rotation of the new token), so the cache rows are in flight while that runs.
constexpr int kTripU = 4;
template <int D>
struct KvTrip {
    f16x8 k[kTripU], v[kTripU];
    int   jj[kTripU];
    bool  valid[kTripU];
};

template <int D>
__device__ __forceinline__ void load_trip(KvTrip<D>& t, const f16* __restrict__ kbase, const f16* __restrict__ vbase, long k_ss,
                                          long v_ss, int jb, int j1)
{
    constexpr int LPP = D / 8, STEP = (kAttnThreads / 64) * (64 / LPP);
    const int     grp = (threadIdx.x & 63) / LPP;
#pragma unroll
    for (int u = 0; u < kTripU; ++u) {
        const int j = jb + u * STEP + grp;
        t.valid[u]  = j < j1;
        t.jj[u]     = t.valid[u] ? j : max(j1 - 1, 0);  // clamped, predicated use: no load behind a branch
        t.k[u]      = *reinterpret_cast<const f16x8*>(kbase + (long)t.jj[u] * k_ss);
    }
#pragma unroll
    for (int u = 0; u < kTripU; ++u) t.v[u] = *reinterpret_cast<const f16x8*>(vbase + (long)t.jj[u] * v_ss);
}

// Sum over the LPP = 8 or 16 consecutive lanes that share a position, by DPP (no LDS traffic): quad swaps, then the
// mirrored half row, then the mirrored row.  Every lane of the group ends with the same bits (each step adds the same two
// partial sums in both partners).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, moved);
}
template <int LPP>
__device__ __forceinline__ float group_sum(float v)
{
    static_assert(LPP == 8 || LPP == 16, "a position is shared by 8 or 16 lanes");
    v = dpp_add<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v = dpp_add<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    if (LPP == 16) v = dpp_add<0x140>(v);  // row_mirror
    return v;
}

// The chunk [j0, j1) of one (batch row, head): online softmax over its positions, merged across the workgroup.  On return
// threads tid < D hold (M, L, O) = the chunk's running maximum, its sum of exp(s - M) and channel tid of sum exp(s - M) v.
// qv: this lane's 8 channels of the (rotated) query; scores are scaling * (q . k) with the products summed in fp32
// (v_dot2_f32_f16).  `t` holds the wave's first trip (load_trip at jb = j0 + wave * PPW).  The running state is rescaled
// once per trip (U positions), not once per position.
// SUBST: position `slot` is taken from registers (knew, vnew: this lane's 8 channels of the new token) instead of the cache.
template <int D, bool SUBST>
__device__ __forceinline__ void attn_chunk(const f16x8& qv, float scaling, KvTrip<D>& t, const f16* __restrict__ kbase,
                                           const f16* __restrict__ vbase, long k_ss, long v_ss, const f16* __restrict__ mrow,
                                           int j0, int j1, int slot, const f16x8& knew, const f16x8& vnew, float* sm_m,
                                           float* sm_l, float* sm_o, float& M, float& L, float& O)
{
    // every multiply-add below is explicit and contraction is off: the two launch forms instantiate this code separately
    // and must round identically
#pragma clang fp contract(off)
    constexpr int LPP  = D / 8;               // lanes per position
    constexpr int PPW  = 64 / LPP;            // positions per wave instruction
    constexpr int SETS = (kAttnThreads / 64) * PPW;
    constexpr int U    = kTripU;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = lane / LPP, li = lane % LPP, d0 = li * 8;
    const f16x2 q2[4] = {{qv[0], qv[1]}, {qv[2], qv[3]}, {qv[4], qv[5]}, {qv[6], qv[7]}};

    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;

    // Trips are software-pipelined: the next trip's rows are requested (clamped to the chunk: unconditional loads, never
    // behind a branch) before the current trip is consumed, so a chunk of several trips pays the memory latency once.  The
    // last iteration's request is wasted (one clamped row).
    constexpr int STEP = (kAttnThreads / 64) * PPW;
    KvTrip<D>     nxt;
    for (int jb = j0 + wave * PPW;;) {
        const int jn = jb + U * STEP;
        load_trip<D>(nxt, kbase, vbase, k_ss, v_ss, jn, j1);
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (SUBST && t.jj[u] == slot) {
                t.k[u] = knew;
                t.v[u] = vnew;
            }
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) a = __builtin_amdgcn_fdot2(f16x2{t.k[u][2 * i], t.k[u][2 * i + 1]}, q2[i], a, false);
            a = group_sum<LPP>(a) * scaling;
            if (mrow) a += (float)mrow[t.jj[u]];
            sc[u] = t.valid[u] ? a : -INFINITY;
        }
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) mn = fmaxf(mn, sc[u]);
        if (mn > -INFINITY) {  // group-uniform
            const float keep = __expf(m - mn);
            l *= keep;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] *= keep;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = __expf(sc[u] - mn);  // exp(-inf) = 0 for the positions beyond the chunk
                l += p;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = fmaf(p, (float)t.v[u][i], o[i]);
            }
            m = mn;
        }
        if (jn >= j1) break;  // wave-uniform
        t  = nxt;
        jb = jn;
    }
    const int set = wave * PPW + grp;
    if (li == 0) {
        sm_m[set] = m;
        sm_l[set] = l;
    }
    *reinterpret_cast<f32x4*>(sm_o + set * D + d0)     = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(sm_o + set * D + d0 + 4) = f32x4{o[4], o[5], o[6], o[7]};
    __syncthreads();
    M = -INFINITY, L = 0.f, O = 0.f;
    if (tid < D) {
#pragma unroll
        for (int s2 = 0; s2 < SETS; ++s2) M = fmaxf(M, sm_m[s2]);
        if (M > -INFINITY) {
#pragma unroll
            for (int s2 = 0; s2 < SETS; ++s2) {
                const float w = __expf(sm_m[s2] - M);  // exp(-inf) = 0 for empty sets
                L = fmaf(sm_l[s2], w, L);
                O = fmaf(sm_o[s2 * D + tid], w, O);
            }
        }
    }
}

// rows at and beyond the valid length of a pre-allocated (static) cache hold zeros or stale tokens: never attended
__device__ __forceinline__ int valid_len(int S, const int64_t* kv_len, int kv_len_bias)
{
    return kv_len ? max(0, (int)min((int64_t)S, *kv_len + kv_len_bias)) : S;
}

template <int D>
__global__ __launch_bounds__(kAttnThreads) void attn_decode_partial_kernel(
    const f16* __restrict__ q, const f16* __restrict__ kc, const f16* __restrict__ vc, const f16* __restrict__ mask,
    float* __restrict__ ws, float scaling, int S, int groups, long q_sb, long q_sh, long k_sb, long k_sh,
    long k_ss, long v_sb, long v_sh, long v_ss, long m_sb, const int64_t* __restrict__ kv_len, int kv_len_bias)
{
    constexpr int SETS = (kAttnThreads / 64) * (64 / (D / 8));
    __shared__ float sm_m[SETS], sm_l[SETS];
    __shared__ __attribute__((aligned(16))) float sm_o[SETS * D];

    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z, hk = h / groups;
    const int tid = threadIdx.x, d0 = ((tid & 63) % (D / 8)) * 8;
    const int Sv = valid_len(S, kv_len, kv_len_bias);
    const int chunk = (Sv + (int)gridDim.x - 1) / (int)gridDim.x;
    const int j0 = split * chunk, j1 = min(Sv, j0 + chunk);

    const f16 *kbase = kc + b * k_sb + hk * k_sh + d0, *vbase = vc + b * v_sb + hk * v_sh + d0;
    KvTrip<D>  trip;
    load_trip<D>(trip, kbase, vbase, k_ss, v_ss, j0 + (tid >> 6) * (64 / (D / 8)), j1);
    const f16x8 qv = *reinterpret_cast<const f16x8*>(q + b * q_sb + h * q_sh + d0);
    float       M, L, O;
    const f16x8 none = {};
    attn_chunk<D, false>(qv, scaling, trip, kbase, vbase, k_ss, v_ss, mask ? mask + b * m_sb : nullptr, j0, j1, -1, none, none,
                         sm_m, sm_l, sm_o, M, L, O);
    if (tid < D) {
        float* out = ws + (((size_t)b * gridDim.x + split) * gridDim.y + h) * (D + kRecPad);  // [batch][split][head][record]
        out[kRecPad + tid] = O;
        if (tid == 0) {
            out[0] = M;
            out[1] = L;
        }
    }
}


// how the merge reads records: plain loads in the two-launch form (the records come from an earlier launch), loads served
// below the per-CU L1 in the one-launch form (they come from other workgroups of this launch)
struct PlainRecords {
    const float* base;
    __device__ f32x2 v2(long off) const { return *reinterpret_cast<const f32x2*>(base + off); }
    __device__ f32x4 v4(long off) const { return *reinterpret_cast<const f32x4*>(base + off); }
};
struct CoherentRecords {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ f32x2 v2(long off) const
    {
        return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(off * 4), 0, /*sc1*/ 16));
    }
    __device__ f32x4 v4(long off) const
    {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off * 4), 0, /*sc1*/ 16));
    }
};

// Merge of a head's `splits` chunk records at offsets s * stride (floats) by a workgroup of NT threads (all of them call:
// there is a barrier inside); thread d < D returns channel d, normalised.  The cost of this step is the number of memory
// INSTRUCTIONS a wave issues (~90 cycles each: 51 of them per wave made an earlier form take 3.3 us at 17 records), so:
// lane i of every wave fetches (m, l) of record i with one 8-byte load; a group of D/4 lanes owns a subset of the records
// (every NS-th) and fetches four channels of each with one 16-byte load -- 1 + ceil(splits / NS) loads per wave; the subsets
// are added through lane swaps and one LDS exchange (sm_x: NT/64 * D floats).  Sums run in a fixed order that depends only
// on (D, NT, splits): both launch forms use NT = 256 and give the same bits.  A fully masked row yields zeros, not NaN.
template <int D, int NT, typename Records>
__device__ __forceinline__ float attn_merge(const Records& rec, long stride, int splits, int tid, float* sm_x)
{
#pragma clang fp contract(off)
    constexpr int G = D / 4, WS = 64 / G, NW = NT / 64, NS = NW * WS;  // lanes per record, subsets per wave / per workgroup
    const int lane = tid & 63, wave = tid >> 6, cg = lane % G, rs = wave * WS + lane / G;
    float M = -INFINITY, L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // one block of up to 64 records, JN loads of this thread's records in flight (JN chosen by the caller from the block's
    // record count: straight-line loads, no load behind a branch)
    auto block = [&](int s0, int n, auto jn_tag) {
        constexpr int JN = decltype(jn_tag)::value;
        f32x4 o4[JN];
#pragma unroll
        for (int j = 0; j < JN; ++j) o4[j] = rec.v4((s0 + min(rs + NS * j, n - 1)) * stride + kRecPad + 4 * cg);
        const f32x2 ml = rec.v2((s0 + min(lane, n - 1)) * stride);
        const float mi = lane < n ? ml.x : -INFINITY, li = lane < n ? ml.y : 0.f;
        float Mb = mi;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) Mb = fmaxf(Mb, __shfl_xor(Mb, off, 64));
        Mb = fmaxf(Mb, M);
        if (Mb > -INFINITY) {  // uniform
            const float keep = __expf(M - Mb);   // 0 for the first block
            const float wi   = __expf(mi - Mb);  // weight of record s0 + lane (exp(-inf) = 0 beyond the block and for empty chunks)
            float x = li * wi;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
            L = fmaf(L, keep, x);
            acc *= keep;
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                const int   idx = rs + NS * j;
                const float w   = __shfl(wi, idx < n ? idx : 0, 64);
                if (idx < n) {
                    acc.x = fmaf(o4[j].x, w, acc.x);
                    acc.y = fmaf(o4[j].y, w, acc.y);
                    acc.z = fmaf(o4[j].z, w, acc.z);
                    acc.w = fmaf(o4[j].w, w, acc.w);
                }
            }
            M = Mb;
        }
    };
    for (int s0 = 0; s0 < splits; s0 += 64) {
        const int n = min(64, splits - s0);
        if (n <= NS)
            block(s0, n, std::integral_constant<int, 1>{});
        else if (n <= 2 * NS)
            block(s0, n, std::integral_constant<int, 2>{});
        else if (n <= 3 * NS)
            block(s0, n, std::integral_constant<int, 3>{});
        else if (n <= 4 * NS)
            block(s0, n, std::integral_constant<int, 4>{});
        else
            block(s0, n, std::integral_constant<int, (64 + NS - 1) / NS>{});
    }
    // the wave's subsets (lanes G apart), then the waves (in order) through LDS
    float part[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (WS == 4) part[c] = sum_xor16(part[c]);
        part[c] = sum_xor32(part[c]);
    }
    if (lane < G) *reinterpret_cast<f32x4*>(sm_x + wave * D + 4 * cg) = f32x4{part[0], part[1], part[2], part[3]};
    __syncthreads();
    if (tid >= D) return 0.f;
    float O = sm_x[tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) O += sm_x[w * D + tid];
    return L > 0.f ? O / L : 0.f;
}

// one workgroup per (head, batch): thread t < splits fetches that chunk's (m, l) in parallel; D threads then sum the chunk
// outputs with the loads of up to 8 chunks in flight
template <int D>
__global__ __launch_bounds__(kAttnThreads) void attn_decode_merge_kernel(const float* __restrict__ ws, f16* __restrict__ out,
                                                                        int splits, long o_sb, long o_sh, int64_t* advance)
{
    __shared__ __attribute__((aligned(16))) float sm_x[(kAttnThreads / 64) * D];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    // workspace [batch][split][head][record]: the records of one head are a row of heads apart
    const PlainRecords rec{ws + ((size_t)b * splits * gridDim.x + h) * (D + kRecPad)};
    const float        o = attn_merge<D, kAttnThreads>(rec, (long)gridDim.x * (D + kRecPad), splits, tid, sm_x);
    if (tid < D) out[b * o_sb + h * o_sh + tid] = (f16)o;
    // the cache's token counter (every reader of it in this step -- the cache-write launch and the partial kernel -- has
    // completed: they are earlier launches on the stream)
    if (advance && h == 0 && b == 0 && tid == 0) *advance += 1;
}

struct RopeAttnArgs {
    const f16 *    q, *k, *v;  // the new token: [batch][heads][D] with q_sb / k_sb / v_sb elements between batch rows
    int            S, groups, kv_len_bias, slot_stride;
    long           kc_sb, kc_sh, kc_ss, vc_sb, vc_sh, vc_ss;
    long           q_sb, k_sb, v_sb;
    const f16*     cos_sin;
    const f16*     mask;
    long           m_sb;
    f16*           out;
    long           o_sb, o_sh;
    float*         ws;
    unsigned*      tickets;  // [batch * heads] per-head arrival counts + [1] finished heads; zero between launches
    int64_t*       advance;
    float          scaling;
    unsigned long long* stamps;  // diagnostics (eetq_diag_attn_stamps): [workgroup][8] device-clock stamps, or null
};

// device clock (100 MHz) into slot i of this workgroup's stamp row, ordered behind everything issued so far
#define ATTN_STAMP(i)                                                                                                  \
    do {                                                                                                               \
        if (a.stamps) {                                                                                                \
            unsigned long long t_;                                                                                     \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            if (threadIdx.x == 0)                                                                                      \
                a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = t_;     \
        }                                                                                                              \
    } while (0)


// NeoX rotation of this lane's 8 channels [d0, d0 + 8) of one head (rot_dim = D: channel d < D/2 pairs with d + D/2).
// own / other: the lane's channels and the paired ones; cs: the position's cos|sin row.  fp16 arithmetic, one rounding per
// multiply and add, exactly rotary_neox_kvcache_kernel (norm_rope.hip).
template <int D>
__device__ __forceinline__ f16x8 rope8(const f16x8& own, const f16x8& other, const f16* __restrict__ cs, int d0)
{
#pragma clang fp contract(off)
    constexpr int embed = D / 2;
    const bool    low   = d0 < embed;
    const int     off   = low ? d0 : d0 - embed;
    const f16x8   c = *reinterpret_cast<const f16x8*>(cs + off), s = *reinterpret_cast<const f16x8*>(cs + embed + off);
    f16x8         r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f16 vx = low ? own[i] : other[i], vy = low ? other[i] : own[i];
        const f16 xc = vx * c[i], ys = vy * s[i], yc = vy * c[i], xs = vx * s[i];
        const f16 lo = xc - ys, hi = yc + xs;
        r[i] = low ? lo : hi;
    }
    return r;
}

__device__ __forceinline__ float load_sc1(const float* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // served below the per-CU L1
}
__device__ __forceinline__ void store_sc1(float* p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
}

// grid (splits, heads, batch), 256 threads; see the file header.  The five leading pointers are preloaded into SGPRs at
// launch (-amdgpu-kernarg-preload-count): the three scalar reads the chunk bounds depend on go out with the first
// instructions, together with the fetch of the argument block, and nothing else stands before the first cache loads.
template <int D>
__global__ __launch_bounds__(kAttnThreads) void rope_attn_decode_kernel(const int64_t* __restrict__ kv_len,
                                                                        const int64_t* __restrict__ slots,
                                                                        const int64_t* __restrict__ positions,
                                                                        f16* __restrict__ kc, f16* __restrict__ vc,
                                                                        const RopeAttnArgs a)
{
    constexpr int LPP = D / 8, SETS = (kAttnThreads / 64) * (64 / LPP);
    __shared__ float    sm_m[SETS], sm_l[SETS];
    __shared__ __attribute__((aligned(16))) float sm_o[SETS * D];
    __shared__ unsigned sm_ticket;

    ATTN_STAMP(0);
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z, hk = h / a.groups;
    const int splits = gridDim.x, H = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, li = lane % LPP, d0 = li * 8;
    const int d1 = d0 < D / 2 ? d0 + D / 2 : d0 - D / 2;  // the paired channels of the rotation

    // three independent scalar reads, every one from a valid address (no branch before the loads): a missing `slots`
    // reads the position instead, a missing `kv_len` reads the position and ignores it
    // (issued as one batch with one wait: left to itself the compiler waits after each of them)
    const int64_t* p_pos  = positions + b;
    const int64_t* p_slot = slots ? slots + (long)b * a.slot_stride : positions + b;
    const int64_t* p_len  = kv_len ? kv_len : positions;
    int64_t        rpos, slot64, filled;
    asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dwordx2 %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(rpos), "=&s"(slot64), "=&s"(filled)
                 : "s"(p_pos), "s"(p_slot), "s"(p_len)
                 : "memory");
    ATTN_STAMP(1);
    // a slot outside the cache is neither written nor attended, and (like the two-launch form) nothing is rotated then
    const bool have_new = slot64 >= 0 && slot64 < a.S && rpos >= 0;
    const int  slot = have_new ? (int)slot64 : -1;
    // a full cache (or a bad position) used to be silent: count the dropped steps (eetq_decode_dropped_steps)
    if (!have_new && split == 0 && h == 0 && threadIdx.x == 0) atomicAdd(&g_attn_dropped, 1u);
    const int  Sv = kv_len ? max(0, (int)min((int64_t)a.S, filled + a.kv_len_bias)) : a.S;
    const int  chunk = (Sv + splits - 1) / splits;
    const int  j0 = split * chunk, j1 = min(Sv, j0 + chunk);

    // the cache rows of the first trip go in flight before anything else is touched
    const f16 *kbase = kc + b * a.kc_sb + hk * a.kc_sh + d0, *vbase = vc + b * a.vc_sb + hk * a.vc_sh + d0;
    KvTrip<D>  trip;
    load_trip<D>(trip, kbase, vbase, a.kc_ss, a.vc_ss, j0 + (tid >> 6) * (64 / LPP), j1);

    const f16* qp = a.q + b * a.q_sb + (long)h * D;
    const f16* kp = a.k + b * a.k_sb + (long)hk * D;
    f16x8      qv = *reinterpret_cast<const f16x8*>(qp + d0);
    f16x8      knew = *reinterpret_cast<const f16x8*>(kp + d0);
    const f16x8 vnew = *reinterpret_cast<const f16x8*>(a.v + b * a.v_sb + (long)hk * D + d0);
    if (have_new) {
        const f16x8 q2 = *reinterpret_cast<const f16x8*>(qp + d1), k2 = *reinterpret_cast<const f16x8*>(kp + d1);
        const f16*  cs = a.cos_sin + rpos * D;
        qv   = rope8<D>(qv, q2, cs, d0);
        knew = rope8<D>(knew, k2, cs, d0);
        // the cache row of the new token: once per kv head, by one position group of the head's first workgroup
        if (split == 0 && h == hk * a.groups && tid < LPP) {
            *reinterpret_cast<f16x8*>(kc + b * a.kc_sb + hk * a.kc_sh + (long)slot * a.kc_ss + d0) = knew;
            *reinterpret_cast<f16x8*>(vc + b * a.vc_sb + hk * a.vc_sh + (long)slot * a.vc_ss + d0) = vnew;
        }
    }
    ATTN_STAMP(2);
    float M, L, O;
    attn_chunk<D, true>(qv, a.scaling, trip, kbase, vbase, a.kc_ss, a.vc_ss, a.mask ? a.mask + b * a.m_sb : nullptr, j0, j1, slot,
                        knew, vnew, sm_m, sm_l, sm_o, M, L, O);

    ATTN_STAMP(3);
    float*     head_ws = a.ws + ((size_t)b * splits * H + h) * (D + kRecPad);  // [batch][split][head][record]
    const long rec_stride = (long)H * (D + kRecPad);
    if (splits > 1) {
        // ---- publish the chunk record (write-through), take a ticket; every storing wave drains its own stores ----
        if (tid < D) {
            float* rec = head_ws + (size_t)split * rec_stride;
            store_sc1(rec + kRecPad + tid, O);
            if (tid == 0) {
                store_sc1(rec, M);
                store_sc1(rec + 1, L);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ATTN_STAMP(4);
        if (tid == 0)
            sm_ticket = __hip_atomic_fetch_add(a.tickets + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        ATTN_STAMP(5);
        if (sm_ticket != (unsigned)(splits - 1)) return;  // not the head's last chunk
        if (tid == 0) __hip_atomic_store(a.tickets + b * H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- last arriver: merge all chunk records of the head (its own one read back like the others) ----
        const CoherentRecords recs{__builtin_amdgcn_make_buffer_rsrc(head_ws, 0, 0x7fffffff, 0x00020000)};
        O = attn_merge<D, kAttnThreads>(recs, rec_stride, splits, tid, sm_o);  // sm_o is free again: reused for the exchange
    } else if (tid < D) {
        O = L > 0.f ? O / L : 0.f;
    }
    ATTN_STAMP(6);
    if (tid < D) a.out[b * a.o_sb + h * a.o_sh + tid] = (f16)O;
    ATTN_STAMP(7);
    // ---- the last head to finish advances the token counter: by then every workgroup of the launch has read it ----
    if (a.advance && tid == 0) {
        const unsigned heads_total = (unsigned)(H * gridDim.z);
        unsigned*      done = a.tickets + heads_total;
        if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == heads_total - 1) {
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.advance += 1;
        }
    }
}

template <int D>
int launch_d(const f16* q, const f16* k, const f16* v, const f16* mask, f16* out, float* ws, int B, int H, int Hkv, int S,
             int splits, float scaling, const long* st, const int64_t* kv_len, int kv_len_bias, int64_t* advance,
             hipStream_t stream)
{
    attn_decode_partial_kernel<D><<<dim3(splits, H, B), kAttnThreads, 0, stream>>>(
        q, k, v, mask, ws, scaling, S, H / Hkv, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8],
        kv_len, kv_len_bias);
    EETQ_TRY_HIP(hipGetLastError());
    attn_decode_merge_kernel<D><<<dim3(H, B), kAttnThreads, 0, stream>>>(ws, out, splits, st[9], st[10], advance);
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

static unsigned long long* g_attn_stamps = nullptr;  // process-wide diagnostic hook (not thread-safe by design)
void set_attn_stamps(unsigned long long* buf) { g_attn_stamps = buf; }

int launch_rope_attn_decode(const int64_t* positions, const int64_t* slots, int slot_stride, const f16* q, const f16* k,
                            const f16* v, const f16* cos_sin, f16* kc, f16* vc, const f16* mask, f16* out, float* ws,
                            unsigned* tickets, int B, int H, int Hkv, int S, int D, int splits, float scaling,
                            const long* st, const int64_t* kv_len, int kv_len_bias, int64_t* advance, hipStream_t stream)
{
    EETQ_REQUIRE(positions && q && k && v && cos_sin && kc && vc && out && ws && tickets && st, "null pointer");
    EETQ_REQUIRE(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && S > 0 && splits > 0 && splits <= S && splits <= 4096,
                 "invalid attention shape");
    for (int i = 0; i < 9; ++i) EETQ_REQUIRE(st[i] % 8 == 0, "q / k / v / cache strides must be multiples of 8 elements (16-byte accesses)");
    EETQ_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)cos_sin) % 16 == 0,
                 "q, k, v, the caches and the cos|sin table must be 16-byte aligned");
    RopeAttnArgs a;
    a.slot_stride = slot_stride;
    a.q = q, a.k = k, a.v = v, a.q_sb = st[0], a.k_sb = st[1], a.v_sb = st[2];
    a.cos_sin = cos_sin;
    a.kc_sb = st[3], a.kc_sh = st[4], a.kc_ss = st[5], a.vc_sb = st[6], a.vc_sh = st[7], a.vc_ss = st[8];
    a.mask = mask, a.m_sb = st[9], a.out = out, a.o_sb = st[10], a.o_sh = st[11];
    a.ws = ws, a.tickets = tickets, a.kv_len_bias = kv_len_bias, a.advance = advance;
    a.S = S, a.groups = H / Hkv, a.scaling = scaling;
    a.stamps = g_attn_stamps;
    if (D == 128)
        rope_attn_decode_kernel<128><<<dim3(splits, H, B), kAttnThreads, 0, stream>>>(kv_len, slots, positions, kc, vc, a);
    else if (D == 64)
        rope_attn_decode_kernel<64><<<dim3(splits, H, B), kAttnThreads, 0, stream>>>(kv_len, slots, positions, kc, vc, a);
    else
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] decode attention supports head_dim 64 and 128");
    return check_hip(hipGetLastError(), "rope_attn_decode_kernel launch");
}

}  // namespace eetq

namespace eetq {
int attn_dropped_steps(unsigned* count, bool reset)
{
    EETQ_TRY_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_attn_dropped), sizeof(unsigned)));
    if (reset) {
        const unsigned zero = 0;
        EETQ_TRY_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dropped), &zero, sizeof(unsigned)));
    }
    return EETQ_OK;
}
}  // namespace eetq
