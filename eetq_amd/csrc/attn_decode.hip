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
//   phase 2  grid (heads, batch), D threads: merges the chunks, normalises, writes fp16.
// One-launch form (eetq_rope_decode_attention_f16), the decode step of a static cache: the same phase 1 with the NeoX
// rotation of the new token's q and k done in registers on the way in (the arithmetic of rotary_neox_kvcache_kernel, fp16
// with a rounding after every multiply and add), the rotated k and the v written to their cache row by one workgroup per
// kv head, every workgroup taking the new row from registers rather than from the cache (so nothing in the launch reads
// what the launch writes), and phase 2 done by whichever workgroup of a head finishes last: partials are published with
// write-through stores, a ticket per head decides "last" (the in-launch hand-off of gemm_splitk_kernel.hpp), and the last
// head to finish advances the cache's token counter.  Three launches and ~10 us per layer become one.  Both forms run the
// same chunk code and the same merge arithmetic: their outputs are bit-identical.
#include "common.hpp"

namespace eetq {

namespace {

constexpr int kAttnThreads = 256;

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

    constexpr int STEP = (kAttnThreads / 64) * PPW;
    for (int jb = j0 + wave * PPW;;) {
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
        jb += U * STEP;
        if (jb >= j1) break;  // wave-uniform
        load_trip<D>(t, kbase, vbase, k_ss, v_ss, jb, j1);
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
        float* out = ws + (((size_t)b * gridDim.y + h) * gridDim.x + split) * (D + 2);
        out[2 + tid] = O;
        if (tid == 0) {
            out[0] = M;
            out[1] = L;
        }
    }
}

// Merge of a head's `splits` chunk records p[s] = (m, l, o[D]) by a workgroup of NT threads (all of them call; thread
// d < D returns channel d); LOAD fetches one float.  sm_w: 2 * splits + NT/64 floats of LDS.  ONE memory phase for up to 32
// records: every (m, l) and the first 32 outputs of the thread's channel are requested before anything is waited for.
// A fully masked row yields zeros, not NaN.  Sums run in chunk order whatever NT is: the two launch forms give the same bits.
template <int D, int NT, typename Load>
__device__ __forceinline__ float attn_merge(const float* p, int splits, int d, float* sm_w, Load load)
{
#pragma clang fp contract(off)
    constexpr int R = 32;
    float* sm_l    = sm_w + splits;
    float* sm_part = sm_w + 2 * splits;
    const int dc   = d < D ? d : D - 1;  // threads beyond D only help with (m, l); their output loads are clamped duplicates
    float o[R];
#pragma unroll
    for (int i = 0; i < R; ++i) o[i] = load(p + min(i, splits - 1) * (D + 2) + 2 + dc);
    float M = -INFINITY;
    for (int s = d; s < splits; s += NT) {
        const float m1 = load(p + s * (D + 2)), l1 = load(p + s * (D + 2) + 1);
        sm_w[s] = m1;
        sm_l[s] = l1;
        M       = fmaxf(M, m1);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    if (NT > 64) {
        if ((d & 63) == 0) sm_part[d >> 6] = M;
        __syncthreads();
        M = sm_part[0];
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) M = fmaxf(M, sm_part[i]);
    }
    for (int s = d; s < splits; s += NT) sm_w[s] = M > -INFINITY ? __expf(sm_w[s] - M) : 0.f;  // the thread's own entries
    __syncthreads();
    if (d >= D) return 0.f;
    float L = 0.f, O = 0.f;
    for (int s = 0; s < splits; ++s) L = fmaf(sm_l[s], sm_w[s], L);
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (i < splits) O = fmaf(o[i], sm_w[i], O);
    for (int s = R; s < splits; ++s) O = fmaf(load(p + s * (D + 2) + 2 + d), sm_w[s], O);
    return L > 0.f ? O / L : 0.f;
}

// one workgroup per (head, batch): thread t < splits fetches that chunk's (m, l) in parallel; D threads then sum the chunk
// outputs with the loads of up to 8 chunks in flight
template <int D>
__global__ __launch_bounds__(D) void attn_decode_merge_kernel(const float* __restrict__ ws, f16* __restrict__ out,
                                                             int splits, long o_sb, long o_sh, int64_t* advance)
{
    extern __shared__ float sm_w[];  // attn_merge's scratch: 2 * splits + D/64 floats
    const int    h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* p = ws + ((size_t)b * gridDim.x + h) * splits * (D + 2);
    out[b * o_sb + h * o_sh + d] = (f16)attn_merge<D, D>(p, splits, d, sm_w, [](const float* a) { return *a; });
    // the cache's token counter (every reader of it in this step -- the cache-write launch and the partial kernel -- has
    // completed: they are earlier launches on the stream)
    if (advance && h == 0 && b == 0 && d == 0) *advance += 1;
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
    extern __shared__ float sm_w[];  // attn_merge's scratch: 2 * splits + one maximum per wave
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
    float* head_ws = a.ws + ((size_t)b * H + h) * splits * (D + 2);
    if (splits > 1) {
        // ---- publish the chunk record (write-through), take a ticket; every storing wave drains its own stores ----
        if (tid < D) {
            float* rec = head_ws + (size_t)split * (D + 2);
            store_sc1(rec + 2 + tid, O);
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
        O = attn_merge<D, kAttnThreads>(head_ws, splits, tid, sm_w, [](const float* p) { return load_sc1(p); });
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
    attn_decode_merge_kernel<D><<<dim3(H, B), D, (2 * splits + D / 64) * sizeof(float), stream>>>(ws, out, splits, st[9], st[10], advance);
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
    const size_t smem = (size_t)(2 * splits + kAttnThreads / 64) * sizeof(float);
    if (D == 128)
        rope_attn_decode_kernel<128><<<dim3(splits, H, B), kAttnThreads, smem, stream>>>(kv_len, slots, positions, kc, vc, a);
    else if (D == 64)
        rope_attn_decode_kernel<64><<<dim3(splits, H, B), kAttnThreads, smem, stream>>>(kv_len, slots, positions, kc, vc, a);
    else
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] decode attention supports head_dim 64 and 128");
    return check_hip(hipGetLastError(), "rope_attn_decode_kernel launch");
}

}  // namespace eetq
