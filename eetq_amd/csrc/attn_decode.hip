// Single-query ("decode") attention over a KV cache, fp16 in / fp16 out, fp32 softmax and accumulation.
//
// No counterpart in the reference's csrc: its EETLlamaAttention delegates the attention product to flash-attn
// (python/eetq/modules/llama_modules.py:131-143).  The stock library kernel this box offers for a single query token runs
// one workgroup per head (40 workgroups streaming a 25 MB cache: 68 us per layer at Llama-13B shapes), so the decode step
// of eet_accelerator's attention block uses this split-KV form instead: HBM/L2-bound byte work, no MFMA.
//
// Two-launch form (eetq_decode_attention_f16):
//   phase 1  grid (splits, heads, batch), 256 threads: a workgroup owns every splits-th 16-row block of the VALID cache
//            positions (round 6: "Work assignment" below) and has all of them in flight before it consumes any; a group of
//            D/8 lanes owns one position at a time (16-byte loads of its k and v rows, fully coalesced across the wave),
//            keeps an online-softmax state (m, l) and 8 output channels per lane; the groups of a wave are merged by lane
//            swaps, the waves through LDS, and the chunk's (m, l, o[D]) goes to an fp32 workspace;
//   phase 2  grid (heads, batch), 256 threads: merges the chunks (attn_merge), normalises, writes fp16.
// One-launch form (eetq_rope_decode_attention_f16), the decode step of a static cache: the same phase 1 with the NeoX
// rotation of the new token's q and k done in registers on the way in (the arithmetic of rotary_neox_kvcache_kernel, fp16
// with a rounding after every multiply and add), the rotated k and the v written to their cache row by one workgroup per
// kv head, every workgroup taking the new row from registers rather than from the cache (so nothing in the launch reads
// what the launch writes), and phase 2 done by whichever workgroup of a head finishes last: partials are published with
// write-through stores, a ticket per head decides "last" (the in-launch hand-off of gemm_splitk_kernel.hpp), and the last
// head to finish advances the cache's token counter.  Three launches (15.0 us per layer at 13B shapes, 1 k rows) became one
// (10.9 - 11.6 us, rounds 2-5; 9.5 us with the round-6 chunk code, of which 8.7 are there with an almost empty cache: launch
// gap 1.1, scalar reads 0.35, new token + rotation 0.8, the chunk's arithmetic 1.5, workgroup merge 0.4, publish 0.5, ticket
// 0.75, head merge 1.6, store 0.2 -- profiles/r06_attn_stamps.txt, r06_attn_bench.jsonl).  Both forms run the same chunk code
// and the same merge arithmetic: their outputs are bit-identical.
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

// Work assignment (round 6).  The cache rows of a (batch row, kv head) are cut into BLOCKS of BLK = 16 (D = 128) or 32 (D = 64)
// consecutive rows -- one wave-wide 16-byte load per wave covers its share of a block, the workgroup's four waves the whole
// block, 4 or 8 KiB of contiguous bytes -- and chunk `split` of `splits` owns blocks split, split + splits, split + 2 splits, ...:
// all chunks of a head carry the same number of rows to within one block whatever the valid length is.
// A workgroup requests its first kUA + kUB blocks before it consumes anything: at <= (kUA + kUB) * BLK * splits valid rows
// (1536 at 6 chunks, D = 128) every row of the launch is in flight at once and the launch pays the memory latency once
// (rounds 2-5: trips of 4 blocks one ahead -- 5.4 vs 4.7 us for the same bytes as a bare read kernel,
// profiles/r06_attn_probe.txt).  LONG launches (the cache can hold more rows than that) continue in software-pipelined trips
// of kUL blocks.
// Every cache load is an unconditional buffer load, nt (a row is read once per token), through descriptors that END AT THE
// VALID LENGTH: a row that does not count is out of range and comes back as zeros without touching memory -- no load behind a
// branch, no clamped duplicate rows, no select on the way in.  An out-of-range load is not free, though (0.5 us per launch for
// one per block, same probe): the mask loads exist only in the MASK instantiation, the further trips only in the LONG one.
// The running softmax state is rescaled once per trip (A, B, then each further trip), not once per position.
// Measured and shelved (tools/experiments/attn_loader_consumer.patch, profiles/r06_attn_forms.txt): eight waves per workgroup, four
// LOADERS moving the chunk into LDS by LDS-DMA and four consumers taking it from there trip by trip (bit-identical) -- the
// consumers start at 1.6 us instead of 3.4, but the rows arrive later than through registers: 9.9 - 10.7 us per step against 9.4.
constexpr int kUA = 8, kUB = 8, kUL = 4, kUG = 4;
constexpr int kNt = 2;  // aux bits of a buffer load: nt

template <int D>
struct AttnGeo {
    static constexpr int LPP = D / 8;                        // lanes per position
    static constexpr int PPW = 64 / LPP;                     // positions per wave instruction
    static constexpr int BLK = (kAttnThreads / 64) * PPW;    // positions per block
};

struct KvSrc {
    __amdgpu_buffer_rsrc_t k, v, m;   // the valid rows of one (batch row, kv head); m: the additive mask row (MASK only)
    unsigned               k_row, v_row;  // bytes between cache rows
    unsigned               d0b;           // this lane's channel offset in bytes
};

template <int U>
struct KvBlocks {
    f16x8          k[U], v[U];
    unsigned short m[U];
};

// blocks blk0 + u * dblk for u in [U0, U0 + UN) of the U a KvBlocks holds: this lane's row of each, K and V of a block back
// to back (returns are in order: a block is complete when its V row is)
template <int D, int U0, int UN, bool MASK, int U>
__device__ __forceinline__ void load_range(KvBlocks<U>& t, const KvSrc& s, int blk0, int dblk, int rib)
{
    constexpr int BLK = AttnGeo<D>::BLK;
    static_assert(U0 + UN <= U, "range");
#pragma unroll
    for (int u = U0; u < U0 + UN; ++u) {
        const unsigned j = (unsigned)((blk0 + u * dblk) * BLK + rib);
        t.k[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(s.k, (int)(j * s.k_row + s.d0b), 0, kNt));
        t.v[u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(s.v, (int)(j * s.v_row + s.d0b), 0, kNt));
        if constexpr (MASK) t.m[u] = __builtin_amdgcn_raw_buffer_load_b16(s.m, (int)(j * 2u), 0, 0);
    }
}
template <int D, int U, bool MASK>
__device__ __forceinline__ void load_blocks(KvBlocks<U>& t, const KvSrc& s, int blk0, int dblk, int rib)
{
    load_range<D, 0, U, MASK>(t, s, blk0, dblk, rib);
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

// One trip: the online-softmax update of this lane's running state (m, l, o[8]) with positions [U0, U0 + UN) of the U it holds.
// scores are scaling * (q . k) (+ mask) with the products summed in fp32 (v_dot2_f32_f16); rows >= Sv count for nothing (they
// were out of range of the descriptors: k = v = 0).  SUBST: position `slot` is taken from registers (knew, vnew: this lane's 8
// channels of the new token) instead of the cache -- tested per block (workgroup-uniform), selected per lane inside it.
template <int D, int U0, int UN, bool SUBST, bool MASK, int U>
__device__ __forceinline__ void consume_range(KvBlocks<U>& t, int blk0, int dblk, int rib, int Sv, int slot, const f16x8& knew,
                                              const f16x8& vnew, const f16x2 (&q2)[4], float scaling, float& m, float& l,
                                              float (&o)[8])
{
    // every multiply-add below is explicit and contraction is off: the two launch forms instantiate this code separately
    // and must round identically
#pragma clang fp contract(off)
    static_assert(U0 + UN <= U, "range");
    constexpr int LPP = AttnGeo<D>::LPP, BLK = AttnGeo<D>::BLK;
    if ((blk0 + U0 * dblk) * BLK >= Sv) return;  // workgroup-uniform: the trip's first block is beyond the valid rows, so are the others
    const int     slot_blk = SUBST && slot >= 0 ? slot / BLK : -1;
    float         sc[U];
#pragma unroll
    for (int u = U0; u < U0 + UN; ++u) {
        const int blk = blk0 + u * dblk, j = blk * BLK + rib;
        if (SUBST && blk == slot_blk) {  // workgroup-uniform
            const bool mine = j == slot;
            t.k[u] = mine ? knew : t.k[u];
            t.v[u] = mine ? vnew : t.v[u];
        }
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) a = __builtin_amdgcn_fdot2(f16x2{t.k[u][2 * i], t.k[u][2 * i + 1]}, q2[i], a, false);
        a = group_sum<LPP>(a) * scaling;
        if constexpr (MASK) a += (float)__builtin_bit_cast(f16, t.m[u]);
        sc[u] = j < Sv ? a : -INFINITY;
    }
    float mn = m;
#pragma unroll
    for (int u = U0; u < U0 + UN; ++u) mn = fmaxf(mn, sc[u]);
    if (mn > -INFINITY) {  // group-uniform
        const float keep = __expf(m - mn);
        l *= keep;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] *= keep;
#pragma unroll
        for (int u = U0; u < U0 + UN; ++u) {
            const float p = __expf(sc[u] - mn);  // exp(-inf) = 0 for the positions that do not count
            l += p;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf(p, (float)t.v[u][i], o[i]);
        }
        m = mn;
    }
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// value of lane ^ OFF (OFF = 8, 16, 32)
template <int OFF>
__device__ __forceinline__ float lane_xor(float v)
{
    const u32 u = __builtin_bit_cast(u32, v);
    if constexpr (OFF == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __builtin_bit_cast(float, (u32)((threadIdx.x & 32) ? r[0] : r[1]));  // r[0]: the lower half everywhere, r[1]: the upper
    } else if constexpr (OFF == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return __builtin_bit_cast(float, (u32)((threadIdx.x & 16) ? r[0] : r[1]));  // r[0]: the even 16-lane rows everywhere, r[1]: the odd
    } else {
        return __shfl_xor(v, OFF, 64);
    }
}

// The position groups of a wave (lane swaps, no LDS), then the waves through LDS (NW sets): (m, l, o[8]) per lane -> threads
// tid < D hold (M, L, O) of the chunk.  One barrier; only the consuming waves (kAttnThreads threads) call it.
template <int D>
__device__ __forceinline__ void chunk_merge(float m, float l, const float (&o)[8], float* sm_m, float* sm_l, float* sm_o, float& M,
                                            float& L, float& O)
{
#pragma clang fp contract(off)
    constexpr int LPP = AttnGeo<D>::LPP, PPW = AttnGeo<D>::PPW, NW = kAttnThreads / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane % LPP, d0 = li * 8;
    float mw = m;
    if constexpr (PPW == 8) mw = fmaxf(mw, lane_xor<8>(mw));
    mw = fmaxf(mw, lane_xor<16>(mw));
    mw = fmaxf(mw, lane_xor<32>(mw));
    const float w = mw > -INFINITY ? __expf(m - mw) : 0.f;  // exp(-inf) = 0 for a group without a valid position
    float       part[9];
    part[8] = l * w;
#pragma unroll
    for (int i = 0; i < 8; ++i) part[i] = o[i] * w;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if constexpr (PPW == 8) part[i] += lane_xor<8>(part[i]);
        part[i] = sum_xor32(sum_xor16(part[i]));
    }
    if (lane < LPP) {
        if (li == 0) {
            sm_m[wave] = mw;
            sm_l[wave] = part[8];
        }
        *reinterpret_cast<f32x4*>(sm_o + wave * D + d0)     = f32x4{part[0], part[1], part[2], part[3]};
        *reinterpret_cast<f32x4*>(sm_o + wave * D + d0 + 4) = f32x4{part[4], part[5], part[6], part[7]};
    }
    __syncthreads();
    M = -INFINITY, L = 0.f, O = 0.f;
    if (tid < D) {
#pragma unroll
        for (int s2 = 0; s2 < NW; ++s2) M = fmaxf(M, sm_m[s2]);
        if (M > -INFINITY) {
#pragma unroll
            for (int s2 = 0; s2 < NW; ++s2) {
                const float ws = __expf(sm_m[s2] - M);  // exp(-inf) = 0 for empty sets
                L = fmaf(sm_l[s2], ws, L);
                O = fmaf(sm_o[s2 * D + tid], ws, O);
            }
        }
    }
}

// The consumption of chunk `split` of one (batch row, head) and its merge across the workgroup.  ta: the chunk's first kUA
// blocks, already requested (load_blocks at blk0 = split, stride splits); tb: registers for the next kUB.  On return
// threads tid < D hold (M, L, O) = the chunk's running maximum, its sum of exp(s - M) and channel tid of sum exp(s - M) v.
// qv: this lane's 8 channels of the (rotated) query.
template <int D, bool SUBST, bool MASK, bool LONG>
__device__ __forceinline__ void attn_chunk(const f16x8& qv, float scaling, KvBlocks<kUA>& ta, KvBlocks<kUB>& tb, const KvSrc& src,
                                           int split, int splits, int Sv, int slot, const f16x8& knew, const f16x8& vnew,
                                           float* sm_m, float* sm_l, float* sm_o, float& M, float& L, float& O)
{
#pragma clang fp contract(off)
    constexpr int LPP = AttnGeo<D>::LPP, PPW = AttnGeo<D>::PPW, BLK = AttnGeo<D>::BLK;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = lane / LPP, rib = wave * PPW + grp;
    const f16x2 q2[4] = {{qv[0], qv[1]}, {qv[2], qv[3]}, {qv[4], qv[5]}, {qv[6], qv[7]}};

    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;

    // A is on its way (requested by the caller); B is requested here, kUG blocks at a time, between the trips of A: a wave that
    // asks for more than the memory system takes at once is held AT THE REQUEST until there is room, and a wave held there
    // consumes nothing -- with all of B requested up front the first trip of A was consumed 4.2 us into the launch although its
    // rows had landed well before (profiles/r06_attn_stamps.txt).  Trips are kUG blocks each.
    static_assert(kUA == kUB && kUA % kUG == 0, "trip structure");
    const int blk_b = split + kUA * splits;
    auto trips_ab = [&](auto i_tag) {
        constexpr int I = decltype(i_tag)::value;
        load_range<D, I * kUG, kUG, MASK>(tb, src, blk_b, splits, rib);
        __builtin_amdgcn_sched_barrier(0);
        consume_range<D, I * kUG, kUG, SUBST, MASK>(ta, split, splits, rib, Sv, slot, knew, vnew, q2, scaling, m, l, o);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto trips_b = [&](auto i_tag) {
        constexpr int I = decltype(i_tag)::value;
        consume_range<D, I * kUG, kUG, SUBST, MASK>(tb, blk_b, splits, rib, Sv, slot, knew, vnew, q2, scaling, m, l, o);
    };
    constexpr int NT = kUA / kUG;  // trips per batch
    static_for<NT>(trips_ab);
    if constexpr (LONG) {
        // further trips of kUL blocks, software-pipelined one ahead; the first is requested before B is consumed
        const int     nblk = (Sv + BLK - 1) / BLK;
        int           ub   = kUA + kUB;
        KvBlocks<kUL> tl, tn;
        load_blocks<D, kUL, MASK>(tl, src, split + ub * splits, splits, rib);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NT>(trips_b);
        while (split + ub * splits < nblk) {  // workgroup-uniform
            load_blocks<D, kUL, MASK>(tn, src, split + (ub + kUL) * splits, splits, rib);
            consume_range<D, 0, kUL, SUBST, MASK>(tl, split + ub * splits, splits, rib, Sv, slot, knew, vnew, q2, scaling, m, l, o);
            tl = tn;
            ub += kUL;
        }
    } else {
        static_for<NT>(trips_b);
    }

    chunk_merge<D>(m, l, o, sm_m, sm_l, sm_o, M, L, O);
}

// the buffer descriptors of one (batch row, kv head), ending at the valid length Sv
template <int D, bool MASK>
__device__ __forceinline__ KvSrc make_kv_src(const f16* kbase, const f16* vbase, long k_ss, long v_ss, int Sv, const f16* mrow)
{
    KvSrc s;
    s.k_row = (unsigned)(k_ss * 2);
    s.v_row = (unsigned)(v_ss * 2);
    s.d0b   = (unsigned)(((threadIdx.x & 63) % AttnGeo<D>::LPP) * 16);
    s.k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kbase), 0, (int)((unsigned)Sv * s.k_row), 0x00020000);
    s.v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vbase), 0, (int)((unsigned)Sv * s.v_row), 0x00020000);
    s.m = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(MASK ? mrow : kbase), 0, MASK ? Sv * 2 : 0, 0x00020000);
    return s;
}

// rows at and beyond the valid length of a pre-allocated (static) cache hold zeros or stale tokens: never attended
__device__ __forceinline__ int valid_len(int S, const int64_t* kv_len, int kv_len_bias)
{
    return kv_len ? max(0, (int)min((int64_t)S, *kv_len + kv_len_bias)) : S;
}

template <int D, bool MASK, bool LONG>
__global__ __launch_bounds__(kAttnThreads) void attn_decode_partial_kernel(
    const f16* __restrict__ q, const f16* __restrict__ kc, const f16* __restrict__ vc, const f16* __restrict__ mask,
    float* __restrict__ ws, float scaling, int S, int groups, long q_sb, long q_sh, long k_sb, long k_sh,
    long k_ss, long v_sb, long v_sh, long v_ss, long m_sb, const int64_t* __restrict__ kv_len, int kv_len_bias)
{
    constexpr int NW = kAttnThreads / 64;
    __shared__ float sm_m[NW], sm_l[NW];
    __shared__ __attribute__((aligned(16))) float sm_o[NW * D];

    const int split = blockIdx.x, splits = gridDim.x, h = blockIdx.y, b = blockIdx.z, hk = h / groups;
    const int tid = threadIdx.x, d0 = ((tid & 63) % (D / 8)) * 8;
    const int rib = (tid >> 6) * AttnGeo<D>::PPW + (tid & 63) / AttnGeo<D>::LPP;
    const int Sv = valid_len(S, kv_len, kv_len_bias);

    const f16x8 qv  = *reinterpret_cast<const f16x8*>(q + b * q_sb + h * q_sh + d0);
    const KvSrc src = make_kv_src<D, MASK>(kc + b * k_sb + hk * k_sh, vc + b * v_sb + hk * v_sh, k_ss, v_ss, Sv,
                                           MASK ? mask + b * m_sb : nullptr);
    KvBlocks<kUA> ta;
    KvBlocks<kUB> tb;
    load_blocks<D, kUA, MASK>(ta, src, split, splits, rib);
    float       M, L, O;
    const f16x8 none = {};
    attn_chunk<D, false, MASK, LONG>(qv, scaling, ta, tb, src, split, splits, Sv, -1, none, none, sm_m, sm_l, sm_o, M, L, O);
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
// own / other: the lane's channels and the paired ones; c / s: the lane's 8 values of the position's cos | sin row.  fp16 arithmetic, one rounding per
// multiply and add, exactly rotary_neox_kvcache_kernel (norm_rope.hip).
template <int D>
__device__ __forceinline__ f16x8 rope8(const f16x8& own, const f16x8& other, const f16x8& c, const f16x8& s, int d0)
{
#pragma clang fp contract(off)
    constexpr int embed = D / 2;
    const bool    low   = d0 < embed;
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

// Publish the chunk record, take the head's ticket; the last arriver merges the head's chunks and stores the output, and the last
// head advances the cache's token counter.  Threads tid < D hold the chunk's (M, L, O); kAttnThreads threads call (barriers inside).
template <int D>
__device__ __forceinline__ void head_handoff(const RopeAttnArgs& a, int split, int splits, int h, int b, int H, float M, float L,
                                             float O, float* sm_o, unsigned* sm_ticket)
{
    const int tid = threadIdx.x;
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
            *sm_ticket = __hip_atomic_fetch_add(a.tickets + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        ATTN_STAMP(5);
        if (*sm_ticket != (unsigned)(splits - 1)) return;  // not the head's last chunk
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

// grid (splits, heads, batch), 256 threads; see the file header.  The five leading pointers are preloaded into SGPRs at
// launch (-amdgpu-kernarg-preload-count): the three scalar reads the chunk bounds depend on go out with the first
// instructions, together with the fetch of the argument block, and nothing else stands before the first cache loads.
template <int D, bool MASK, bool LONG>
__global__ __launch_bounds__(kAttnThreads) void rope_attn_decode_kernel(const int64_t* __restrict__ kv_len,
                                                                        const int64_t* __restrict__ slots,
                                                                        const int64_t* __restrict__ positions,
                                                                        f16* __restrict__ kc, f16* __restrict__ vc,
                                                                        const RopeAttnArgs a)
{
    constexpr int LPP = D / 8, NW = kAttnThreads / 64;
    __shared__ float    sm_m[NW], sm_l[NW];
    __shared__ __attribute__((aligned(16))) float sm_o[NW * D];
    __shared__ unsigned sm_ticket;

    ATTN_STAMP(0);
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z, hk = h / a.groups;
    const int splits = gridDim.x, H = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, li = lane % LPP, d0 = li * 8;
    const int d1 = d0 < D / 2 ? d0 + D / 2 : d0 - D / 2;  // the paired channels of the rotation

    // three independent scalar reads, every one from a valid address (no branch before the loads): a missing `slots`
    // reads the position instead, a missing `kv_len` reads the position and ignores it
    // (issued as one batch with one wait: left to itself the compiler waits after each of them).  They go out with the first
    // instructions, on a quiet memory system (0.3 - 0.6 us): behind a burst of cache loads the same reads take 1.2 us and
    // hold back everything that depends on the valid length (measured: profiles/r06_attn_stamps.txt).
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

    // Memory queue order (a wave's loads return in order): the new token and its cos | sin row first, then the chunk's first
    // kUA blocks; the rotation runs on the former while the latter are on their way; attn_chunk requests the rest.
    const f16*  qp = a.q + b * a.q_sb + (long)h * D;
    const f16*  kp = a.k + b * a.k_sb + (long)hk * D;
    f16x8       qv   = *reinterpret_cast<const f16x8*>(qp + d0);
    f16x8       knew = *reinterpret_cast<const f16x8*>(kp + d0);
    const f16x8 vnew = *reinterpret_cast<const f16x8*>(a.v + b * a.v_sb + (long)hk * D + d0);
    const f16x8 qpair = *reinterpret_cast<const f16x8*>(qp + d1), kpair = *reinterpret_cast<const f16x8*>(kp + d1);
    // (a bad position reads row 0 of the table and is not used)
    const f16*  cs = a.cos_sin + (have_new ? rpos : 0) * D + (d0 < D / 2 ? d0 : d0 - D / 2);
    const f16x8 rc = *reinterpret_cast<const f16x8*>(cs), rs = *reinterpret_cast<const f16x8*>(cs + D / 2);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler sinks these seven loads below the block loads otherwise)
    const KvSrc src = make_kv_src<D, MASK>(kc + b * a.kc_sb + hk * a.kc_sh, vc + b * a.vc_sb + hk * a.vc_sh, a.kc_ss, a.vc_ss,
                                           Sv, MASK ? a.mask + b * a.m_sb : nullptr);
    const int   rib = (tid >> 6) * AttnGeo<D>::PPW + lane / LPP;
    KvBlocks<kUA> ta;
    KvBlocks<kUB> tb;
    load_blocks<D, kUA, MASK>(ta, src, split, splits, rib);
    __builtin_amdgcn_sched_barrier(0);
    if (have_new) {
        qv   = rope8<D>(qv, qpair, rc, rs, d0);
        knew = rope8<D>(knew, kpair, rc, rs, d0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the cache row of the new token: once per kv head, by one position group of the head's first workgroup
    if (have_new && split == 0 && h == hk * a.groups && tid < LPP) {
        *reinterpret_cast<f16x8*>(kc + b * a.kc_sb + hk * a.kc_sh + (long)slot * a.kc_ss + d0) = knew;
        *reinterpret_cast<f16x8*>(vc + b * a.vc_sb + hk * a.vc_sh + (long)slot * a.vc_ss + d0) = vnew;
    }
    ATTN_STAMP(2);
    float M, L, O;
    attn_chunk<D, true, MASK, LONG>(qv, a.scaling, ta, tb, src, split, splits, Sv, slot, knew, vnew, sm_m, sm_l, sm_o, M, L, O);

    ATTN_STAMP(3);
    head_handoff<D>(a, split, splits, h, b, H, M, L, O, sm_o, &sm_ticket);
}

// LONG: some chunk may own more than kUA + kUB blocks (a launch-time fact of the capacity S and the chunk count)
template <int D>
bool long_chunks(int S, int splits)
{
    constexpr int BLK = AttnGeo<D>::BLK;
    return ((S + BLK - 1) / BLK + splits - 1) / splits > kUA + kUB;
}

template <int D, bool MASK, bool LONG>
int launch_d2(const f16* q, const f16* k, const f16* v, const f16* mask, f16* out, float* ws, int B, int H, int Hkv, int S,
              int splits, float scaling, const long* st, const int64_t* kv_len, int kv_len_bias, int64_t* advance,
              hipStream_t stream)
{
    launch_kernel(attn_decode_partial_kernel<D, MASK, LONG>, dim3(splits, H, B), dim3(kAttnThreads), 0, stream, q, k, v, mask, ws,
                  scaling, S, H / Hkv, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], kv_len, kv_len_bias);
    EETQ_TRY_HIP(hipGetLastError());
    launch_kernel(attn_decode_merge_kernel<D>, dim3(H, B), dim3(kAttnThreads), 0, stream, (const float*)ws, out, splits, st[9],
                  st[10], advance);
    return check_hip(hipGetLastError(), "attn_decode kernels launch");
}

template <int D>
int launch_d(const f16* q, const f16* k, const f16* v, const f16* mask, f16* out, float* ws, int B, int H, int Hkv, int S,
             int splits, float scaling, const long* st, const int64_t* kv_len, int kv_len_bias, int64_t* advance,
             hipStream_t stream)
{
    const bool lng = long_chunks<D>(S, splits);
#define EETQ_ATTN_GO(MASK, LONG) \
    return launch_d2<D, MASK, LONG>(q, k, v, mask, out, ws, B, H, Hkv, S, splits, scaling, st, kv_len, kv_len_bias, advance, stream)
    if (mask) {
        if (lng) EETQ_ATTN_GO(true, true);
        EETQ_ATTN_GO(true, false);
    }
    if (lng) EETQ_ATTN_GO(false, true);
    EETQ_ATTN_GO(false, false);
#undef EETQ_ATTN_GO
}

template <int D>
int launch_rope_d(const int64_t* kv_len, const int64_t* slots, const int64_t* positions, f16* kc, f16* vc, const RopeAttnArgs& a,
                  dim3 grid, hipStream_t stream)
{
    const bool lng = long_chunks<D>(a.S, (int)grid.x);
#define EETQ_ATTN_GO(MASK, LONG) \
    launch_kernel(rope_attn_decode_kernel<D, MASK, LONG>, grid, dim3(kAttnThreads), 0, stream, kv_len, slots, positions, kc, vc, a)
    if (a.mask) {
        if (lng) EETQ_ATTN_GO(true, true);
        else EETQ_ATTN_GO(true, false);
    } else {
        if (lng) EETQ_ATTN_GO(false, true);
        else EETQ_ATTN_GO(false, false);
    }
#undef EETQ_ATTN_GO
    return check_hip(hipGetLastError(), "rope_attn_decode_kernel launch");
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
    EETQ_REQUIRE(strides[4] > 0 && strides[7] > 0 && ((long)S + 4096) * strides[4] * 2 < (1L << 31) &&
                     ((long)S + 4096) * strides[7] * 2 < (1L << 31),
                 "one head's cache rows must span less than 2 GiB");
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
    EETQ_REQUIRE(st[5] > 0 && st[8] > 0 && ((long)S + 4096) * st[5] * 2 < (1L << 31) && ((long)S + 4096) * st[8] * 2 < (1L << 31),
                 "one head's cache rows must span less than 2 GiB");
    RopeAttnArgs a;
    a.slot_stride = slot_stride;
    a.q = q, a.k = k, a.v = v, a.q_sb = st[0], a.k_sb = st[1], a.v_sb = st[2];
    a.cos_sin = cos_sin;
    a.kc_sb = st[3], a.kc_sh = st[4], a.kc_ss = st[5], a.vc_sb = st[6], a.vc_sh = st[7], a.vc_ss = st[8];
    a.mask = mask, a.m_sb = st[9], a.out = out, a.o_sb = st[10], a.o_sh = st[11];
    a.ws = ws, a.tickets = tickets, a.kv_len_bias = kv_len_bias, a.advance = advance;
    a.S = S, a.groups = H / Hkv, a.scaling = scaling;
    a.stamps = g_attn_stamps;
    if (D == 128) return launch_rope_d<128>(kv_len, slots, positions, kc, vc, a, dim3(splits, H, B), stream);
    if (D == 64) return launch_rope_d<64>(kv_len, slots, positions, kc, vc, a, dim3(splits, H, B), stream);
    return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] decode attention supports head_dim 64 and 128");
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
