// Causal attention over a prompt ("prefill"), fp16 in / fp16 out, fp32 scores, softmax state and accumulation, on MFMA.
//
// No counterpart in the reference's csrc: its EETLlamaAttention hands the attention product to flash-attn
// (python/eetq/modules/llama_modules.py:131-143).  The library kernel torch offers here runs a 1 024-token prompt of 40 heads x 128
// at 76 us per layer (0.14 PFLOP/s of causal work); this is the flash form written for this path: the rotated query rows of the
// fused QKV projection (strided [batch][token][head][D]) against the rows a static KV cache has just been given
// ([batch][kv head][row][D], rows 0 .. keys - 1), query t attending rows <= t + causal_offset.
//
// Workgroup = 4 waves = 128 query rows of one (batch row, head), 32 rows per wave; keys in blocks of 64.
//   S^T = K Q^T   (v_mfma_f32_32x32x16_f16, K rows as the A operand, the wave's Q rows as B: a lane owns ONE query column and 32 of
//                  the block's 64 keys, its partner lane ^ 32 the others -- the row maximum and sum are 31 local operations and one
//                  lane swap; the probabilities, rounded to fp16, ARE the B operand of the next product, no data movement)
//   O^T += V^T P^T (V^T rows = channels as the A operand: V is transposed on its way into LDS -- a thread holds one 16-byte chunk of four
//                  consecutive rows and stores four keys of one channel at a time into a [D][64 keys] image with a 136-byte pitch --
//                  so that a lane's 8 k-slots are two 8-byte reads; the k-slot <-> key assignment
//                  follows what the S^T accumulator layout hands out: slots 0..3 = keys base + 4 hi + j, 4..7 = base + 8 + 4 hi + j)
// K rows sit in LDS as 256-byte rows with their 16-byte chunks XOR-swizzled by (row & 15) (conflict-free 16-byte fragment reads).
// Two LDS buffers, one barrier per block: block j + 1 goes from registers into the free buffer at the top of block j, block j + 2 is
// fetched then and has a whole block of arithmetic to land.  (Issuing block j + 1's score product between block j's softmax and its
// second product -- two barriers per block -- measured the same: the waves of the two resident workgroups run their phases in step.)  Two workgroups per CU (<= 256 registers: everything in the
// architectural file, no accumulator copies).    Blocks beyond a wave's last query row are skipped by that wave; heavy query blocks are
// dispatched first (1-D grid ordered by weight).
#include <type_traits>

#include "common.hpp"

namespace eetq {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kPfThreads = 256, kPfBM = 128, kPfBN = 64;

struct PrefillArgs {
    const f16* q;
    const f16* k;
    const f16* v;
    f16*       out;
    long       q_sb, q_st, q_sh, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_st, o_sh;
    int        Tq, Tk, koff, groups, H, B, nqb;
    float      c;   // scaling * log2(e)
};

template <int D>
__global__ __launch_bounds__(kPfThreads, 2) void prefill_attn_kernel(const PrefillArgs a)
{
    static_assert(D == 128 || D == 64, "head_dim 128 or 64");
    constexpr int KS = D / 16;            // k-steps of the score product
    constexpr int DT = D / 32;            // 32-channel blocks of the output
    constexpr int VT_PITCH = kPfBN + 4;   // halfs per channel row of the transposed V image (136 bytes)
    constexpr int CH = D / 8;             // 16-byte chunks per K / V row
    constexpr int NLD = kPfBN * CH / kPfThreads;  // 16-byte loads per thread and tile
    // two LDS buffers (2 x 33 KiB): block j is read from buffer j & 1 while block j + 1 -- fetched into registers during block j - 1 -- is
    // written into the other one at the top of block j and block j + 2 is fetched: global loads have a whole block of arithmetic to
    // land, one barrier per block
    __shared__ __attribute__((aligned(16))) f16 k_lds2[2][kPfBN * D];
    __shared__ __attribute__((aligned(16))) f16 vt_lds2[2][D * VT_PITCH];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, ln = lane & 31, hi = lane >> 5;
    // (wave in a scalar register: the per-wave tests below -- block skipped, block on the diagonal -- must be branches, not selects)
    // 1-D grid, heaviest query blocks first: id -> (query block rank, head, batch row).  With H * B a multiple of 8 all query blocks
    // of a head land on one XCD (its K / V rows are shared in that L2), and a CU's second workgroup is a light one
    const int hb = a.H * a.B, qb = a.nqb - 1 - (int)blockIdx.x / hb, rest = (int)blockIdx.x % hb;
    const int h = rest % a.H, b = rest / a.H, hk = h / a.groups;
    const int q0 = qb * kPfBM, qrow = q0 + wave * 32 + ln;           // this lane's query row
    const int qlast_wave = q0 + wave * 32 + 31;                       // the wave's last row
    // keys this workgroup needs: rows <= its last query + koff
    const int kend  = min(a.Tk, q0 + kPfBM + a.koff);
    const int nblk  = kend > 0 ? (kend + kPfBN - 1) / kPfBN : 0;
    // the valid rows of this (batch row, kv head) behind buffer descriptors: rows at and beyond Tk are out of range and read as zeros
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.k + b * a.k_sb + hk * a.k_sh), 0,
                                                                         (int)((unsigned)a.Tk * (unsigned)a.k_ss * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.v + b * a.v_sb + hk * a.v_sh), 0,
                                                                         (int)((unsigned)a.Tk * (unsigned)a.v_ss * 2u), 0x00020000);

    // ---- the wave's query rows as B fragments: lane (ln, hi) holds channels 16 ks + 8 hi .. + 7 of row ln ----
    f16x8 qf[KS];
    {
        const f16* qp = a.q + b * a.q_sb + (long)min(qrow, a.Tq - 1) * a.q_st + h * a.q_sh + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const f16x8*>(qp + 16 * ks);
    }

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging registers: thread t holds 16-byte chunk t % CH of the NLD consecutive rows NLD (t / CH) .. of a tile (four rows at D = 128,
    // two at D = 64) -- so that the transposed V image is written as 8- / 4-byte pieces (NLD keys of one channel: 8 LDS writes per
    // thread and tile; 2-byte writes, 64 of them, kept the LDS pipe busy for ~4 000 cycles per tile and workgroup)
    static_assert((NLD == 4 && CH == 16) || (NLD == 2 && CH == 8), "rows per thread");
    u32x4 kreg[NLD], vreg[NLD];
    const int c16 = tid % CH, p4 = NLD * (tid / CH);
    const int krow_b = (int)(a.k_ss * 2), vrow_b = (int)(a.v_ss * 2);
    const int kvo = p4 * krow_b + c16 * 16, vvo = p4 * vrow_b + c16 * 16;  // constant; tile and row offsets ride in the scalar operand
    auto fetch = [&](int j) {
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            kreg[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo, (j * kPfBN + r) * krow_b, 0));
            vreg[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vvo, (j * kPfBN + r) * vrow_b, 0));
        }
    };
    // K rows: 16-byte chunks XOR-swizzled by (row & (CH - 1)).  V^T image: channel row d, key column key ^ (((d >> 5) & 3) << 3): a
    // wave's stores land in different banks.
    auto stage = [&](int buf) {
        f16* kl = k_lds2[buf];
        f16* vl = vt_lds2[buf];
#pragma unroll
        for (int r = 0; r < NLD; ++r) *reinterpret_cast<u32x4*>(kl + (p4 + r) * D + 8 * (c16 ^ ((p4 + r) & (CH - 1)))) = kreg[r];
        f16* vp = vl + 8 * c16 * VT_PITCH + (p4 ^ (((c16 >> 2) & 3) << 3));
        const f16x8 v0 = __builtin_bit_cast(f16x8, vreg[0]), v1 = __builtin_bit_cast(f16x8, vreg[1]);
        if constexpr (NLD == 4) {
            const f16x8 v2 = __builtin_bit_cast(f16x8, vreg[2]), v3 = __builtin_bit_cast(f16x8, vreg[3]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f16x2 lo = {v0[e], v1[e]}, hh = {v2[e], v3[e]};
                *reinterpret_cast<u32x2*>(vp + e * VT_PITCH) = u32x2{as_u32(lo), as_u32(hh)};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f16x2 lo = {v0[e], v1[e]};
                *reinterpret_cast<u32*>(vp + e * VT_PITCH) = as_u32(lo);
            }
        }
    };

    if (nblk > 0) {
        fetch(0);
        stage(0);
        if (nblk > 1) fetch(1);
    }
    __syncthreads();

    for (int j = 0; j < nblk; ++j) {
        const int  key0   = j * kPfBN;
        const bool active = key0 <= qlast_wave + a.koff;  // wave-uniform: some row of this wave attends this block
        if (j + 1 < nblk) stage((j + 1) & 1);             // block j + 1: registers -> the buffer block j - 1 was read from
        if (j + 2 < nblk) fetch(j + 2);
        const f16* k_lds  = k_lds2[j & 1];
        const f16* vt_lds = vt_lds2[j & 1];
        if (active) {
            // ---- S^T = K Q^T: two 32-key halves (alternating: back-to-back MFMAs on one accumulator wait out each other's latency) ----
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const int   krow = 32 * kb + ln;
                    const f16x8 kf = *reinterpret_cast<const f16x8*>(k_lds + krow * D + 8 * ((2 * ks + hi) ^ (krow & (CH - 1))));
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
                }
            // ---- mask (only where the block touches the wave's diagonal or the end of the keys: a wave-uniform branch), running maximum
            // (scaled domain) ----
            const bool edge = key0 + kPfBN - 1 > q0 + wave * 32 + a.koff || key0 + kPfBN > a.Tk;
            float mloc = -INFINITY;
            if (edge) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int  key   = key0 + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool valid = key < a.Tk && key <= qrow + a.koff;
                        float      t     = valid ? s[kb][r] : -INFINITY;
                        asm volatile("" : "+v"(t));  // pins the select inside the branch (else: 170 select instructions on EVERY block)
                        s[kb][r] = t;
                    }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kb][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * a.c;              // (c > 0)
            const float m_new = fmaxf(m_run, mloc);
            const float m_use = m_new > -INFINITY ? m_new : 0.f;             // a row with nothing to attend yet: every p = 0
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);       // exp2(-inf) = 0 for the first block
            float       psum  = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], a.c, -m_use));
                    psum += s[kb][r];
                }
            psum += __shfl_xor(psum, 32, 64);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            // ---- O^T += V^T P^T: per 32-key half two 16-key steps; the lane's own accumulator values are its B operand ----
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    f16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (f16)s[kb][8 * st + e];
                    const int kcol = 32 * kb + 16 * st + 4 * hi;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const f16*  vrow = vt_lds + (32 * dt + ln) * VT_PITCH;
                        const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow + (kcol ^ (dt << 3)));
                        const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + ((kcol + 8) ^ (dt << 3)));
                        const f16x8 vf = __builtin_bit_cast(f16x8, u32x4{v0.x, v0.y, v1.x, v1.y});
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dt], 0, 0, 0);
                    }
                }
        }
        __syncthreads();       // block j has been read by every wave; block j + 1 is in the other buffer
    }

    // ---- normalise and store: lane (ln, hi) holds row `qrow`, channels 32 dt + 8 g + 4 hi .. + 3 ----
    if (qrow < a.Tq) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        f16*        op  = a.out + b * a.o_sb + (long)qrow * a.o_st + h * a.o_sh + 4 * hi;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f16x2 lo = {(f16)(o[dt][4 * g + 0] * inv), (f16)(o[dt][4 * g + 1] * inv)};
                const f16x2 hh = {(f16)(o[dt][4 * g + 2] * inv), (f16)(o[dt][4 * g + 3] * inv)};
                *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g) = u32x2{as_u32(lo), as_u32(hh)};
            }
    }
}

}  // namespace

bool prefill_attention_supports(int D) { return D == 128 || D == 64; }

// strides (elements): {q_b, q_token, q_head, k_b, k_head, k_row, v_b, v_head, v_row, out_b, out_token, out_head}
int launch_prefill_attention(const f16* q, const f16* k, const f16* v, f16* out, int B, int H, int Hkv, int Tq, int Tk, int D,
                             int causal_offset, float scaling, const long* st, hipStream_t stream)
{
    EETQ_REQUIRE(q && k && v && out && st, "null pointer");
    EETQ_REQUIRE(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && Tq > 0 && Tk > 0, "invalid attention shape");
    if (!prefill_attention_supports(D)) return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] prompt attention supports head_dim 64 and 128");
    for (int i = 0; i < 12; ++i) EETQ_REQUIRE(st[i] % 4 == 0, "strides must be multiples of 4 elements");
    for (int i = 0; i < 9; ++i) EETQ_REQUIRE(st[i] % 8 == 0, "q / k / v strides must be multiples of 8 elements (16-byte loads)");
    EETQ_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)out % 8 == 0, "q, k, v must be 16-byte aligned, out 8-byte");
    PrefillArgs a;
    a.q = q, a.k = k, a.v = v, a.out = out;
    a.q_sb = st[0], a.q_st = st[1], a.q_sh = st[2], a.k_sb = st[3], a.k_sh = st[4], a.k_ss = st[5];
    a.v_sb = st[6], a.v_sh = st[7], a.v_ss = st[8], a.o_sb = st[9], a.o_st = st[10], a.o_sh = st[11];
    a.Tq = Tq, a.Tk = Tk, a.koff = causal_offset, a.groups = H / Hkv;
    a.c = scaling * 1.4426950408889634f;
    EETQ_REQUIRE((long)Tk * st[5] * 2 < (1L << 31) && (long)Tk * st[8] * 2 < (1L << 31), "one head's cache rows must span less than 2 GiB");
    a.H = H, a.B = B, a.nqb = (Tq + kPfBM - 1) / kPfBM;
    EETQ_REQUIRE((long)a.nqb * H * B < (1L << 31), "too many workgroups");
    if (D == 128)
        launch_kernel(prefill_attn_kernel<128>, dim3((unsigned)(a.nqb * H * B)), dim3(kPfThreads), 0, stream, a);
    else
        launch_kernel(prefill_attn_kernel<64>, dim3((unsigned)(a.nqb * H * B)), dim3(kPfThreads), 0, stream, a);
    return check_hip(hipGetLastError(), "prefill_attn_kernel launch");
}

}  // namespace eetq
