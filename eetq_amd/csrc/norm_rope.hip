// Side ops of the EETQ operator surface, plain HIP (HBM/L2-bound elementwise + row reduction).
//   rmsnorm : replaces generalT5LayerNorm (/root/reference/csrc/layernorm_kernels/layernorm.cu:25-51) and its
//             launcher (:54-77, :98-113): out = clamp_fp16((x * rsqrt(mean(x^2) + eps)) * gamma), fp32 math,
//             clamp to +-(65504-1000) (reduction.cuh:78-82).
//   rotary  : replaces rotary_embedding_neox_kernel (/root/reference/csrc/embedding_kernels/
//             pos_encoding_kernels.cu:12-53) for fp16: in-place NeoX rotation, fp16 arithmetic with a
//             rounding after every multiply/add exactly like the half operators there.
#include "common.hpp"

namespace eetq {
__device__ unsigned g_rope_dropped = 0;  // decode steps dropped by rotary_neox_kvcache_kernel (cache row outside the cache)
}  // namespace eetq

namespace eetq {

namespace {

constexpr float kHalfClamp = 65504.f - 1000.f;

__device__ __forceinline__ float clamp_for_half(float v)
{
    return v > 0.f ? fminf(v, kHalfClamp) : fmaxf(v, -kHalfClamp);
}

template <int THREADS>
__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < THREADS / 64; ++i) s += red[i];
    return s;
}

// one workgroup per row; 8 fp16 (16 B) per lane per step when cols % 8 == 0
template <int THREADS, bool VEC>
__global__ __launch_bounds__(THREADS) void rmsnorm_kernel(const f16* __restrict__ x, const f16* __restrict__ gamma,
                                                          f16* __restrict__ out, float eps, int cols)
{
    __shared__ float red[THREADS / 64];
    const f16* xr = x + (size_t)blockIdx.x * cols;
    f16*       orow = out + (size_t)blockIdx.x * cols;
    float      ss = 0.f;
    if constexpr (VEC) {
        for (int i = threadIdx.x * 8; i < cols; i += THREADS * 8) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(xr + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
        }
    } else {
        for (int i = threadIdx.x; i < cols; i += THREADS) ss += (float)xr[i] * (float)xr[i];
    }
    const float var = block_sum<THREADS>(ss, red);
    const float s   = rsqrtf(var / (float)cols + eps);
    if constexpr (VEC) {
        for (int i = threadIdx.x * 8; i < cols; i += THREADS * 8) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(xr + i);
            const f16x8 g = *reinterpret_cast<const f16x8*>(gamma + i);
            f16x8       o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)clamp_for_half(((float)v[j] * s) * (float)g[j]);
            *reinterpret_cast<f16x8*>(orow + i) = o;
        }
    } else {
        for (int i = threadIdx.x; i < cols; i += THREADS)
            orow[i] = (f16)clamp_for_half(((float)xr[i] * s) * (float)gamma[i]);
    }
}

// q: [tokens][q_heads][head_size] with q_stride elements between tokens (the reference's layout is the contiguous case,
// q_stride = heads * head_size, k_heads = q_heads); a fused QKV projection output is rotated in place with
// q_stride = k_stride = its row length, and grouped-query models have k_heads < q_heads.
// T = f16 (every product and sum an fp16 operation, like the reference's half operators), float or double (plain IEEE
// operations of that type, no contraction): the reference dispatches float / double / half / bfloat16
// (pos_encoding_kernels.cu:73-86); bf16 is outside this library's scope.
template <typename T>
__global__ void rotary_neox_kernel(const int64_t* __restrict__ positions, T* __restrict__ query,
                                   T* __restrict__ key, const T* __restrict__ cache, int rot_dim, int q_stride,
                                   int k_stride, int q_heads, int k_heads, int head_size)
{
#pragma clang fp contract(off)
    const int     token = blockIdx.x;
    const int64_t pos   = positions[token];
    const T*      cp    = cache + pos * rot_dim;
    const int     embed = rot_dim / 2;
    const int     nq = q_heads * embed, n = nq + k_heads * embed;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const bool   is_k = i >= nq;
        const int    j    = is_k ? i - nq : i;
        const int    head = j / embed;
        const int    off  = j - head * embed;
        T*           p    = (is_k ? key + (size_t)token * k_stride : query + (size_t)token * q_stride) +
               (size_t)head * head_size;
        const T c = cp[off], s = cp[embed + off];
        const T vx = p[off], vy = p[embed + off];
        const T xc = vx * c, ys = vy * s, yc = vy * c, xs = vx * s;
        p[off]         = xc - ys;
        p[embed + off] = yc + xs;
    }
}

// out[r][i] = silu(gu[r][i]) * gu[r][I + i]: the gated-MLP activation on a fused gate|up projection output, one launch
// (torch: silu, then mul).  silu in fp32, rounded to fp16, then an fp16 multiply -- the same roundings as the two torch ops.
__global__ void silu_mul_kernel(const f16* __restrict__ gu, f16* __restrict__ out, int inter, long rows_x_inter)
{
    const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (idx >= rows_x_inter) return;
    const long  r = idx / inter, i = idx - r * inter;
    const f16x8 g = *reinterpret_cast<const f16x8*>(gu + r * 2 * inter + i);
    const f16x8 u = *reinterpret_cast<const f16x8*>(gu + r * 2 * inter + inter + i);
    f16x8       o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = silu_mul_f16(g[j], u[j]);
    *reinterpret_cast<f16x8*>(out + idx) = o;
}

// the same on the "glu8" column order of a fused gate|up projection (groups of 16 = 8 gate + the 8 matching up columns):
// out[r][8 t + c] = silu_mul(gu[r][16 t + c], gu[r][16 t + 8 + c])
__global__ void silu_mul_glu8_kernel(const f16* __restrict__ gu, f16* __restrict__ out, long rows_x_inter)
{
    const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;   // = r * inter + 8 t
    if (idx >= rows_x_inter) return;
    const f16x8 g = *reinterpret_cast<const f16x8*>(gu + 2 * idx);
    const f16x8 u = *reinterpret_cast<const f16x8*>(gu + 2 * idx + 8);
    f16x8       o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = silu_mul_f16(g[j], u[j]);
    *reinterpret_cast<f16x8*>(out + idx) = o;
}

// Decode-step form: one token per batch row.  blockIdx.y = 0 rotates q in place; 1 rotates k and writes it into the KV
// cache at [b][head][pos]; 2 copies v there.  Replaces, for a static cache, the stock sequence arange + add + two index_copy
// launches + the rotary launch by one.
// Prefill form (tokens > 0): blockIdx.x = token b * tokens + t of batch row b (uniform token stride in q / k / v), rotated by
// positions[token], cached at row base + t of cache batch b, base = *slots (a static cache's token counter) or first_row.
// VEC (round 6): rot_dim == head_size, head_size % 16 == 0 and every pointer / stride a multiple of 8 elements -> eight channels per
// thread through 16-byte loads and stores (the scalar form moved a prompt's 62 MB per layer at 3.1 TB/s: 20 us at 1 024 tokens).
// The same fp16 products and sums per channel, contraction off: the same bits.
template <bool VEC>
__global__ void rotary_neox_kvcache_kernel(const int64_t* __restrict__ positions, const int64_t* __restrict__ slots,
                                           int slot_stride, f16* __restrict__ query,
                                           const f16* __restrict__ key, const f16* __restrict__ value,
                                           const f16* __restrict__ cache, f16* __restrict__ kcache,
                                           f16* __restrict__ vcache, int rot_dim, long q_stride, long k_stride,
                                           long v_stride, long c_sb, long c_sh, long c_ss, int q_heads, int k_heads,
                                           int head_size, int max_pos, int tokens, int first_row)
{
#pragma clang fp contract(off)
    const int     b    = blockIdx.x;
    const int     cb   = tokens > 0 ? b / tokens : b;                // batch row of the cache
    const int64_t rpos = positions[b];                               // index into the cos|sin table
    const int64_t pos  = tokens > 0 ? (slots ? slots[0] : (int64_t)first_row) + (b - cb * tokens)
                                    : slots ? slots[(long)b * slot_stride] : rpos;  // cache row the new token is written to
    if (pos < 0 || pos >= max_pos || rpos < 0) {  // never write outside the cache; counted (eetq_decode_dropped_steps)
        if (blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&g_rope_dropped, 1u);
        return;
    }
    const f16* cp    = cache + rpos * rot_dim;
    const int  embed = rot_dim / 2;
    if constexpr (VEC) {
        const int e8 = embed >> 3;  // 8-channel vectors per half head
        auto rot8 = [&](const f16* src, f16* dst, int off) {
            const f16x8 c = *reinterpret_cast<const f16x8*>(cp + off), sn = *reinterpret_cast<const f16x8*>(cp + embed + off);
            const f16x8 vx = *reinterpret_cast<const f16x8*>(src + off), vy = *reinterpret_cast<const f16x8*>(src + embed + off);
            f16x8       lo, hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f16 xc = vx[j] * c[j], ys = vy[j] * sn[j], yc = vy[j] * c[j], xs = vx[j] * sn[j];
                lo[j] = xc - ys;
                hi[j] = yc + xs;
            }
            *reinterpret_cast<f16x8*>(dst + off)         = lo;
            *reinterpret_cast<f16x8*>(dst + embed + off) = hi;
        };
        if (blockIdx.y == 0) {
            for (int i = threadIdx.x; i < q_heads * e8; i += blockDim.x) {
                const int head = i / e8, off = (i - head * e8) * 8;
                f16*      p = query + b * q_stride + (long)head * head_size;
                rot8(p, p, off);
            }
        } else if (blockIdx.y == 1) {
            for (int i = threadIdx.x; i < k_heads * e8; i += blockDim.x) {
                const int head = i / e8, off = (i - head * e8) * 8;
                rot8(key + b * k_stride + (long)head * head_size, kcache + cb * c_sb + head * c_sh + pos * c_ss, off);
            }
        } else {
            const int h8 = head_size >> 3;
            for (int i = threadIdx.x; i < k_heads * h8; i += blockDim.x) {
                const int head = i / h8, off = (i - head * h8) * 8;
                *reinterpret_cast<f16x8*>(vcache + cb * c_sb + head * c_sh + pos * c_ss + off) =
                    *reinterpret_cast<const f16x8*>(value + b * v_stride + (long)head * head_size + off);
            }
        }
        return;
    }
    if (blockIdx.y == 0) {
        for (int i = threadIdx.x; i < q_heads * embed; i += blockDim.x) {
            const int head = i / embed, off = i - head * embed;
            f16*      p = query + b * q_stride + (long)head * head_size;
            const f16 c = cp[off], s = cp[embed + off], vx = p[off], vy = p[embed + off];
            const f16 xc = vx * c, ys = vy * s, yc = vy * c, xs = vx * s;
            p[off]         = xc - ys;
            p[embed + off] = yc + xs;
        }
    } else if (blockIdx.y == 1) {
        for (int i = threadIdx.x; i < k_heads * embed; i += blockDim.x) {
            const int  head = i / embed, off = i - head * embed;
            const f16* p = key + b * k_stride + (long)head * head_size;
            f16*       d = kcache + cb * c_sb + head * c_sh + pos * c_ss;
            const f16  c = cp[off], s = cp[embed + off], vx = p[off], vy = p[embed + off];
            const f16  xc = vx * c, ys = vy * s, yc = vy * c, xs = vx * s;
            d[off]         = xc - ys;
            d[embed + off] = yc + xs;
        }
        const int tail = head_size - rot_dim;  // channels beyond the rotated ones are cached as they are
        for (int i = threadIdx.x; i < k_heads * tail; i += blockDim.x) {
            const int head = i / tail, off = rot_dim + (i - head * tail);
            kcache[cb * c_sb + head * c_sh + pos * c_ss + off] = key[b * k_stride + (long)head * head_size + off];
        }
    } else {
        for (int i = threadIdx.x; i < k_heads * head_size; i += blockDim.x) {
            const int head = i / head_size, off = i - head * head_size;
            vcache[cb * c_sb + head * c_sh + pos * c_ss + off] = value[b * v_stride + (long)head * head_size + off];
        }
    }
}

// Greedy decode hand-over (extension for the HIP-graph decoder, utils/graph_decoder.py): per batch row the index of the row's
// maximum logit -- the FIRST one on ties, a NaN counts as the maximum, like torch.argmax -- written to out_buf[b][*s_idx] and to
// s_tok[b]; then *s_pos += 1 and *s_idx += 1.  ONE workgroup walks the rows (batch is small), so the column is read before anyone
// advances it.  Replaces argmax + scatter_ + copy_ + two add_ launches of every decoded token (~25 us of launches -> ~5).
__global__ __launch_bounds__(1024) void greedy_handover_kernel(const f16* __restrict__ logits, long row_stride, int vocab, int batch,
                                                               int64_t* __restrict__ out_buf, long out_stride, int out_cols,
                                                               int64_t* __restrict__ s_idx, int64_t* __restrict__ s_tok,
                                                               int64_t* __restrict__ s_pos)
{
    __shared__ unsigned long long best_of_wave[16];
    const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t col = *s_idx;
    auto key_of = [](f16 v, int idx) -> unsigned long long {
        const unsigned short u = __builtin_bit_cast(unsigned short, v);
        // monotone 16-bit key of the fp16 value, every NaN above +inf; the low word prefers the smaller index
        // (-0 and +0 share one key: they compare equal)
        const unsigned k = ((u & 0x7FFF) > 0x7C00) ? 0xFFFFu : ((u & 0x7FFF) == 0) ? 0x8000u : (u & 0x8000) ? (unsigned)(unsigned short)~u : (unsigned)(u | 0x8000);
        return ((unsigned long long)k << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)idx);
    };
    for (int b = 0; b < batch; ++b) {
        const f16*         row  = logits + b * row_stride;
        unsigned long long best = 0;
        const int          vec  = ((uintptr_t)row % 16 == 0) ? (vocab & ~7) : 0;
        for (int i = tid * 8; i < vec; i += 1024 * 8) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(row + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned long long k = key_of(v[j], i + j);
                best = k > best ? k : best;
            }
        }
        for (int i = vec + tid; i < vocab; i += 1024) {
            const unsigned long long k = key_of(row[i], i);
            best = k > best ? k : best;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = __shfl_xor(best, m, 64);
            best = o > best ? o : best;
        }
        if (lane == 0) best_of_wave[wave] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = best_of_wave[0];
#pragma unroll
            for (int w = 1; w < 16; ++w) t = best_of_wave[w] > t ? best_of_wave[w] : t;
            const int64_t tok = (int64_t)(0xFFFFFFFFu - (unsigned)(t & 0xFFFFFFFFu));
            if (col >= 0 && col < out_cols) out_buf[b * out_stride + col] = tok;
            s_tok[b] = tok;
        }
        __syncthreads();
    }
    if (tid == 0) {
        *s_pos += 1;
        *s_idx = col + 1;
    }
}

}  // namespace

int launch_greedy_handover(const f16* logits, long row_stride, int vocab, int batch, int64_t* out_buf, long out_stride, int out_cols,
                           int64_t* s_idx, int64_t* s_tok, int64_t* s_pos, hipStream_t stream)
{
    EETQ_REQUIRE(logits && out_buf && s_idx && s_tok && s_pos, "null pointer");
    EETQ_REQUIRE(vocab > 0 && batch >= 0 && out_cols > 0 && row_stride >= vocab && out_stride >= out_cols, "invalid shape");
    if (batch == 0) return EETQ_OK;
    greedy_handover_kernel<<<1, 1024, 0, stream>>>(logits, row_stride, vocab, batch, out_buf, out_stride, out_cols, s_idx, s_tok, s_pos);
    return check_hip(hipGetLastError(), "greedy_handover_kernel launch");
}

int launch_rotary_kvcache(const int64_t* pos, const int64_t* slots, int slot_stride, f16* q, const f16* k, const f16* v,
                          const f16* cache, f16* kcache, f16* vcache, int batch, int q_heads, int k_heads, int head_size, int rot_dim, long q_stride,
                          long k_stride, long v_stride, long c_sb, long c_sh, long c_ss, int max_pos, hipStream_t stream, int tokens,
                          int first_row)
{
    EETQ_REQUIRE(pos && q && k && v && cache && kcache && vcache, "null pointer");
    EETQ_REQUIRE(tokens >= 0 && first_row >= 0 && (slots || (long)first_row + tokens <= max_pos), "the prefill rows must lie inside the cache");
    EETQ_REQUIRE(batch >= 0 && q_heads > 0 && k_heads > 0 && head_size > 0 && rot_dim > 0 && rot_dim % 2 == 0 &&
                     rot_dim <= head_size && max_pos > 0,
                 "invalid rotary shape");
    if (batch == 0) return EETQ_OK;
    const int  blocks = tokens > 0 ? batch * tokens : batch;
    const bool vec    = rot_dim == head_size && head_size % 16 == 0 &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)cache | (uintptr_t)kcache | (uintptr_t)vcache) & 15) == 0 &&
                     ((q_stride | k_stride | v_stride | c_sb | c_sh | c_ss) & 7) == 0;
    if (vec)
        rotary_neox_kvcache_kernel<true><<<dim3(blocks, 3), 256, 0, stream>>>(pos, slots, slot_stride, q, k, v, cache, kcache, vcache,
                                                                               rot_dim, q_stride, k_stride, v_stride, c_sb, c_sh, c_ss,
                                                                               q_heads, k_heads, head_size, max_pos, tokens, first_row);
    else
        rotary_neox_kvcache_kernel<false><<<dim3(blocks, 3), 512, 0, stream>>>(pos, slots, slot_stride, q, k, v, cache, kcache, vcache,
                                                                                rot_dim, q_stride, k_stride, v_stride, c_sb, c_sh, c_ss,
                                                                                q_heads, k_heads, head_size, max_pos, tokens, first_row);
    return check_hip(hipGetLastError(), "rotary_neox_kvcache_kernel launch");
}

int launch_silu_mul(const f16* gu, f16* out, int rows, int inter, hipStream_t stream, bool glu8)
{
    EETQ_REQUIRE(gu && out, "null pointer");
    EETQ_REQUIRE(rows >= 0 && inter > 0 && inter % 8 == 0, "silu_mul: the intermediate size must be a multiple of 8");
    if (rows == 0) return EETQ_OK;
    const long n = (long)rows * inter;
    if (glu8)
        silu_mul_glu8_kernel<<<(unsigned)((n / 8 + 255) / 256), 256, 0, stream>>>(gu, out, n);
    else
        silu_mul_kernel<<<(unsigned)((n / 8 + 255) / 256), 256, 0, stream>>>(gu, out, inter, n);
    return check_hip(hipGetLastError(), "silu_mul_kernel launch");
}

int launch_rmsnorm(const f16* x, const f16* gamma, f16* out, float eps, int rows, int cols, hipStream_t stream)
{
    EETQ_REQUIRE(x && gamma && out, "null pointer");
    EETQ_REQUIRE(rows >= 0 && cols > 0, "invalid shape");
    if (rows == 0) return EETQ_OK;
    if (cols % 8 == 0)
        rmsnorm_kernel<256, true><<<rows, 256, 0, stream>>>(x, gamma, out, eps, cols);
    else
        rmsnorm_kernel<256, false><<<rows, 256, 0, stream>>>(x, gamma, out, eps, cols);
    return check_hip(hipGetLastError(), "rmsnorm_kernel launch");
}

template <typename T>
static int launch_rotary_t(const int64_t* pos, T* q, T* k, const T* cache, int tokens, int q_heads, int k_heads,
                           int head_size, int rot_dim, int q_stride, int k_stride, hipStream_t stream)
{
    EETQ_REQUIRE(pos && q && k && cache, "null pointer");
    EETQ_REQUIRE(tokens >= 0 && q_heads > 0 && k_heads > 0 && head_size > 0 && rot_dim > 0 && rot_dim % 2 == 0 &&
                     rot_dim <= head_size,
                 "invalid rotary shape");
    EETQ_REQUIRE(q_stride >= q_heads * head_size && k_stride >= k_heads * head_size, "invalid rotary token stride");
    if (tokens == 0) return EETQ_OK;
    int threads = (q_heads + k_heads) * rot_dim / 2;
    threads     = threads < 512 ? threads : 512;
    threads     = (threads + 63) / 64 * 64;
    rotary_neox_kernel<T><<<tokens, threads, 0, stream>>>(pos, q, k, cache, rot_dim, q_stride, k_stride, q_heads, k_heads,
                                                          head_size);
    return check_hip(hipGetLastError(), "rotary_neox_kernel launch");
}

int launch_rotary(const int64_t* pos, f16* q, f16* k, const f16* cache, int tokens, int q_heads, int k_heads,
                  int head_size, int rot_dim, int q_stride, int k_stride, hipStream_t stream)
{
    return launch_rotary_t<f16>(pos, q, k, cache, tokens, q_heads, k_heads, head_size, rot_dim, q_stride, k_stride, stream);
}

// dtype: EETQ_DTYPE_F16 / F32 / F64 (query, key and cache share it)
int launch_rotary_any(const int64_t* pos, void* q, void* k, const void* cache, int dtype, int tokens, int q_heads,
                      int k_heads, int head_size, int rot_dim, int q_stride, int k_stride, hipStream_t stream)
{
    switch (dtype) {
        case EETQ_DTYPE_F16:
            return launch_rotary_t<f16>(pos, static_cast<f16*>(q), static_cast<f16*>(k), static_cast<const f16*>(cache), tokens,
                                        q_heads, k_heads, head_size, rot_dim, q_stride, k_stride, stream);
        case EETQ_DTYPE_F32:
            return launch_rotary_t<float>(pos, static_cast<float*>(q), static_cast<float*>(k), static_cast<const float*>(cache),
                                          tokens, q_heads, k_heads, head_size, rot_dim, q_stride, k_stride, stream);
        case EETQ_DTYPE_F64:
            return launch_rotary_t<double>(pos, static_cast<double*>(q), static_cast<double*>(k),
                                           static_cast<const double*>(cache), tokens, q_heads, k_heads, head_size, rot_dim,
                                           q_stride, k_stride, stream);
        default: return fail(EETQ_ERR_INVALID, "[eetq_amd] rotary_embedding_neox: dtype must be float16, float32 or float64");
    }
}

}  // namespace eetq

namespace eetq {
int rope_dropped_steps(unsigned* count, bool reset)
{
    EETQ_TRY_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_rope_dropped), sizeof(unsigned)));
    if (reset) {
        const unsigned zero = 0;
        EETQ_TRY_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rope_dropped), &zero, sizeof(unsigned)));
    }
    return EETQ_OK;
}
}  // namespace eetq
