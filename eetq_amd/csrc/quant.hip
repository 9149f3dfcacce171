// Per-channel symmetric int8 quantiser and weight re-layout kernels (plain HIP, HBM-bound byte work).
//
// Reference behaviour restated (paths relative to /root/reference):
//   quantise  csrc/cutlass_kernels/cutlass_preprocessors.cc:581-678  (ft::symmetric_quantize)
//   sm80 pack csrc/cutlass_kernels/cutlass_preprocessors.cc:497-534  (preprocess_weights_for_mixed_gemm)
// The reference runs these single-threaded on the CPU (three strided K x N passes + four re-layout passes); here
// quant_weights is two launches: per-row-block column maxima (plain stores: no atomics, no zero fill), then one kernel that
// reduces those rows for its columns, quantises and writes the target layout.  For the native layout without a row-major
// copy that kernel is quant_pack_kernel (LDS-DMA ring, column-major in registers, division-free but bit-exact); the sm80
// wire layout and the calls that also return the row-major int8 tensor use strip_quant_kernel (row-wise reads, the same
// arithmetic, byte transpose through LDS).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace eetq {

namespace {

constexpr int kQT     = 64;       // tile edge
constexpr int kQPitch = kQT + 16; // LDS row pitch in bytes (keeps 16-B alignment, breaks power-of-2 stride)

__constant__ int kPerm16[16] = {0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15};

// ---- pass 1: per-column max |w| -----------------------------------------------------------------------
// std::max(a, |w|) with a starting at 0.f ignores NaN (a < NaN is false), see :619-628.  |w| >= 0, so
// the IEEE bit pattern orders like an unsigned integer and atomicMax on the bits is exact.
// Workgroup = kColmaxWaves waves over one strip of 64*V columns (one contiguous 1 KiB per wave-load) x kRowsPerBlock rows:
// wave j takes rows j, j+8, ...: all its 16-byte loads are in flight at once; the waves' maxima meet in LDS and ONE
// store (or atomicMax) per column and workgroup follows (K / 128 per column in total -- the first version issued one atomic
// per column per 32 rows, 524 k atomics at 4096^2, and ran at 1.2 TB/s).
// PARTIALS: no atomics and no zero-fill launch before the kernel -- row block y stores its maxima to row y of a
// [ceil(K / kRowsPerBlock)][N] array and the pack kernel reduces the rows for its 64 columns (the int8 quantiser: two
// launches per call instead of fill + maxima + pack).
constexpr int kRowsPerBlock = 128;
constexpr int kColmaxWaves  = 8;  // 16 rows per wave, all 16 loads of a lane in flight at once: 128 KiB per CU on the wire
                                  // (4 waves x 8 loads = 32 KiB per CU ran at 3.5 TB/s: latency-bound, profiles/r03_quant_pmc.txt)
template <typename T, int V, bool PARTIALS>
__global__ __launch_bounds__(64 * kColmaxWaves) void colmax_kernel(const T* __restrict__ w, size_t K, size_t N,
                                                                  u32* __restrict__ colmax_bits)
{
    __shared__ float part[kColmaxWaves][64 * V];
    const int    wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t col0 = ((size_t)blockIdx.x * 64 + lane) * V;
    const size_t k_begin = (size_t)blockIdx.y * kRowsPerBlock;
    size_t       k_end   = k_begin + kRowsPerBlock;
    if (k_end > K) k_end = K;
    const bool   live = col0 < N;
    const size_t cc   = live ? col0 : 0;  // dead lanes (ragged last strip) read column 0 and are ignored
    constexpr int kBatch = kRowsPerBlock / kColmaxWaves;
    u32x4 raw[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
        const size_t k  = k_begin + wave + (size_t)kColmaxWaves * j;
        const size_t kk = k < k_end ? k : k_end - 1;  // clamped, never a branch around the load
        raw[j]          = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + kk * N + cc));
    }
    // max |w| with NaN ignored (std::max(a, |w|) keeps a when |w| is NaN, :619-628): the hardware's maxnum does exactly that,
    // in the input type (a maximum is exact in any format)
    float m[V];
    if constexpr (sizeof(T) == 2) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 acc[4] = {h2{0, 0}, h2{0, 0}, h2{0, 0}, h2{0, 0}};
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const u32 d[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_max(acc[i], __builtin_bit_cast(h2, d[i] & 0x7fff7fffu));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[2 * i]     = (float)acc[i].x;
            m[2 * i + 1] = (float)acc[i].y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) m[i] = 0.f;
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            T v[V];
            *reinterpret_cast<u32x4*>(v) = raw[j];
#pragma unroll
            for (int i = 0; i < V; ++i) m[i] = __builtin_fmaxf(m[i], __builtin_fabsf((float)v[i]));
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) part[wave][lane * V + i] = m[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * V; c += 64 * kColmaxWaves) {
        const size_t col = (size_t)blockIdx.x * 64 * V + c;
        if (col < N) {
            float a = part[0][c];
#pragma unroll
            for (int j = 1; j < kColmaxWaves; ++j) a = __builtin_fmaxf(a, part[j][c]);
            if constexpr (PARTIALS)
                colmax_bits[(size_t)blockIdx.y * N + col] = __builtin_bit_cast(u32, a);
            else
                atomicMax(colmax_bits + col, __builtin_bit_cast(u32, a));
        }
    }
}

// ---- element quantiser ----------------------------------------------------------------------------------
// :644-648  q = int8(max(-128.f, min(127.f, round(w / s)))) with std::min/std::max NaN behaviour and C
// round() (half away from zero).  IEEE fp32 division (no reciprocal), subnormals kept.
__device__ __forceinline__ int8_t quantize_elt(float w, float s)
{
    const float scaled = __builtin_roundf(w / s);
    const float hi     = (scaled < 127.f) ? scaled : 127.f;  // std::min(127.f, scaled)
    const float lo     = (-128.f < hi) ? hi : -128.f;        // std::max(-128.f, hi)
    return (int8_t)(int)lo;
}

// int4 (PACKED_INT4_WEIGHT_ONLY, :581-678): q = clamp(int(round(w / s)), -8, 7) with scale = amax / 8; int(NaN) / out of range is
// INT_MIN on the reference's x86 hosts, i.e. -8.  Returns the unsigned nibble q + 8.
__device__ __forceinline__ unsigned quantize_i4_elt(float w, float s)
{
    const float scaled = __builtin_roundf(w / s);
    int         iw;
    if (scaled != scaled || scaled >= 2147483648.f || scaled < -2147483648.f) iw = (int)0x80000000;
    else iw = (int)scaled;
    const int c = iw < -8 ? -8 : (iw > 7 ? 7 : iw);
    return (unsigned)(c + 8);
}

// ---- the same result without a division ------------------------------------------------------------------------
// Returns q + 128 as an integer-valued float in [0, 256] (256 = the reference's min(127, .) case, saturated by the
// conversion to a byte).  r = rcp(s), trusted only for 1e-30 < s < 1e30 (the caller passes NaN otherwise).
// b = fma(w, r, 128) is w / s + 128 to within 3.1e-5 (t = w / s, |t| <= 128: the 1-ulp reciprocal moves w * r by <= 1.53e-5,
// the fma's rounding below 256 by <= 0.77e-5, and the reference's correctly rounded quotient is within 0.77e-5 of t), which
// names the two integers the result can be, f = floor(b) and f + 1 -- an f that is off by one next to an integer picks the
// same q.  Which one: the SIGN of m * s - w, m = f - 127.5 the tie between them as a quotient, one fma, exact in sign (one
// rounding, never to zero); zero is an exact tie, which C round() sends away from zero.  "z < 0, or z == 0 and w > 0" is ONE
// comparison of the bit patterns as signed integers, z <= (w < 0 ? -1 : 0): a negative float is a negative integer, +0 is 0
// (an exact cancellation gives +0), a positive float is >= 1; for fp16 input any negative bound >= -32768 does (only NaN
// patterns lie above it), so the sign-extended fp16 bits themselves serve.
// The caller must redo the element with quantize_elt when `nanacc` comes out NaN (a NaN weight or reciprocal) or -- fp32
// input only -- its lane is set in `gray`: the reference rounds the QUOTIENT to fp32 first, and it becomes the tie m itself
// when 0 < |w / s - m| <= half an ulp of m, i.e. |z| <= s * 2^(exponent(m) - 24); fp16 inputs cannot get that close (two
// 11-bit significands put w / s at least 2^-19 |m| away from an m it does not equal).
// BIAS = 128 (int8: |w / s| <= 128) or 8 (int4: scale = max / 8, |w / s| <= 8, the error bound only shrinks).
template <typename T, int BIAS = 128>
__device__ __forceinline__ float quantize_biased_nodiv(T raw, float s, float r, float& nanacc, unsigned long long& gray)
{
    const float wf = (float)raw;
    const float b  = __builtin_fmaf(wf, r, (float)BIAS);
    const float f  = __builtin_floorf(b);
    const float m  = f - ((float)BIAS - 0.5f);
    const float z  = __builtin_fmaf(m, s, -wf);
    int         bound;
    if constexpr (sizeof(T) == 2)
        bound = (int)(short)__builtin_bit_cast(unsigned short, raw);
    else
        bound = __builtin_bit_cast(int, wf) >> 31;
    bound  = bound < 0 ? bound : 0;
    nanacc = __builtin_fmaf(b, 0.f, nanacc);
    if constexpr (sizeof(T) == 4) {
        const float hulp = __builtin_bit_cast(float, (__builtin_bit_cast(int, m) & 0x7f800000) - (24 << 23));
        gray |= __builtin_amdgcn_ballot_w64(z != 0.f && __builtin_fabsf(z) <= s * hulp);
    }
    return f + ((__builtin_bit_cast(int, z) <= bound) ? 1.f : 0.f);
}

// the reciprocal the function above may use for scale s, or NaN (-> every element of the column takes the exact path)
__device__ __forceinline__ float trusted_rcp(float s)
{
    return (s > 1e-30f && s < 1e30f) ? __builtin_amdgcn_rcpf(s) : __builtin_nanf("");
}

// ---- re-layout of a raw int8 tensor: one 64x64 tile per workgroup -----------------------------------------------------
// grid = (ceil(N/64), K/64), block = 256.  Thread t loads row r = t/4, columns seg*16..+15 (seg = t%4).
template <int LAYOUT>
__global__ __launch_bounds__(256) void tile_pack_kernel(const int8_t* __restrict__ src, size_t K, size_t N,
                                                        uint8_t* __restrict__ q_packed)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kQT][kQPitch];
    const int    t   = threadIdx.x;
    const size_t n0  = (size_t)blockIdx.x * kQT;
    const size_t kt  = blockIdx.y;
    const size_t k0  = kt * kQT;
    const int    r   = t >> 2;
    const int    seg = t & 3;
    const size_t nc  = n0 + (size_t)seg * 16;

    if (nc < N) *reinterpret_cast<u32x4*>(&tile[r][seg * 16]) = *reinterpret_cast<const u32x4*>(src + (k0 + r) * N + nc);
    __syncthreads();

    if constexpr (LAYOUT == EETQ_LAYOUT_GFX950) {
        // 4 tiles of 16 columns; thread = (chunk, lane); lane = g*16 + c holds k-locals 16g..16g+15 of column c
        const int chunk = t >> 6, lane = t & 63, g = lane >> 4, c = lane & 15;
        if (n0 + (size_t)chunk * 16 < N) {
            u32 d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 b0 = tile[16 * g + 4 * i + 0][chunk * 16 + c];
                const u32 b1 = tile[16 * g + 4 * i + 2][chunk * 16 + c];  // bytes 1<->2 swapped
                const u32 b2 = tile[16 * g + 4 * i + 1][chunk * 16 + c];
                const u32 b3 = tile[16 * g + 4 * i + 3][chunk * 16 + c];
                d[i]         = (b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)) ^ 0x80808080u;  // +128
            }
            const size_t ntile = (n0 >> 4) + chunk;
            uint8_t*     dst   = q_packed + (ntile * (K >> 6) + kt) * (size_t)kTileBytes + (size_t)lane * 16;
            *reinterpret_cast<u32x4*>(dst) = u32x4{d[0], d[1], d[2], d[3]};
        }
    } else if constexpr (LAYOUT == EETQ_LAYOUT_SM80) {
        // closed form of P1..P4: for column pair cp and 64-row tile kt the 128 output
        // bytes are [even column k 0..63][odd column k 0..63] with rows permuted inside 16 and bytes 1<->2
        // of each dword swapped, +128.
        const int pair = t >> 3, sg = t & 7, half = sg >> 2, kk = (sg & 3) * 16;
        u32       d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 b0 = tile[kk + kPerm16[4 * i + 0]][2 * pair + half];
            const u32 b1 = tile[kk + kPerm16[4 * i + 2]][2 * pair + half];
            const u32 b2 = tile[kk + kPerm16[4 * i + 1]][2 * pair + half];
            const u32 b3 = tile[kk + kPerm16[4 * i + 3]][2 * pair + half];
            d[i]         = (b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)) ^ 0x80808080u;
        }
        uint8_t* dst = q_packed + ((n0 >> 1) + pair) * (2 * K) + kt * 128 + (size_t)half * 64 + kk;
        *reinterpret_cast<u32x4*>(dst) = u32x4{d[0], d[1], d[2], d[3]};
    }
}

// ---- pass 1b (EETQ_AMD_QUANT_FOLD=1 only): fold the P row-block maxima of every column into row 0 (one thread per
// column, coalesced rows), so that the pack kernel reads one row.
__global__ __launch_bounds__(256) void colmax_fold_kernel(float* __restrict__ part, size_t N, int P)
{
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float m = part[n];
    for (int p = 1; p < P; ++p) {
        const float a = part[(size_t)p * N + n];
        m             = (m < a) ? a : m;
    }
    part[n] = m;
}

// ---- pass 2, int8 quantiser form: a workgroup takes kStripTiles (1, 2 or 4) consecutive 64x64 tiles of one 64-column strip
// grid = (ceil(N/64), ceil(K/64 / kStripTiles)), block = 256.  All of the strip's loads are issued first (2 x 16 B per lane
// and tile at fp16); while they fly the workgroup reduces the P row-block maxima of its 64 columns ONCE (thread t: column
// t % 64, rows t / 64 + 4 i) -- maxima are order-independent, so the result is the reference's single running maximum
// (:619-628) bit for bit.  Then every tile is quantised into its own LDS image, one barrier, and written in the target
// layout with 16-byte stores exactly like tile_pack_kernel.
template <typename T, int LAYOUT, int kStripTiles, bool NT = false>
__global__ __launch_bounds__(256) void strip_quant_kernel(const T* __restrict__ src, size_t K, size_t N,
                                                          const float* __restrict__ part, int P,
                                                          int8_t* __restrict__ q_raw, uint8_t* __restrict__ q_packed,
                                                          void* __restrict__ scales, int scales_f32)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kStripTiles][kQT][kQPitch];
    __shared__ float cm[4][kQT];
    __shared__ __attribute__((aligned(16))) float cmx[kQT], crx[kQT];
    constexpr int kVecs = (int)(sizeof(T) * 16 / 16);  // 16-byte loads per 16 elements
    const int    t   = threadIdx.x;
    const size_t n0  = (size_t)blockIdx.x * kQT;
    const size_t kt0 = (size_t)blockIdx.y * kStripTiles;
    const size_t KT  = K / kQT;
    const int    nt  = (int)(KT - kt0 < (size_t)kStripTiles ? KT - kt0 : (size_t)kStripTiles);
    const int    r   = t >> 2;
    const int    seg = t & 3;
    const size_t nc  = n0 + (size_t)seg * 16;
    const bool   live = nc < N;
    const size_t ncc  = live ? nc : 0;  // ragged last strip: dead segments read column 0 and are ignored

    u32x4 raw[kStripTiles][kVecs];
#pragma unroll
    for (int j = 0; j < kStripTiles; ++j) {
        const size_t kt = kt0 + (j < nt ? j : nt - 1);  // clamped: no load behind a branch
        const u32x4* p  = reinterpret_cast<const u32x4*>(src + (kt * kQT + r) * N + ncc);
#pragma unroll
        for (int i = 0; i < kVecs; ++i) raw[j][i] = NT ? __builtin_nontemporal_load(p + i) : p[i];
    }
    {
        const int    col = t & 63, p0 = t >> 6;
        const size_t cg  = n0 + col < N ? n0 + col : 0;
        float        m   = 0.f;
        for (int p = p0; p < P; p += 32) {  // 8 independent loads in flight per thread (clamped rows: a repeat is harmless)
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pp = p + 4 * j;
                a[j]         = part[(size_t)(pp < P ? pp : P - 1) * N + cg];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) m = (m < a[j]) ? a[j] : m;
        }
        cm[p0][col] = m;
    }
    __syncthreads();
    if (t < kQT) {
        float a = cm[0][t];
#pragma unroll
        for (int j = 1; j < 4; ++j) a = (a < cm[j][t]) ? cm[j][t] : a;
        cmx[t] = a;
        crx[t] = trusted_rcp(a * (1.f / 128.f));
        if (kt0 == 0 && n0 + t < N && scales) {  // :633-634 scale = T(colmax * 2^-7), written once per column
            const float s32 = a * (1.f / 128.f);
            if (scales_f32)
                reinterpret_cast<float*>(scales)[n0 + t] = s32;
            else
                reinterpret_cast<f16*>(scales)[n0 + t] = (f16)s32;
        }
    }
    __syncthreads();
    float s[16], rc[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        reinterpret_cast<f32x4*>(s)[i]  = reinterpret_cast<const f32x4*>(cmx + seg * 16)[i];
        reinterpret_cast<f32x4*>(rc)[i] = reinterpret_cast<const f32x4*>(crx + seg * 16)[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] *= (1.f / 128.f);
#pragma unroll
    for (int j = 0; j < kStripTiles; ++j) {
        if (j < nt && live) {
            union {
                int8_t b[16];
                u32x4  v;
            } out;
            T v[16];
#pragma unroll
            for (int i = 0; i < kVecs; ++i) reinterpret_cast<u32x4*>(v)[i] = raw[j][i];
            // division-free, bit-exact (quantize_biased_nodiv); the rare lanes it cannot decide redo their 16 elements
            u32                d[4]   = {0u, 0u, 0u, 0u};
            unsigned long long gray   = 0;
            float              nanacc = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                nanacc = __builtin_fmaf(rc[i], 0.f, nanacc);
                d[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(quantize_biased_nodiv<T>(v[i], s[i], rc[i], nanacc, gray), i & 3, d[i >> 2]);
            }
            out.v = u32x4{d[0] ^ 0x80808080u, d[1] ^ 0x80808080u, d[2] ^ 0x80808080u, d[3] ^ 0x80808080u};  // q + 128 -> int8
            if (nanacc != nanacc || ((gray >> (t & 63)) & 1)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) out.b[i] = quantize_elt((float)v[i], s[i]);
            }
            *reinterpret_cast<u32x4*>(&tile[j][r][seg * 16]) = out.v;
            if (q_raw) *reinterpret_cast<u32x4*>(q_raw + ((kt0 + j) * kQT + r) * N + nc) = out.v;
        }
    }
    if (!q_packed) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kStripTiles; ++j) {
        if (j >= nt) break;
        const size_t kt = kt0 + j;
        if constexpr (LAYOUT == EETQ_LAYOUT_GFX950) {
            const int chunk = t >> 6, lane = t & 63, g = lane >> 4, c = lane & 15;
            if (n0 + (size_t)chunk * 16 < N) {
                u32 d[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 b0 = tile[j][16 * g + 4 * i + 0][chunk * 16 + c];
                    const u32 b1 = tile[j][16 * g + 4 * i + 2][chunk * 16 + c];  // bytes 1<->2 swapped
                    const u32 b2 = tile[j][16 * g + 4 * i + 1][chunk * 16 + c];
                    const u32 b3 = tile[j][16 * g + 4 * i + 3][chunk * 16 + c];
                    d[i]         = (b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)) ^ 0x80808080u;  // +128
                }
                const size_t ntile = (n0 >> 4) + chunk;
                uint8_t*     dst   = q_packed + (ntile * KT + kt) * (size_t)kTileBytes + (size_t)lane * 16;
                *reinterpret_cast<u32x4*>(dst) = u32x4{d[0], d[1], d[2], d[3]};
            }
        } else if constexpr (LAYOUT == EETQ_LAYOUT_SM80) {
            const int pair = t >> 3, sg = t & 7, half = sg >> 2, kk = (sg & 3) * 16;
            u32       d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 b0 = tile[j][kk + kPerm16[4 * i + 0]][2 * pair + half];
                const u32 b1 = tile[j][kk + kPerm16[4 * i + 2]][2 * pair + half];
                const u32 b2 = tile[j][kk + kPerm16[4 * i + 1]][2 * pair + half];
                const u32 b3 = tile[j][kk + kPerm16[4 * i + 3]][2 * pair + half];
                d[i]         = (b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)) ^ 0x80808080u;
            }
            uint8_t* dst = q_packed + ((n0 >> 1) + pair) * (2 * K) + kt * 128 + (size_t)half * 64 + kk;
            *reinterpret_cast<u32x4*>(dst) = u32x4{d[0], d[1], d[2], d[3]};
        }
    }
}

// ---- quantise + native (gfx950) layout in one pass, column-major in registers -----------------------------------------
// The native layout gives lane (g, c) of a 16-column x 64-row tile the 16 bytes of ONE column (rows 16g .. 16g+15, rows
// 1 <-> 2 of every four swapped), i.e. the transpose of how the weight is read.  strip_quant_kernel does that transpose on
// the quantised BYTES (16 ds_read_u8 + shifts/ors per thread) after an IEEE division per element, and is instruction-bound:
// 33 VALU ops per element, 15.5 us of issue time at 4096^2 against 20 us measured (profiles/r03_quant_isa.txt).  Here
//   * the 64 x 64 tile goes HBM -> LDS by LDS-DMA (no registers, no address math per element), rows padded per 1 KiB block
//     so that the column-wise reads below are bank-conflict free;
//   * every lane reads the 16 elements of its column as T and quantises them straight into the output byte positions
//     (v_cvt_pk_u8_f32 of q + 128 does the [-128, 127] clamp, the +128 bias and the byte packing in one instruction);
//   * no IEEE division (quantize_biased_nodiv above).  Exact ties are not rare on fp16 grids (1.6 % of the elements of a
//     uniform fp16 matrix, 7.9 % of nn.Linear's default init), so they are resolved inline; lanes with a NaN weight, a
//     zero / subnormal / huge / NaN scale or (fp32 input) a quotient whose fp32 rounding could land on a tie redo their 16
//     elements with quantize_elt (the reference's arithmetic verbatim).
// grid = (ceil(N/64), ceil(K/64/tiles_per_wg)), block = 256: wave = 16-column chunk, lane = (g, c).
typedef __attribute__((address_space(3))) void qp_lds_void;
// 64 lanes x 16 bytes, global (buffer descriptor + per-lane offset) -> LDS at lds_wave_base + lane * 16, no registers
__device__ __forceinline__ void qp_dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, uint8_t* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (qp_lds_void*)lds_wave_base, 16, voff, 0, 0, 0);
}

// BITS = 8: 64-row tiles, a lane reads 16 rows of its column.  BITS = 4: the int4 native tile is 16 columns x 128 k, a lane
// holds 32 k values of its column (int4.hip): 128-row tiles, 32 rows per lane.
template <typename T, int BITS = 8>
struct QpCfg {
    static constexpr int kRows      = BITS == 8 ? 64 : 128;     // rows of a tile
    static constexpr int kLaneRows  = kRows / 4;                // rows a lane quantises: 16 / 32
    static constexpr int kRowBytes  = kQT * (int)sizeof(T);     // 128 / 256
    static constexpr int kRowsPerB  = 1024 / kRowBytes;         // rows per 1 KiB DMA block: 8 / 4
    static constexpr int kBlocks    = kRows / kRowsPerB;        // DMA blocks per tile
    // lanes g = 0..3 read rows kLaneRows apart: the pad puts them 16 banks apart (int8: both element sizes; int4: fp16 --
    // fp32 input takes a two-way conflict there, a multiple of 16 bytes cannot avoid it)
    static constexpr int kPad       = BITS == 8 ? (sizeof(T) == 2 ? 32 : 16) : 16;
    static constexpr int kBlockLds  = 1024 + kPad;
    static constexpr int kTileLds   = kBlocks * kBlockLds;
};

// A workgroup walks `tiles_per_wg` consecutive 64-row tiles of its 64-column strip with a two-slot LDS ring: tile j+2's DMA
// is issued as soon as every wave has read tile j out of its slot, so loads stay in flight under the arithmetic and the
// stores (one tile per workgroup and launch-wide lock-step phases -- everyone loads, then everyone computes, then everyone
// stores -- measured 18 us at 4096^2 with 45 % fewer instructions than the row-major kernel's 17 us).  The column scales
// (the reduction of the P row-block maxima) are computed once per workgroup, not once per tile.
// BITS = 4: the same walk over the int4 native layout (`part` then holds the FINAL column maxima, P = 1: the int4 entry's
// workspace is N floats), 32 rows per lane, nibbles q + 8 at positions [0, 4, 1, 5, 2, 6, 3, 7] of each dword (int4.hip).
template <typename T, int BITS = 8>
__global__ __launch_bounds__(256) void quant_pack_kernel(const T* __restrict__ src, unsigned K, unsigned N,
                                                         const float* __restrict__ part, int P,
                                                         uint8_t* __restrict__ q_packed, void* __restrict__ scales,
                                                         int scales_f32, int tiles_per_wg)
{
    using C = QpCfg<T, BITS>;
    constexpr int kRows = C::kRows, kLR = C::kLaneRows;
    constexpr int kPer = C::kBlocks / 4;  // DMA instructions per wave and tile
    // the ring is DYNAMIC LDS (2 * kTileLds bytes): with a static array hipcc puts an s_waitcnt vmcnt(0) in front of the
    // first LDS read after every LDS-DMA issue (it can name the object both touch), i.e. waits for the prefetched tile too
    extern __shared__ __attribute__((aligned(16))) uint8_t tiles[];
    __shared__ float cm[4][kQT];
    __shared__ float col_s[kQT], col_r[kQT];
    const int      t    = threadIdx.x;
    const int      wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int      lane = t & 63;
    const unsigned n0   = blockIdx.x * kQT;
    const unsigned KT   = K / kRows;
    const unsigned kt0  = blockIdx.y * (unsigned)tiles_per_wg;
    const int      nt   = (int)(KT - kt0 < (unsigned)tiles_per_wg ? KT - kt0 : (unsigned)tiles_per_wg);

    // ---- tiles: HBM -> LDS.  The descriptor starts at this workgroup's first element, so offsets stay 32-bit for any
    // tensor size; a ragged last strip reads on into the next row (ignored columns) or past the end (zero-filled).
    const size_t first = ((size_t)kt0 * kRows * N + n0) * sizeof(T);
    const size_t left  = (size_t)K * N * sizeof(T) - first;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(src) + ((size_t)kt0 * kRows * N + n0), 0, left > 0xffffffffull ? (int)0xffffffffu : (int)left, 0x00020000);
    constexpr int  kSegs    = C::kRowBytes / 16;  // 16-byte pieces per row
    const unsigned lane_off = ((unsigned)(lane / kSegs) + (unsigned)wave * (kPer * C::kRowsPerB)) * N * (unsigned)sizeof(T) +
                              (unsigned)(lane % kSegs) * 16u;
    const unsigned tile_pitch = kRows * N * (unsigned)sizeof(T);
    auto issue = [&](int j, int slot) {
        uint8_t* base = tiles + slot * C::kTileLds + wave * kPer * C::kBlockLds;
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (qp_lds_void*)(base + i * C::kBlockLds), 16,
                                                     (int)(lane_off + (unsigned)(i * C::kRowsPerB) * N * (unsigned)sizeof(T)),
                                                     (int)((unsigned)j * tile_pitch), 0, 0);
    };
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    // ---- this strip's 64 column maxima from the P row-block partials ----
    {
        const int      col = t & 63, p0 = t >> 6;
        const unsigned cg  = n0 + col < N ? n0 + col : 0;
        float          m   = 0.f;
        for (int p = p0; p < P; p += 32) {  // 8 independent loads in flight per thread (clamped rows: a repeat is harmless)
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pp = p + 4 * j;
                a[j]         = part[(size_t)((unsigned)(pp < P ? pp : P - 1) * N + cg)];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) m = (m < a[j]) ? a[j] : m;
        }
        cm[p0][col] = m;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t < kQT) {
        float a = cm[0][t];
#pragma unroll
        for (int j = 1; j < 4; ++j) a = (a < cm[j][t]) ? cm[j][t] : a;
        const float s32 = a * (BITS == 8 ? 1.f / 128.f : 1.f / 8.f);  // :633-634 scale = T(colmax * 2^-7 | 2^-3), written once per column
        if (kt0 == 0 && n0 + t < N && scales) {
            if (scales_f32)
                reinterpret_cast<float*>(scales)[n0 + t] = s32;
            else
                reinterpret_cast<f16*>(scales)[n0 + t] = (f16)s32;
        }
        col_s[t] = s32;
        col_r[t] = trusted_rcp(s32);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the scale stores too: from here on vmcnt counts tiles
    __builtin_amdgcn_s_barrier();
    const bool  live = n0 + (unsigned)wave * 16 < N;  // false: whole 16-column chunk beyond a ragged N (DMA and barriers only)
    const int   g = lane >> 4, c = lane & 15;
    const int   col = wave * 16 + c;
    const float s = col_s[col], r = col_r[col];
    // rows kLR*g .. start at block kLR*g / kRowsPerB; element (row, col) sits at row % kRowsPerB * kRowBytes + col * sizeof(T)
    typedef __attribute__((address_space(3))) const T lds_cT;  // integer LDS addresses as in gemm_kernel.hpp::lds_read16
    const int lbase = (int)(uint32_t)(uintptr_t)(qp_lds_void*)tiles + (kLR * g / C::kRowsPerB) * C::kBlockLds + col * (int)sizeof(T);
    uint8_t* dst = q_packed + ((size_t)((n0 >> 4) + wave) * KT + kt0) * (size_t)kTileBytes + (size_t)lane * 16;
    u32x4 done = {0u, 0u, 0u, 0u};  // tile j-1's bytes, stored during step j
    for (int j = 0; j < nt; ++j) {
        // In issue order this wave has outstanding: tile j's pieces, [the store of tile j-2], [tile j+1's pieces].  "At most
        // kPer outstanding" then means tile j has landed WITHOUT assuming that a store retires in order with the loads around
        // it (LLVM does not assume it either on gfx9: mixed load / store events make its vmcnt "out of order"): loads do retire
        // in order among themselves, so one missing piece of tile j would leave all kPer of tile j+1 outstanding as well.
        // The store itself is a whole step old by now -- which is why tile j-1 is stored below, after this wait, and not at
        // the end of its own step.
        if (j + 1 < nt)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPer) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone's pieces of tile j have landed
        if (live && j > 0) *reinterpret_cast<u32x4*>(dst + (size_t)(j - 1) * kTileBytes) = done;
        T raw[kLR];
        if (live) {
            const int lt = lbase + (j & 1) * C::kTileLds;
#pragma unroll
            for (int i = 0; i < kLR; ++i)
                raw[i] = *(lds_cT*)(uintptr_t)(uint32_t)(lt + (i / C::kRowsPerB) * C::kBlockLds + (i % C::kRowsPerB) * C::kRowBytes);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone has read the slot: refill it
        if (j + 2 < nt) issue(j + 2, j & 1);
        if (!live) continue;
        u32 d[4] = {0u, 0u, 0u, 0u};
        unsigned long long gray = 0;    // fp32 input only: lanes where the fp32 ROUNDING of w / s may land on a tie
        float              nanacc = r;  // NaN as soon as a w (or the column's reciprocal) is
        if constexpr (BITS == 4) {
#pragma unroll
            for (int i = 0; i < kLR; ++i) {
                // q + 8 in [0, 16]; 16 (w = +max of the column) is the reference's min(7, .): clamp, then the nibble goes to
                // position (j >> 1) + 4 * (j & 1) of dword i / 8, j = i % 8
                const float    q = __builtin_fminf(quantize_biased_nodiv<T, 8>(raw[i], s, r, nanacc, gray), 15.f);
                constexpr int  kNib[8] = {0, 4, 1, 5, 2, 6, 3, 7};
                d[i >> 3] |= (unsigned)q << (4 * kNib[i & 7]);
            }
            if (nanacc != nanacc || ((gray >> lane) & 1)) {  // rare lanes: the reference's arithmetic verbatim
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d[q] = 0u;
                    constexpr int kNib[8] = {0, 4, 1, 5, 2, 6, 3, 7};
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[q] |= quantize_i4_elt((float)raw[8 * q + e], s) << (4 * kNib[e]);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float q = quantize_biased_nodiv<T>(raw[i], s, r, nanacc, gray);
            // output byte of row i inside its group of four: rows 1 <-> 2 swapped.  The conversion saturates: 256 -> 255 is the
            // reference's min(127, .) for w = +max of the column.
            constexpr int kByte[4] = {0, 2, 1, 3};
            d[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(q, kByte[i & 3], d[i >> 2]);
        }
        if (nanacc != nanacc || ((gray >> lane) & 1)) {  // rare lanes: the reference's arithmetic verbatim
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 b0 = (uint8_t)quantize_elt((float)raw[4 * q + 0], s);
                const u32 b1 = (uint8_t)quantize_elt((float)raw[4 * q + 2], s);
                const u32 b2 = (uint8_t)quantize_elt((float)raw[4 * q + 1], s);
                const u32 b3 = (uint8_t)quantize_elt((float)raw[4 * q + 3], s);
                d[q]         = (b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)) ^ 0x80808080u;
            }
        }
        }
        done = u32x4{d[0], d[1], d[2], d[3]};
    }
    if (live) *reinterpret_cast<u32x4*>(dst + (size_t)(nt - 1) * kTileBytes) = done;
}

// ---- inverse: packed layout -> raw row-major ----------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void tile_unpack_kernel(const uint8_t* __restrict__ q_packed, size_t K, size_t N,
                                                          int8_t* __restrict__ q_raw)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kQT][kQPitch];
    const int    t  = threadIdx.x;
    const size_t n0 = (size_t)blockIdx.x * kQT;
    const size_t kt = blockIdx.y;
    const size_t k0 = kt * kQT;

    if constexpr (LAYOUT == EETQ_LAYOUT_GFX950) {
        const int chunk = t >> 6, lane = t & 63, g = lane >> 4, c = lane & 15;
        if (n0 + (size_t)chunk * 16 < N) {
            const size_t   ntile = (n0 >> 4) + chunk;
            const uint8_t* s     = q_packed + (ntile * (K >> 6) + kt) * (size_t)kTileBytes + (size_t)lane * 16;
            const u32x4    v     = *reinterpret_cast<const u32x4*>(s);
            const u32      d[4]  = {v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tile[16 * g + 4 * i + 0][chunk * 16 + c] = (uint8_t)(d[i]);
                tile[16 * g + 4 * i + 2][chunk * 16 + c] = (uint8_t)(d[i] >> 8);
                tile[16 * g + 4 * i + 1][chunk * 16 + c] = (uint8_t)(d[i] >> 16);
                tile[16 * g + 4 * i + 3][chunk * 16 + c] = (uint8_t)(d[i] >> 24);
            }
        }
    } else {
        const int      pair = t >> 3, sg = t & 7, half = sg >> 2, kk = (sg & 3) * 16;
        const uint8_t* s    = q_packed + ((n0 >> 1) + pair) * (2 * K) + kt * 128 + (size_t)half * 64 + kk;
        const u32x4    v    = *reinterpret_cast<const u32x4*>(s);
        const u32      d[4] = {v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tile[kk + kPerm16[4 * i + 0]][2 * pair + half] = (uint8_t)(d[i]);
            tile[kk + kPerm16[4 * i + 2]][2 * pair + half] = (uint8_t)(d[i] >> 8);
            tile[kk + kPerm16[4 * i + 1]][2 * pair + half] = (uint8_t)(d[i] >> 16);
            tile[kk + kPerm16[4 * i + 3]][2 * pair + half] = (uint8_t)(d[i] >> 24);
        }
    }
    __syncthreads();
    const int    r = t >> 2, seg = t & 3;
    const size_t nc = n0 + (size_t)seg * 16;
    if (nc < N)
        *reinterpret_cast<u32x4*>(q_raw + (k0 + r) * N + nc) = *reinterpret_cast<const u32x4*>(&tile[r][seg * 16]);
}

int check_layout_shape(size_t K, size_t N, int layout)
{
    EETQ_REQUIRE(K > 0 && N > 0, "weight should not be empty tensor");
    EETQ_REQUIRE(K % 64 == 0, "the number of rows (K) of the quantized matrix must be a multiple of 64");
    EETQ_REQUIRE(N % 16 == 0, "the number of columns (N) must be a multiple of 16");
    if (layout == EETQ_LAYOUT_SM80)
        EETQ_REQUIRE(N % 64 == 0, "The number of columns must be a multiple of 64 (sm80 layout)");
    EETQ_REQUIRE(layout == EETQ_LAYOUT_GFX950 || layout == EETQ_LAYOUT_SM80 || layout == EETQ_LAYOUT_ROW_MAJOR,
                 "unknown weight layout");
    return EETQ_OK;
}

int launch_tile_pack(const int8_t* src, size_t K, size_t N, int8_t* q_packed, int layout, hipStream_t stream)
{
    dim3     grid((unsigned)((N + kQT - 1) / kQT), (unsigned)(K / kQT));
    uint8_t* p = reinterpret_cast<uint8_t*>(q_packed);
    if (layout == EETQ_LAYOUT_SM80)
        tile_pack_kernel<EETQ_LAYOUT_SM80><<<grid, 256, 0, stream>>>(src, K, N, p);
    else
        tile_pack_kernel<EETQ_LAYOUT_GFX950><<<grid, 256, 0, stream>>>(src, K, N, p);
    return check_hip(hipGetLastError(), "tile_pack_kernel launch");
}

}  // namespace

// per-column max |w| (fp32, NaN ignored like std::max) into colmax[N]; shared by the int8 and int4 quantisers
int launch_colmax(const void* w, int w_dtype, size_t K, size_t N, float* colmax, hipStream_t stream)
{
    EETQ_REQUIRE(w && colmax, "null pointer");
    EETQ_REQUIRE(N % 8 == 0, "the number of columns (N) must be a multiple of 8");
    EETQ_TRY_HIP(hipMemsetAsync(colmax, 0, N * sizeof(float), stream));
    // strips of 64 lanes x V columns, kRowsPerBlock rows per workgroup: (N / 512) x (K / 128) workgroups at fp16 (256 at 4096^2)
    const unsigned yb = (unsigned)((K + kRowsPerBlock - 1) / kRowsPerBlock);
    if (w_dtype == EETQ_DTYPE_F16)
        colmax_kernel<f16, 8, false><<<dim3((unsigned)((N + 511) / 512), yb), 64 * kColmaxWaves, 0, stream>>>(static_cast<const f16*>(w), K, N,
                                                                                      reinterpret_cast<u32*>(colmax));
    else
        colmax_kernel<float, 4, false><<<dim3((unsigned)((N + 255) / 256), yb), 64 * kColmaxWaves, 0, stream>>>(static_cast<const float*>(w), K, N,
                                                                                        reinterpret_cast<u32*>(colmax));
    return check_hip(hipGetLastError(), "colmax_kernel launch");
}

// floats of workspace the int8 quantiser needs: one row of N maxima per block of kRowsPerBlock weight rows
size_t quantize_workspace_floats(size_t K, size_t N) { return N * ((K + kRowsPerBlock - 1) / kRowsPerBlock); }

namespace {
template <typename T, int V>
int launch_quantize_typed(const T* w, size_t K, size_t N, int8_t* raw_out, int8_t* packed_out, int layout, void* scales,
                          int scales_f32, float* part, bool final_maxima, hipStream_t stream)
{
    unsigned P = (unsigned)((K + kRowsPerBlock - 1) / kRowsPerBlock);
    int      st;
    if (final_maxima) {
        // the caller's workspace holds N floats only (the contract of the first ABI revision): zero fill + atomicMax on the
        // bit patterns leave the FINAL maxima in row 0 -- same values, one memset node more, a few us slower
        st = launch_colmax(w, std::is_same<T, float>::value ? EETQ_DTYPE_F32 : EETQ_DTYPE_F16, K, N, part, stream);
        P  = 1;
    } else {
        colmax_kernel<T, V, true><<<dim3((unsigned)((N + 64 * V - 1) / (64 * V)), P), 64 * kColmaxWaves, 0, stream>>>(
            w, K, N, reinterpret_cast<u32*>(part));
        st = check_hip(hipGetLastError(), "colmax_kernel launch");
    }
    if (st != EETQ_OK) return st;
    // by default every pack workgroup reduces the P rows of maxima itself (two launches per call); EETQ_AMD_QUANT_FOLD=1
    // folds them into row 0 with a small launch of their own first (measured slower: one more launch costs more than the
    // reads it saves, profiles/r03_quant_sweep.txt)
    static const bool fold = [] {
        const char* e = tuning_env("EETQ_AMD_QUANT_FOLD");
        return e && *e == '1';
    }();
    unsigned rows = P;
    if (fold && P > 1) {
        colmax_fold_kernel<<<(unsigned)((N + 255) / 256), 256, 0, stream>>>(part, N, (int)P);
        st = check_hip(hipGetLastError(), "colmax_fold_kernel launch");
        if (st != EETQ_OK) return st;
        rows = 1;
    }
    uint8_t*   p    = reinterpret_cast<uint8_t*>(packed_out);
    const bool sm80 = layout == EETQ_LAYOUT_SM80 && p;
    // tiles per workgroup: 1 (most workgroups in flight) for short K, 4 once the P rows of maxima a workgroup reduces
    // outweigh one tile (K > 8192: 13824 x 5120 94 vs 113 us); EETQ_AMD_QUANT_STRIP = 1 / 2 / 4 overrides (tuning hook)
    static const int forced = [] {
        const char* e = tuning_env("EETQ_AMD_QUANT_STRIP");
        const int   v = e ? atoi(e) : 0;
        return v == 1 || v == 2 || v == 4 ? v : 0;
    }();
    const int strip = forced ? forced : (K > 8192 ? 4 : 1);
    static const bool nt = [] {
        const char* e = tuning_env("EETQ_AMD_QUANT_NT");
        return e && *e == '1';
    }();
    // native layout, no row-major copy asked for: the column-major kernel (EETQ_AMD_QUANT_KERNEL=strip keeps the older one
    // for A/B runs); it indexes with 32-bit row offsets
    static const bool old_kernel = [] {
        const char* e = tuning_env("EETQ_AMD_QUANT_KERNEL");
        return e && e[0] == 's';
    }();
    const bool colmajor = p && !sm80 && !raw_out && !old_kernel && layout == EETQ_LAYOUT_GFX950 && rows * N < 0xffffffffull &&
                          (size_t)2 * kQT * N * sizeof(T) < 0x7fffffffull && K < 0x7fffffffull && N < 0x7fffffffull;
    if (colmajor) {
        // tiles per workgroup: about 512 workgroups (2 per CU: the per-workgroup reduction of the maxima is paid fewer times,
        // measured 20.1 vs 20.9 us at 4096^2 against 2048 workgroups of 2 tiles), never fewer than two tiles each (one tile =
        // no overlap of loads and arithmetic), at most 16; EETQ_AMD_QUANT_TILES overrides (tuning hook)
        static const int forced_tiles = [] {
            const char* e = tuning_env("EETQ_AMD_QUANT_TILES");
            const int   v = e ? atoi(e) : 0;
            return v >= 1 && v <= 64 ? v : 0;
        }();
        const size_t strips = (N + kQT - 1) / kQT, KT = K / kQT;
        size_t       tpw    = (strips * KT + 511) / 512;
        tpw                 = tpw < 2 ? 2 : (tpw > 16 ? 16 : tpw);
        if (forced_tiles) tpw = (size_t)forced_tiles;
        if (tpw > KT) tpw = KT;
        while (tpw > 1 && (tpw + 1) * kQT * N * sizeof(T) >= 0x7fffffffull) --tpw;  // 32-bit offsets inside a workgroup's rows
        const dim3 grid((unsigned)strips, (unsigned)((KT + tpw - 1) / tpw));
        quant_pack_kernel<T><<<grid, 256, 2 * QpCfg<T>::kTileLds, stream>>>(w, (unsigned)K, (unsigned)N, part, (int)rows, p, scales, scales_f32, (int)tpw);
        return check_hip(hipGetLastError(), "quant_pack_kernel launch");
    }
    auto launch = [&](auto tiles) {
        constexpr int TT = decltype(tiles)::value;
        const dim3    grid((unsigned)((N + kQT - 1) / kQT), (unsigned)((K / kQT + TT - 1) / TT));
        if (sm80)
            strip_quant_kernel<T, EETQ_LAYOUT_SM80, TT><<<grid, 256, 0, stream>>>(w, K, N, part, (int)rows, raw_out, p, scales,
                                                                                 scales_f32);
        else if (nt)
            strip_quant_kernel<T, EETQ_LAYOUT_GFX950, TT, true><<<grid, 256, 0, stream>>>(w, K, N, part, (int)rows, raw_out, p,
                                                                                         scales, scales_f32);
        else
            strip_quant_kernel<T, EETQ_LAYOUT_GFX950, TT><<<grid, 256, 0, stream>>>(w, K, N, part, (int)rows, raw_out, p, scales,
                                                                                   scales_f32);
    };
    if (strip == 4)
        launch(std::integral_constant<int, 4>{});
    else if (strip == 2)
        launch(std::integral_constant<int, 2>{});
    else
        launch(std::integral_constant<int, 1>{});
    return check_hip(hipGetLastError(), "strip_quant_kernel launch");
}
}  // namespace

// `workspace`: `workspace_floats` floats.  >= quantize_workspace_floats(K, N): row-block maxima (no atomics, no zero fill),
// then quantise + pack (each pack workgroup folds the rows of maxima of its columns).  >= N only (what the first revision
// of the ABI asked for): zero fill + atomicMax maxima, then quantise + pack.  Fewer: EETQ_ERR_INVALID, nothing launched.
int launch_quantize(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                    void* scales, float* workspace, size_t workspace_floats, hipStream_t stream)
{
    int st = check_layout_shape(K, N, q_packed ? layout : EETQ_LAYOUT_ROW_MAJOR);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(w && scales && workspace, "null pointer");
    EETQ_REQUIRE(w_dtype == EETQ_DTYPE_F16 || w_dtype == EETQ_DTYPE_F32,
                 "Invalid datatype. Weight must be FP16 or FP32");
    EETQ_REQUIRE(workspace_floats >= N, "quantise: the workspace holds fewer than N floats");
    const bool final_maxima = workspace_floats < quantize_workspace_floats(K, N);
    // ROW_MAJOR "packed" output is just the raw tensor again
    int8_t* raw_out    = q_raw;
    int8_t* packed_out = q_packed;
    int8_t* raw_copy   = nullptr;
    if (layout == EETQ_LAYOUT_ROW_MAJOR) {
        if (!raw_out)
            raw_out = q_packed;
        else if (q_packed && q_packed != raw_out)
            raw_copy = q_packed;
        packed_out = nullptr;
    }
    if (w_dtype == EETQ_DTYPE_F16)
        st = launch_quantize_typed<f16, 8>(static_cast<const f16*>(w), K, N, raw_out, packed_out, layout, scales, 0, workspace,
                                           final_maxima, stream);
    else
        st = launch_quantize_typed<float, 4>(static_cast<const float*>(w), K, N, raw_out, packed_out, layout, scales, 1,
                                             workspace, final_maxima, stream);
    if (st != EETQ_OK) return st;
    if (raw_copy) EETQ_TRY_HIP(hipMemcpyAsync(raw_copy, raw_out, K * N, hipMemcpyDeviceToDevice, stream));
    return EETQ_OK;
}

// int4, native layout, no row-major copy: quantise + pack in one launch from the FINAL column maxima (launch_colmax).
// EETQ_ERR_UNSUPPORTED (no message) when the 32-bit offsets of the kernel do not cover the tensor: the caller takes the
// three-kernel route of int4.hip.
int launch_quantize_pack_i4_native(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_packed, void* scales,
                                   const float* colmax, hipStream_t stream)
{
    if (K % 128 != 0 || N % 16 != 0 || K >= 0x7fffffffull || N >= 0x7fffffffull) return EETQ_ERR_UNSUPPORTED;
    const size_t esz    = w_dtype == EETQ_DTYPE_F16 ? 2 : 4;
    const size_t strips = (N + kQT - 1) / kQT, KT = K / 128;
    size_t       tpw    = (strips * KT + 511) / 512;  // about 512 workgroups, two to sixteen tiles each (as the int8 kernel)
    tpw                 = tpw < 2 ? 2 : (tpw > 16 ? 16 : tpw);
    if (tpw > KT) tpw = KT;
    while (tpw > 1 && (tpw + 1) * 128 * N * esz >= 0x7fffffffull) --tpw;
    if ((size_t)2 * 128 * N * esz >= 0x7fffffffull) return EETQ_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)strips, (unsigned)((KT + tpw - 1) / tpw));
    uint8_t*   p = reinterpret_cast<uint8_t*>(q_packed);
    if (w_dtype == EETQ_DTYPE_F16) {
        quant_pack_kernel<f16, 4><<<grid, 256, 2 * QpCfg<f16, 4>::kTileLds, stream>>>(static_cast<const f16*>(w), (unsigned)K, (unsigned)N,
                                                                                   colmax, 1, p, scales, 0, (int)tpw);
    } else {
        // 66.5 KiB of dynamic LDS: the per-kernel opt-in, for exactly that much (the kernel also has 1.5 KiB of static LDS, so
        // common.hpp's "whole CU" request is refused for it)
        static std::atomic<unsigned long long> opted{0};
        int dev = 0;
        EETQ_TRY_HIP(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(opted.load(std::memory_order_relaxed) & bit)) {
            EETQ_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(quant_pack_kernel<float, 4>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * QpCfg<float, 4>::kTileLds));
            opted.fetch_or(bit, std::memory_order_relaxed);
        }
        quant_pack_kernel<float, 4><<<grid, 256, 2 * QpCfg<float, 4>::kTileLds, stream>>>(static_cast<const float*>(w), (unsigned)K,
                                                                                       (unsigned)N, colmax, 1, p, scales, 1, (int)tpw);
    }
    return check_hip(hipGetLastError(), "quant_pack_kernel (int4) launch");
}

int launch_pack(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, hipStream_t stream)
{
    int st = check_layout_shape(K, N, layout);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(q_raw && q_packed && q_raw != q_packed, "null or aliased pointer");
    if (layout == EETQ_LAYOUT_ROW_MAJOR) {
        EETQ_TRY_HIP(hipMemcpyAsync(q_packed, q_raw, K * N, hipMemcpyDeviceToDevice, stream));
        return EETQ_OK;
    }
    return launch_tile_pack(q_raw, K, N, q_packed, layout, stream);
}

int launch_unpack(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, hipStream_t stream)
{
    int st = check_layout_shape(K, N, layout);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(q_raw && q_packed && q_raw != q_packed, "null or aliased pointer");
    if (layout == EETQ_LAYOUT_ROW_MAJOR) {
        EETQ_TRY_HIP(hipMemcpyAsync(q_raw, q_packed, K * N, hipMemcpyDeviceToDevice, stream));
        return EETQ_OK;
    }
    dim3           grid((unsigned)((N + kQT - 1) / kQT), (unsigned)(K / kQT));
    const uint8_t* p = reinterpret_cast<const uint8_t*>(q_packed);
    if (layout == EETQ_LAYOUT_SM80)
        tile_unpack_kernel<EETQ_LAYOUT_SM80><<<grid, 256, 0, stream>>>(p, K, N, q_raw);
    else
        tile_unpack_kernel<EETQ_LAYOUT_GFX950><<<grid, 256, 0, stream>>>(p, K, N, q_raw);
    return check_hip(hipGetLastError(), "tile_unpack_kernel launch");
}

}  // namespace eetq
