// Per-channel symmetric int8 quantiser and weight re-layout kernels (plain HIP, HBM-bound byte work).
//
// Reference behaviour restated (paths relative to /root/reference):
//   quantise  csrc/cutlass_kernels/cutlass_preprocessors.cc:581-678  (ft::symmetric_quantize)
//   sm80 pack csrc/cutlass_kernels/cutlass_preprocessors.cc:497-534  (preprocess_weights_for_mixed_gemm)
// The reference runs these single-threaded on the CPU (three strided K x N passes + four re-layout
// passes); here quant_weights is three launches -- per-row-block column maxima (plain stores: no atomics, no zero fill), a
// small fold of those rows, and a kernel that quantises and writes the target layout -- and every pass is one coalesced sweep: 64(k) x 64(n) tiles are read
// row-wise, transposed through LDS and written in the destination layout with 16-byte stores.
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
// Workgroup = 4 waves over one strip of 64*V columns (one contiguous 1 KiB per wave-load) x kRowsPerBlock rows: wave j
// takes rows j, j+4, ...; eight independent 16-byte loads in flight per lane; the four waves' maxima meet in LDS and ONE
// atomicMax per column and workgroup follows (K / 128 per column in total -- the first version issued one per column per 32
// rows, 524 k atomics at 4096^2, and ran at 1.2 TB/s).
// PARTIALS: no atomics and no zero-fill launch before the kernel -- row block y stores its maxima to row y of a
// [ceil(K / kRowsPerBlock)][N] array and the pack kernel reduces the rows for its 64 columns (the int8 quantiser: two
// launches per call instead of fill + maxima + pack).
constexpr int kRowsPerBlock = 128;
template <typename T, int V, bool PARTIALS>
__global__ __launch_bounds__(256) void colmax_kernel(const T* __restrict__ w, size_t K, size_t N,
                                                     u32* __restrict__ colmax_bits)
{
    __shared__ float part[4][64 * V];
    const int    wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t col0 = ((size_t)blockIdx.x * 64 + lane) * V;
    const size_t k_begin = (size_t)blockIdx.y * kRowsPerBlock;
    size_t       k_end   = k_begin + kRowsPerBlock;
    if (k_end > K) k_end = K;
    const bool   live = col0 < N;
    const size_t cc   = live ? col0 : 0;  // dead lanes (ragged last strip) read column 0 and are ignored
    float m[V];
#pragma unroll
    for (int i = 0; i < V; ++i) m[i] = 0.f;
    constexpr int kBatch = 8;
    for (size_t k = k_begin + wave; k < k_end; k += 4 * kBatch) {
        u32x4 raw[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const size_t kk = k + 4 * j < k_end ? k + 4 * j : k_end - 1;  // clamped, never a branch around the load
            raw[j]          = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + kk * N + cc));
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            T v[V];
            *reinterpret_cast<u32x4*>(v) = raw[j];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float a = __builtin_fabsf((float)v[i]);
                m[i]          = (m[i] < a) ? a : m[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) part[wave][lane * V + i] = m[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * V; c += 256) {
        const size_t col = (size_t)blockIdx.x * 64 * V + c;
        if (col < N) {
            float a = part[0][c];
#pragma unroll
            for (int j = 1; j < 4; ++j) a = (a < part[j][c]) ? part[j][c] : a;
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
    __shared__ __attribute__((aligned(16))) float cmx[kQT];
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
        if (kt0 == 0 && n0 + t < N && scales) {  // :633-634 scale = T(colmax * 2^-7), written once per column
            const float s32 = a * (1.f / 128.f);
            if (scales_f32)
                reinterpret_cast<float*>(scales)[n0 + t] = s32;
            else
                reinterpret_cast<f16*>(scales)[n0 + t] = (f16)s32;
        }
    }
    __syncthreads();
    float s[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<f32x4*>(s)[i] = reinterpret_cast<const f32x4*>(cmx + seg * 16)[i];
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
#pragma unroll
            for (int i = 0; i < 16; ++i) out.b[i] = quantize_elt((float)v[i], s[i]);
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
        colmax_kernel<f16, 8, false><<<dim3((unsigned)((N + 511) / 512), yb), 256, 0, stream>>>(static_cast<const f16*>(w), K, N,
                                                                                      reinterpret_cast<u32*>(colmax));
    else
        colmax_kernel<float, 4, false><<<dim3((unsigned)((N + 255) / 256), yb), 256, 0, stream>>>(static_cast<const float*>(w), K, N,
                                                                                        reinterpret_cast<u32*>(colmax));
    return check_hip(hipGetLastError(), "colmax_kernel launch");
}

// floats of workspace the int8 quantiser needs: one row of N maxima per block of kRowsPerBlock weight rows
size_t quantize_workspace_floats(size_t K, size_t N) { return N * ((K + kRowsPerBlock - 1) / kRowsPerBlock); }

namespace {
template <typename T, int V>
int launch_quantize_typed(const T* w, size_t K, size_t N, int8_t* raw_out, int8_t* packed_out, int layout, void* scales,
                          int scales_f32, float* part, hipStream_t stream)
{
    const unsigned P = (unsigned)((K + kRowsPerBlock - 1) / kRowsPerBlock);
    colmax_kernel<T, V, true><<<dim3((unsigned)((N + 64 * V - 1) / (64 * V)), P), 256, 0, stream>>>(
        w, K, N, reinterpret_cast<u32*>(part));
    int st = check_hip(hipGetLastError(), "colmax_kernel launch");
    if (st != EETQ_OK) return st;
    // by default every pack workgroup reduces the P rows of maxima itself (two launches per call); EETQ_AMD_QUANT_FOLD=1
    // folds them into row 0 with a small launch of their own first (measured slower: one more launch costs more than the
    // reads it saves, profiles/r03_quant_sweep.txt)
    static const bool fold = [] {
        const char* e = getenv("EETQ_AMD_QUANT_FOLD");
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
        const char* e = getenv("EETQ_AMD_QUANT_STRIP");
        const int   v = e ? atoi(e) : 0;
        return v == 1 || v == 2 || v == 4 ? v : 0;
    }();
    const int strip = forced ? forced : (K > 8192 ? 4 : 1);
    static const bool nt = [] {
        const char* e = getenv("EETQ_AMD_QUANT_NT");
        return e && *e == '1';
    }();
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

// `workspace`: quantize_workspace_floats(K, N) floats.  Launches: row-block maxima (no atomics, no zero fill), the fold of
// those rows, quantise + pack.
int launch_quantize(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                    void* scales, float* workspace, hipStream_t stream)
{
    int st = check_layout_shape(K, N, q_packed ? layout : EETQ_LAYOUT_ROW_MAJOR);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(w && scales && workspace, "null pointer");
    EETQ_REQUIRE(w_dtype == EETQ_DTYPE_F16 || w_dtype == EETQ_DTYPE_F32,
                 "Invalid datatype. Weight must be FP16 or FP32");
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
                                           stream);
    else
        st = launch_quantize_typed<float, 4>(static_cast<const float*>(w), K, N, raw_out, packed_out, layout, scales, 1,
                                             workspace, stream);
    if (st != EETQ_OK) return st;
    if (raw_copy) EETQ_TRY_HIP(hipMemcpyAsync(raw_copy, raw_out, K * N, hipMemcpyDeviceToDevice, stream));
    return EETQ_OK;
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
