// W4A16 (int4 weight-only) path: quantise, pack / unpack, and the GEMM dispatch (decode GEMV on int4 tiles for M <= 4 --
// gemv_kernel.hpp with BITS = 4 -- and the route to the int8 kernels for larger M).
//
// Reference behaviour restated (paths relative to /root/reference):
//   quantise      csrc/cutlass_kernels/cutlass_preprocessors.cc:581-678 with PACKED_INT4_WEIGHT_ONLY (scale = amax / 8,
//                 q = clamp(int(round(w / scale)), -8, 7), two values per byte along N, even column in the low nibble)
//   sm80 layout   cutlass_preprocessors.cc:137-195 (32-row permute), :201-335 (sub-byte transpose), :432-495 (column
//                 interleave, 4 columns x 64 rows), :360-418 (+8 bias, nibble interleave inside each 32-bit register)
//   GEMV          csrc/weightOnlyBatchedGemv/kernel.h:68-116 (Int4b details) + the int4 -> fp16 converter
//                 (cutlass_extensions/.../interleaved_numeric_conversion.h:215-280): (w & 0x000f000f) | 0x64006400 etc.
// The reference binds only the quantise / preprocess half of int4 (fpA_intB_gemm_wrapper.cu:41-66, 109-128); its
// w8_a16_gemm is hard-wired to int8 (:154-171).  The GEMM here is the extension SURVEY.md 8f row 4 asks for.
//
// Native layout (gfx950, int4): 1 KiB tiles of 16 columns x 128 k, tiles ordered [n/16][k/128]; lane = ((k >> 5) & 3) * 16 +
// (n & 15) holds the 32 k values of its column in 16 bytes; dword d holds k = 8d .. 8d+7 as unsigned nibbles q + 8 at nibble
// positions [0, 4, 1, 5, 2, 6, 3, 7], so the four mask / shift extractions of the reference's converter yield the fp16 pairs
// (k0,k1), (k2,k3), (k4,k5), (k6,k7) that v_dot2 wants.  quant_weights in the native layout is two launches + a fill (column maxima,
// then quant.hip's fused quantise + pack kernel); the other routes (sm80 layout, row-major copy, pack / unpack alone) are plain
// one-thread-per-byte gathers, not tuned.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.hpp"

namespace eetq {

namespace {

// ---- index maps -------------------------------------------------------------------------------------------------------
// nibble position of raw element (k, n) in the gfx950 int4 stream
__device__ __forceinline__ size_t gfx950_i4_pos(size_t k, size_t n, size_t K)
{
    const size_t tile = (n >> 4) * (K >> 7) + (k >> 7);
    const size_t lane = ((k >> 5) & 3) * 16 + (n & 15);
    const size_t d = (k >> 3) & 3, j = k & 7;
    return (tile * 1024 + lane * 16 + d * 4) * 2 + ((j >> 1) + 4 * (j & 1));
}
// inverse: nibble position -> (k, n)
__device__ __forceinline__ void gfx950_i4_inv(size_t pos, size_t K, size_t& k, size_t& n)
{
    const size_t nib = pos & 7, dw = pos >> 3;             // nibble inside its dword, dword index
    const size_t d = dw & 3, lane = (dw >> 2) & 63, tile = dw >> 8;
    const size_t j = nib < 4 ? 2 * nib : 2 * (nib - 4) + 1;
    const size_t ktiles = K >> 7;
    k = (tile % ktiles) * 128 + (lane >> 4) * 32 + d * 8 + j;
    n = (tile / ktiles) * 16 + (lane & 15);
}
// reference layout (closed form of P1..P4 for int4): written row kw <- source row perm32(kw);
// stream index P = (n/4)*4K + (kw/64)*256 + (n%4)*64 + kw%64; inside an aligned group of 8 nibbles source e sits at dest
// (e even ? e/2 : 4 + e/2)
__device__ __forceinline__ size_t perm32(size_t kw)
{
    const size_t t = kw & 31;
    return kw - t + (8 * ((t & 7) >> 1) + (t & 1) + 2 * (t >> 3));
}
__device__ __forceinline__ size_t perm32_inv(size_t ks)
{
    // t -> r = 8a + b + 2c with a = (t%8)/2, b = t%2, c = t/8;  inverse: a = r/8, b = r%2, c = (r%8)/2 -> t = 8c + 2a + b
    const size_t r = ks & 31;
    return ks - r + (8 * ((r & 7) >> 1) + 2 * (r >> 3) + (r & 1));
}
__device__ __forceinline__ size_t sm80_i4_pos(size_t k, size_t n, size_t K)
{
    const size_t kw = perm32_inv(k);
    const size_t P  = (n >> 2) * 4 * K + (kw >> 6) * 256 + (n & 3) * 64 + (kw & 63);
    const size_t e  = P & 7;
    return (P - e) + ((e & 1) ? 4 + (e >> 1) : (e >> 1));
}
__device__ __forceinline__ void sm80_i4_inv(size_t pos, size_t K, size_t& k, size_t& n)
{
    const size_t d = pos & 7, e = d < 4 ? 2 * d : 2 * (d - 4) + 1;
    const size_t P = (pos - d) + e;
    const size_t col4 = P / (4 * K), r = P % (4 * K);
    const size_t kw = (r >> 8) * 64 + (r & 63);
    n = col4 * 4 + ((r >> 6) & 3);
    k = perm32(kw);
}

__device__ __forceinline__ unsigned raw_nibble(const uint8_t* raw, size_t k, size_t n, size_t N)
{
    const uint8_t b = raw[k * (N >> 1) + (n >> 1)];
    return (n & 1) ? (b >> 4) : (b & 0xF);
}

// ---- quantise: one thread per raw byte (two adjacent columns of one row) ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void quant_i4_kernel(const T* __restrict__ w, size_t K, size_t N, const float* __restrict__ colmax,
                                                       uint8_t* __restrict__ q_raw, void* __restrict__ scales, int scales_f32)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t half = N >> 1;
    if (idx >= K * half) return;
    const size_t k = idx / half, jb = idx % half;
    unsigned     out = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const size_t n = 2 * jb + p;
        const float  s = colmax[n] * (1.f / 8.f);                    // :610, :633: fp32 scale
        const float  scaled = __builtin_roundf((float)w[k * N + n] / s);   // IEEE divide, round half away
        // int(scaled) then clamp (:660-661); int(NaN) / out-of-range is INT_MIN on the reference's x86 hosts
        int iw;
        if (scaled != scaled || scaled >= 2147483648.f || scaled < -2147483648.f) iw = (int)0x80000000;
        else iw = (int)scaled;
        const int c = iw < -8 ? -8 : (iw > 7 ? 7 : iw);
        out |= (unsigned)(c & 0xF) << (4 * p);
        if (k == 0 && scales) {
            if (scales_f32) reinterpret_cast<float*>(scales)[n] = s;
            else reinterpret_cast<f16*>(scales)[n] = (f16)s;
        }
    }
    q_raw[idx] = (uint8_t)out;
}

// ---- pack: one thread per OUTPUT byte (two nibble positions, each gathered from the raw tensor) -------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void pack_i4_kernel(const uint8_t* __restrict__ raw, size_t K, size_t N, uint8_t* __restrict__ out)
{
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= K * N / 2) return;
    unsigned v = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        size_t k, n;
        if constexpr (LAYOUT == EETQ_LAYOUT_GFX950) gfx950_i4_inv(2 * b + p, K, k, n);
        else sm80_i4_inv(2 * b + p, K, k, n);
        const unsigned q = raw_nibble(raw, k, n, N);          // two's complement nibble
        v |= ((q + 8) & 0xF) << (4 * p);                      // + 8: unsigned 0..15
    }
    out[b] = (uint8_t)v;
}

// ---- unpack: one thread per RAW byte -------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void unpack_i4_kernel(const uint8_t* __restrict__ packed, size_t K, size_t N, uint8_t* __restrict__ raw)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t half = N >> 1;
    if (idx >= K * half) return;
    const size_t k = idx / half, jb = idx % half;
    unsigned     v = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const size_t n = 2 * jb + p;
        const size_t pos = LAYOUT == EETQ_LAYOUT_GFX950 ? gfx950_i4_pos(k, n, K) : sm80_i4_pos(k, n, K);
        const unsigned u = (packed[pos >> 1] >> (4 * (pos & 1))) & 0xF;
        v |= ((u - 8) & 0xF) << (4 * p);
    }
    raw[idx] = (uint8_t)v;
}

int check_i4_shape(size_t K, size_t N, int layout)
{
    EETQ_REQUIRE(K > 0 && N > 0, "weight should not be empty tensor");
    EETQ_REQUIRE(N % 2 == 0, "int4: the number of columns (N) must be even");
    if (layout == EETQ_LAYOUT_GFX950) {
        EETQ_REQUIRE(K % 128 == 0, "int4: the number of rows (K) must be a multiple of 128");
        EETQ_REQUIRE(N % 16 == 0, "the number of columns (N) must be a multiple of 16");
    } else if (layout == EETQ_LAYOUT_SM80) {
        EETQ_REQUIRE(K % 64 == 0, "the number of rows (K) of the quantized matrix must be a multiple of 64");
        EETQ_REQUIRE(N % 64 == 0, "The number of columns must be a multiple of 64 (sm80 layout)");
    } else {
        EETQ_REQUIRE(layout == EETQ_LAYOUT_ROW_MAJOR, "unknown weight layout");
    }
    return EETQ_OK;
}

// ---- expand int4 tiles to int8 gfx950 tiles (same integers, same scales): the route to the MFMA kernels for M > 4 --------
// one thread per int4 lane chunk (16 bytes = 32 k of one column) -> two int8 lane chunks (16 k each) of one int8 tile
__global__ __launch_bounds__(256) void expand_i4_to_i8_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t chunks,
                                                              size_t ktiles4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunks) return;
    const u32x4  v = src[i];
    const size_t lane = i & 63, tile4 = i >> 6, g = lane >> 4, c = lane & 15;
    const size_t ntile = tile4 / ktiles4, kt4 = tile4 % ktiles4;
    const size_t tile8 = ntile * (2 * ktiles4) + 2 * kt4 + (g >> 1);   // 64-deep int8 tile holding k = 32g .. 32g+31
    const u32    wd[4] = {v.x, v.y, v.z, v.w};
    u32          o[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        // nibble p of the dword holds k-local j(p): p = 0..3 -> j = 0, 2, 4, 6 ; p = 4..7 -> j = 1, 3, 5, 7
        unsigned u[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) u[p < 4 ? 2 * p : 2 * (p - 4) + 1] = ((wd[d] >> (4 * p)) & 0xF) + 120;  // (q4 + 8) + 120 = q + 128
        // int8 dword byte order [k0, k2, k1, k3]
        o[2 * d]     = u[0] | (u[2] << 8) | (u[1] << 16) | (u[3] << 24);
        o[2 * d + 1] = u[4] | (u[6] << 8) | (u[5] << 16) | (u[7] << 24);
    }
    const size_t lane8a = ((2 * g) & 3) * 16 + c, lane8b = ((2 * g + 1) & 3) * 16 + c;
    dst[tile8 * 64 + lane8a] = u32x4{o[0], o[1], o[2], o[3]};
    dst[tile8 * 64 + lane8b] = u32x4{o[4], o[5], o[6], o[7]};
}

// Scratch for the expanded weight: one buffer per (device, stream) -- launches that run concurrently on one device come from
// different streams and must not share it -- grown on demand (creating or growing one is not capturable).  kStreamsPerDevice
// streams per device get their own; further streams share the last slot (their calls are then serialised by a device
// synchronisation whenever the slot changes hands).
struct Scratch {
    uint8_t*    p      = nullptr;
    size_t      bytes  = 0;
    hipStream_t stream = nullptr;
    bool        used   = false;
};
constexpr int kStreamsPerDevice = 8;
std::mutex    g_mutex;
Scratch       g_scratch[64][kStreamsPerDevice];

int expanded_scratch(size_t bytes, hipStream_t stream, uint8_t** out)
{
    int dev = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mutex);
    Scratch* slots = g_scratch[dev & 63];
    Scratch* s     = nullptr;
    for (int i = 0; i < kStreamsPerDevice && !s; ++i)
        if (slots[i].used && slots[i].stream == stream) s = &slots[i];
    for (int i = 0; i < kStreamsPerDevice && !s; ++i)
        if (!slots[i].used) s = &slots[i];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (!s) {  // more streams than slots: the last slot changes hands (whoever used it may still be reading it)
        if (capturing)
            return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] W4A16 prefill: too many streams on this device for graph capture");
        s = &slots[kStreamsPerDevice - 1];
        EETQ_TRY_HIP(hipDeviceSynchronize());
    }
    if (s->bytes < bytes) {
        if (capturing)
            return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] the W4A16 prefill scratch cannot grow during graph capture: run "
                                              "the shape once eagerly first");
        EETQ_TRY_HIP(hipDeviceSynchronize());  // earlier launches may still read the old buffer
        if (s->p) EETQ_TRY_HIP(hipFree(s->p));
        s->p     = nullptr;
        s->bytes = 0;
        EETQ_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&s->p), bytes));
        s->bytes = bytes;
    }
    s->used   = true;
    s->stream = stream;
    *out      = s->p;
    return EETQ_OK;
}

}  // namespace

int release_w4a16_workspace(size_t* freed)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (int d = 0; d < 64; ++d)
        for (Scratch& s : g_scratch[d]) {
            if (s.p && hipSetDevice(d) == hipSuccess) {
                (void)hipDeviceSynchronize();
                (void)hipFree(s.p);
                if (freed) *freed += s.bytes;
            }
            s = Scratch{};
        }
    (void)hipSetDevice(keep);
    return EETQ_OK;
}

int launch_quantize_i4(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                       void* scales, float* colmax, hipStream_t stream)
{
    int st = check_i4_shape(K, N, q_packed ? layout : EETQ_LAYOUT_ROW_MAJOR);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(w && scales && colmax, "null pointer");
    EETQ_REQUIRE(w_dtype == EETQ_DTYPE_F16 || w_dtype == EETQ_DTYPE_F32, "Invalid datatype. Weight must be FP16 or FP32");
    EETQ_REQUIRE(N % 8 == 0, "the number of columns (N) must be a multiple of 8");
    EETQ_REQUIRE(q_raw || q_packed, "null pointer");
    st = launch_colmax(w, w_dtype, K, N, colmax, stream);
    if (st != EETQ_OK) return st;
    // native layout and no row-major copy asked for: one fused quantise + pack launch (quant.hip::quant_pack_kernel<T, 4>:
    // LDS-DMA ring, column-major in registers, division-free but bit-exact) instead of quantise -> temporary -> gather
    if (q_packed && !q_raw && layout == EETQ_LAYOUT_GFX950) {
        st = launch_quantize_pack_i4_native(w, w_dtype, K, N, q_packed, scales, colmax, stream);
        if (st != EETQ_ERR_UNSUPPORTED) return st;
    }
    // the raw tensor is the pack kernel's source: the caller's buffer when it wants one, the output itself for ROW_MAJOR,
    // else a stream-ordered temporary
    uint8_t* raw = reinterpret_cast<uint8_t*>(q_raw);
    void*    tmp = nullptr;
    if (!raw) {
        if (layout == EETQ_LAYOUT_ROW_MAJOR) raw = reinterpret_cast<uint8_t*>(q_packed);
        else {
            EETQ_TRY_HIP(hipMallocAsync(&tmp, K * N / 2, stream));
            raw = static_cast<uint8_t*>(tmp);
        }
    }
    const unsigned blocks = (unsigned)((K * N / 2 + 255) / 256);
    if (w_dtype == EETQ_DTYPE_F16)
        quant_i4_kernel<f16><<<blocks, 256, 0, stream>>>(static_cast<const f16*>(w), K, N, colmax, raw, scales, 0);
    else
        quant_i4_kernel<float><<<blocks, 256, 0, stream>>>(static_cast<const float*>(w), K, N, colmax, raw, scales, 1);
    st = check_hip(hipGetLastError(), "quant_i4_kernel launch");
    if (st == EETQ_OK && q_packed && reinterpret_cast<uint8_t*>(q_packed) != raw) {
        if (layout == EETQ_LAYOUT_ROW_MAJOR)
            st = check_hip(hipMemcpyAsync(q_packed, raw, K * N / 2, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
        else
            st = launch_pack_i4(reinterpret_cast<const int8_t*>(raw), K, N, q_packed, layout, stream);
    }
    if (tmp) (void)hipFreeAsync(tmp, stream);
    return st;
}

int launch_pack_i4(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, hipStream_t stream)
{
    int st = check_i4_shape(K, N, layout);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(q_raw && q_packed && q_raw != q_packed, "null or aliased pointer");
    const uint8_t* src = reinterpret_cast<const uint8_t*>(q_raw);
    uint8_t*       dst = reinterpret_cast<uint8_t*>(q_packed);
    if (layout == EETQ_LAYOUT_ROW_MAJOR) return check_hip(hipMemcpyAsync(dst, src, K * N / 2, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
    const unsigned blocks = (unsigned)((K * N / 2 + 255) / 256);
    if (layout == EETQ_LAYOUT_SM80) pack_i4_kernel<EETQ_LAYOUT_SM80><<<blocks, 256, 0, stream>>>(src, K, N, dst);
    else pack_i4_kernel<EETQ_LAYOUT_GFX950><<<blocks, 256, 0, stream>>>(src, K, N, dst);
    return check_hip(hipGetLastError(), "pack_i4_kernel launch");
}

int launch_unpack_i4(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, hipStream_t stream)
{
    int st = check_i4_shape(K, N, layout);
    if (st != EETQ_OK) return st;
    EETQ_REQUIRE(q_raw && q_packed && q_raw != q_packed, "null or aliased pointer");
    const uint8_t* src = reinterpret_cast<const uint8_t*>(q_packed);
    uint8_t*       dst = reinterpret_cast<uint8_t*>(q_raw);
    if (layout == EETQ_LAYOUT_ROW_MAJOR) return check_hip(hipMemcpyAsync(dst, src, K * N / 2, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
    const unsigned blocks = (unsigned)((K * N / 2 + 255) / 256);
    if (layout == EETQ_LAYOUT_SM80) unpack_i4_kernel<EETQ_LAYOUT_SM80><<<blocks, 256, 0, stream>>>(src, K, N, dst);
    else unpack_i4_kernel<EETQ_LAYOUT_GFX950><<<blocks, 256, 0, stream>>>(src, K, N, dst);
    return check_hip(hipGetLastError(), "unpack_i4_kernel launch");
}

// What W4A16 AUTO launches (one pure function; eetq_diag_auto_path shows it).  Reference: m <= 4 batched GEMV, else CUTLASS
// (fpA_intB_gemm_wrapper.cu:149-162).
int w4a16_auto_path(int M, int N, int K)
{
    // decode: the wave-reduction GEMV on int4 tiles (gemv_kernel<..., BITS = 4>) ...
    if (M == 1) {
        // ... or (round 4) the MFMA small-batch kernel run with ONE row (its activation row in LDS, streamk.hip::pick_plan_i4) where
        // it beats the dot-product GEMV.  The GEMV is VALU-bound (2.1 VALU operations per weight) and saturates near
        // 3.5-4.4 TB/s of int4 weights; the MFMA form leaves the adds to the matrix cores and reaches 5.1
        // (profiles/r04_i4_gemv_vs_stream_m1.txt, us GEMV / stream): 8192 x 28672 33.9 / 22.9, 8192^2 10.86 / 8.81, 28672 x 8192 26.7 /
        // 22.9, 5120 x 27648 18.3 / 15.7, 5120 x 15360 11.6 / 10.3, 8192 x 1024 5.54 / 4.94, 11008 x 4096 7.43 / 6.82.  Kept on the
        // GEMV: the ties and losses -- 4096^2 4.02 / 4.07, 5120^2 5.61 / 5.76, 5120 x 13824 10.15 / 10.20, 4096 x 14336 8.45 / 8.57,
        // 13824 x 5120 10.82 / 11.17 (1 < tile rows per CU < 2: the 8-column units balance those better).
        // EETQ_AMD_I4_M1=gemv|stream forces one (A/B runs; needs EETQ_AMD_TUNING=1).
        static const int forced = [] {
            const char* e = tuning_env("EETQ_AMD_I4_M1");
            return !e ? 0 : !strcmp(e, "gemv") ? 1 : !strcmp(e, "stream") ? 2 : 0;
        }();
        bool stream_form = false;
        if (K % 128 == 0 && K >= 4096) {
            const int  ncu  = device_cu_count();
            const int  rows = N / kTileN;
            const bool deep = K >= 8192 && !(rows > ncu && rows < 2 * ncu);
            const bool big  = (size_t)K * N >= (72ull << 20);
            stream_form     = forced ? forced == 2 : (deep || big);
        }
        return stream_form ? EETQ_PATH_STREAM : EETQ_PATH_GEMV;
    }
    // batched decode, 2 <= M <= 16: the register-streaming MFMA kernel on int4 tiles (streamk_kernel<..., BITS = 4>) -- the
    // weight stream bounds these M, and it is half as long as the int8 one (M = 8, N = K = 4096: 4.5 vs 5.1 us; the dot2
    // GEMV at M = 4 needs 6.9).  Larger M are bound by the activation traffic / the matrix cores, where int4 buys nothing.
    if (M <= 16) return EETQ_PATH_STREAM;
    // 17 <= M <= 128: the split-K MFMA tile on int4 weight tiles (gemm_splitk_kernel<..., BITS = 4>; round 4) -- no expansion
    // pass, no per-stream weight scratch, capturable into a HIP graph
    if (M <= kMidMaxM && (size_t)M * K * 2 < (1ull << 31) && (size_t)N * K / 2 < (1ull << 31)) return EETQ_PATH_SPLITK;
    return EETQ_PATH_MFMA;   // larger batches: the expansion route below
}

int launch_w4a16(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K, hipStream_t stream,
                 int path)
{
    // explicit kernel paths (tests, tuning): GEMV (M <= 4), STREAM (M <= 16), SPLITK (M <= 128, honours EETQ_AMD_SPLITK_PLAN),
    // MFMA (the expansion route); anything else but AUTO is refused
    if (path == EETQ_PATH_GEMV) return launch_gemv_i4(x, w, scales, ep, y, M, N, K, stream);
    if (path == EETQ_PATH_STREAM) {
        if (M > 16) return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] W4A16 stream kernel supports M <= 16");
        return launch_streamk_i4(x, w, scales, ep, y, M, N, K, stream);
    }
    if (path == EETQ_PATH_SPLITK) return launch_gemm_splitk_i4(x, w, scales, ep, y, M, N, K, stream, /*env_plan=*/true);
    if (path != EETQ_PATH_AUTO && path != EETQ_PATH_MFMA)
        return fail(EETQ_ERR_UNSUPPORTED, "[eetq_amd] W4A16: unknown or unimplemented GEMM path");
    if (path == EETQ_PATH_AUTO) path = w4a16_auto_path(M, N, K);
    if (path == EETQ_PATH_GEMV) return launch_gemv_i4(x, w, scales, ep, y, M, N, K, stream);
    if (path == EETQ_PATH_STREAM) return launch_streamk_i4(x, w, scales, ep, y, M, N, K, stream);
    if (path == EETQ_PATH_SPLITK) return launch_gemm_splitk_i4(x, w, scales, ep, y, M, N, K, stream);
    // larger batches: expand the nibbles to the int8 tile layout once per call (K*N/2 bytes read, K*N written; the GEMM that
    // follows is MFMA- or x-bound at these M) and run the W8A16 kernels on it -- same integers, same scales, same contract
    uint8_t* w8 = nullptr;
    int      st = expanded_scratch((size_t)K * N, stream, &w8);
    if (st != EETQ_OK) return st;
    const size_t chunks = (size_t)K * N / 2 / 16;
    launch_kernel(expand_i4_to_i8_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream,
                  reinterpret_cast<const u32x4*>(w), reinterpret_cast<u32x4*>(w8), chunks, (size_t)K >> 7);
    EETQ_TRY_HIP(hipGetLastError());
    return eetq_w8a16_gemm_act(x, reinterpret_cast<const int8_t*>(w8), scales, ep.bias, ep.residual, y, M, N, K, EETQ_PATH_AUTO,
                               ep.act, stream);
}

}  // namespace eetq
