// Shared device/host helpers for the gfx950 kernels.  gfx950 only: wave64, no other targets.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <string>

#include "../../include/eetq_amd.h"

namespace eetq {

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
// native layout constants (DESIGN.md "HBM layout")
constexpr int kTileN     = 16;    // output columns per tile
constexpr int kTileK     = 64;    // k per tile
constexpr int kTileBytes = 1024;  // one wave-wide 16 B/lane load

// ---- status / error plumbing (abi.hip owns the storage) -------------------------------------------
void set_error(const std::string& msg);
int  fail(int code, const std::string& msg);
int  check_hip(hipError_t e, const char* what);

#define EETQ_TRY_HIP(expr)                                   \
    do {                                                     \
        int _st = ::eetq::check_hip((expr), #expr);          \
        if (_st != EETQ_OK) return _st;                      \
    } while (0)

#define EETQ_REQUIRE(cond, msg)                                                          \
    do {                                                                                 \
        if (!(cond)) return ::eetq::fail(EETQ_ERR_INVALID, std::string("[eetq_amd] ") + msg); \
    } while (0)

// ---- device helpers ----------------------------------------------------------------------------------
__device__ __forceinline__ f16x2 as_f16x2(u32 v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ u32 as_u32(f16x2 v) { return __builtin_bit_cast(u32, v); }

// One dword of the native layout holds uint8 (q+128) for k-locals [0,2,1,3] (bytes 0..3).
// v_perm_b32 builds 0x64bb64bb = fp16 pair (1024+b_lo, 1024+b_hi); subtracting 1152 gives the exact
// integer q (reference: interleaved_numeric_conversion.h:53-85 does the same with prmt + sub.f16x2);
// the product with the fp16 scale is rounded once to fp16 (mma_tensorop_dequantizer.h:259-274).
// v_perm_b32 byte selectors: 0-3 pick bytes of src1, 4-7 bytes of src0.
__device__ __forceinline__ void dequant_dword(u32 w, f16x2 scale2, f16x2& k01, f16x2& k23)
{
    const u32   c64  = 0x64646464u;
    const u32   lo   = __builtin_amdgcn_perm(w, c64, 0x00060004u);  // bytes [w.b0, 0x64, w.b2, 0x64]
    const u32   hi   = __builtin_amdgcn_perm(w, c64, 0x00070005u);  // bytes [w.b1, 0x64, w.b3, 0x64]
    const f16x2 bias = {(f16)1152.0f, (f16)1152.0f};
    k01              = (as_f16x2(lo) - bias) * scale2;
    k23              = (as_f16x2(hi) - bias) * scale2;
}

// 16 bytes of one column (k-locals 0..15 in natural order after dequant) -> 8 fp16 pairs.
__device__ __forceinline__ void dequant_16(const u32x4& w, f16x2 scale2, f16x2 (&out)[8])
{
    dequant_dword(w.x, scale2, out[0], out[1]);
    dequant_dword(w.y, scale2, out[2], out[3]);
    dequant_dword(w.z, scale2, out[4], out[5]);
    dequant_dword(w.w, scale2, out[6], out[7]);
}

// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second;
// v_permlane32_swap exchanges the upper half of the first with the lower half of the second.  Feeding the same
// value twice yields (a', b') with a' + b' = v[lane] + v[lane ^ 16] (resp. ^ 32) in every lane.
__device__ __forceinline__ float sum_xor16(float v)
{
    const u32 u = __builtin_bit_cast(u32, v);
    auto      r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (u32)r[0]) + __builtin_bit_cast(float, (u32)r[1]);
}
__device__ __forceinline__ float sum_xor32(float v)
{
    const u32 u = __builtin_bit_cast(u32, v);
    auto      r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (u32)r[0]) + __builtin_bit_cast(float, (u32)r[1]);
}

// ---- fused epilogue operands (either may be null).  y = fp16(acc) [+ bias[n]] [+ residual[m][n]], every step rounded to
// fp16: bit-identical to the reference's separate `output + bias` (qlinear.py:61) and to a separate residual add
// (FT's bias / residual epilogues, cutlass_kernels/fpA_intB_gemm.cu:35-97, are the reference's fused counterpart).
struct Epilogue {
    const f16* bias     = nullptr;  // [N]
    const f16* residual = nullptr;  // [M][N], row stride N; must not alias y unless it is y itself element for element
    int        act      = 0;        // EETQ_ACT_*: 0 = identity (the Python-level `output + bias` above); kActGlu8: see below
};

// The epilogue operands in SGPRs from HERE on.  Kernel arguments beyond the preloaded dwords (14 of the 16 asked for arrive with
// the wave on this toolchain) are fetched by an s_load that the compiler places where the value is first used -- for an epilogue
// that is between a kernel's last weight tile and its store, a scalar-cache miss on the critical tail of every launch
// (0.1 us of the 4.6 us GEMV, profiles/r06_gemv_ladder.txt).  Called right after a kernel has issued its first loads, the
// fetch overlaps their latency instead; the empty asm makes the values opaque, so they cannot be re-fetched later.
__device__ __forceinline__ Epilogue pin_epilogue(const Epilogue& ep)
{
    unsigned long long b = reinterpret_cast<unsigned long long>(ep.bias), r = reinterpret_cast<unsigned long long>(ep.residual);
    int                a = ep.act;
    asm volatile("" : "+s"(b), "+s"(r), "+s"(a));
    Epilogue e;
    e.bias     = reinterpret_cast<const f16*>(b);
    e.residual = reinterpret_cast<const f16*>(r);
    e.act      = a;
    return e;
}

// Gated-MLP epilogue of the M = 1 GEMV (internal value of Epilogue::act, reached through eetq_w8a16_gemv_glu8 only): the
// weight's columns come in groups of 16 = 8 gate columns followed by the 8 matching up columns, and the kernel writes
// y[8 t + c] = silu_mul(gate, up) for its tile t -- N/2 outputs, the silu_mul launch saved.
constexpr int kActGlu8 = 16;

// silu(g) * u in the roundings of the separate torch ops (silu in fp32 rounded to fp16, then an fp16 multiply); the one
// definition every kernel that fuses or performs the gated activation uses, so they all produce the same bits
__device__ __forceinline__ f16 silu_mul_f16(f16 g, f16 u)
{
    const float x = (float)g;
    return (f16)(x / (1.0f + expf(-x))) * u;
}

// Activation epilogues (EETQ_ACT_RELU / GELU / SILU): FT's bias + activation family, which the reference compiles but never
// reaches from Python (cutlass_kernels/fpA_intB_gemm.cu:35-97 -> fpA_intB_gemm_template.h:492-537 -> epilogue_helpers.h:20-71:
// LinearCombinationRelu / LinearCombinationSilu / LinearCombinationGeneric<GELU_taylor>, ScaleType::NoBetaScaling, compute
// type float): D = fp16( act( acc + float(bias) ) ) -- sum and activation in fp32, ONE rounding to fp16.  GELU is the tanh
// form 0.5 z (1 + tanh(0.7978845608 z (1 + 0.044715 z^2))) (cutlass_extensions/epilogue/thread/ft_fused_activations.h:73-84).
__device__ __forceinline__ float apply_act(float z, int act)
{
    if (act == EETQ_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == EETQ_ACT_SILU) return z / (1.0f + expf(-z));
    // EETQ_ACT_GELU
    return 0.5f * z * (1.0f + tanhf(0.7978845608028654f * z * (1.0f + 0.044715f * z * z)));
}

// one output element: identity keeps the reference's Python-level order (round to fp16, then an fp16 bias add)
__device__ __forceinline__ f16 finish_element(float acc, const Epilogue& ep, int n)
{
    if (ep.act == 0) {
        f16 v = (f16)acc;
        if (ep.bias) v = v + ep.bias[n];
        return v;
    }
    return (f16)apply_act(ep.bias ? acc + (float)ep.bias[n] : acc, ep.act);
}

// four consecutive columns n..n+3 (n % 4 == 0; bias 8-byte aligned) -> two packed pairs
__device__ __forceinline__ void finish_quad(const float (&a)[4], const Epilogue& ep, int n, f16x2& lo, f16x2& hi)
{
    if (ep.act == 0) {
        lo = f16x2{(f16)a[0], (f16)a[1]};
        hi = f16x2{(f16)a[2], (f16)a[3]};
        if (ep.bias) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(ep.bias + n);
            lo            = lo + as_f16x2(b.x);
            hi            = hi + as_f16x2(b.y);
        }
        return;
    }
    float z[4] = {a[0], a[1], a[2], a[3]};
    if (ep.bias) {
        const u32x2 b  = *reinterpret_cast<const u32x2*>(ep.bias + n);
        const f16x2 b0 = as_f16x2(b.x), b1 = as_f16x2(b.y);
        z[0] += (float)b0[0];
        z[1] += (float)b0[1];
        z[2] += (float)b1[0];
        z[3] += (float)b1[1];
    }
    lo = f16x2{(f16)apply_act(z[0], ep.act), (f16)apply_act(z[1], ep.act)};
    hi = f16x2{(f16)apply_act(z[2], ep.act), (f16)apply_act(z[3], ep.act)};
}

// ---- fused activation prologue of the M = 1 GEMV (gamma may be null = none): the activation vector is RMS-normalised while
// it is staged in LDS, x_eff[k] = fp16(clamp((x[k] * rsqrt(mean(x^2) + eps)) * gamma[k])) -- the arithmetic of
// eetq_rmsnorm_f16 (layernorm.cu:35-50), so a decoder block's norm -> projection pair is one launch.
struct Prologue {
    const f16* gamma = nullptr;  // [K]
    float      eps   = 0.f;
    // gated-MLP activation instead of a norm: x points at the `gate` half of a fused gate|up row, `up` at the other half;
    // x_eff[k] = fp16(silu(gate[k])) * up[k] (fp32 silu rounded to fp16, fp16 multiply: the roundings of eetq_silu_mul_f16)
    const f16* up = nullptr;     // [K]
};

// ---- launch helper: optionally attaches per-dispatch begin/end timestamps (eetq_prof_begin/_end) ---------
struct ProfEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
ProfEvents next_prof_events();  // {nullptr, nullptr} unless profiling is armed on this thread

template <typename Kern, typename... Args>
inline void launch_kernel(Kern kern, dim3 grid, dim3 block, size_t smem, hipStream_t stream, Args... args)
{
    const ProfEvents ev = next_prof_events();
    if (ev.start)
        hipExtLaunchKernelGGL(kern, grid, block, (unsigned)smem, stream, ev.start, ev.stop, 0, args...);
    else
        hipLaunchKernelGGL(kern, grid, block, (unsigned)smem, stream, args...);
}

// > 64 KiB of dynamic LDS needs a per-kernel opt-in (hipFuncSetAttribute, a host-side call of a few microseconds).  It is
// made ONCE per kernel and device -- `done` is the call site's bit mask of devices already opted in (one static atomic per
// kernel instantiation; lock-free, safe from several host threads) -- and asks for the CU's whole 160 KiB, so the launch
// path of every later call is a relaxed load.
constexpr int kMaxDynamicLds = 160 * 1024;
template <typename Kern>
inline int opt_in_large_lds(Kern kern, std::atomic<unsigned long long>& done)
{
    int dev = 0;
    EETQ_TRY_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return EETQ_OK;
    EETQ_TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kMaxDynamicLds));
    done.fetch_or(bit, std::memory_order_relaxed);
    return EETQ_OK;
}

// ---- A/B hooks ----------------------------------------------------------------------------------------
// Every EETQ_AMD_* variable that steers kernel SELECTION (stream plans, workgroup sizes, column units, quantiser forms ...) is
// an A/B hook for tools/ and tests/: it is read through tuning_env(), which answers only when the process also sets
// EETQ_AMD_TUNING=1.  Without that switch a production process never looks at them, so a stray variable cannot change which
// kernel runs (and with it the result bits: bit identity holds at equal wave count only).  Both the switch and the hooks are
// read once per process.  Operational variables are not hooks and stay unconditional: EETQ_AMD_SPLITK=0 (no library-owned
// scratch), EETQ_AMD_SPLITK_REGIONS (its size); EETQ_AMD_SPLITK_PLAN is honoured on the explicitly forced path only.
inline const char* tuning_env(const char* name)
{
    static const bool on = [] {
        const char* e = getenv("EETQ_AMD_TUNING");
        return e && e[0] == '1';
    }();
    return on ? getenv(name) : nullptr;
}

// ---- kernel launchers (one per .hip file) ------------------------------------------------------------
int launch_quantize(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed,
                    int layout, void* scales, float* workspace, size_t workspace_floats, hipStream_t stream);
size_t quantize_workspace_floats(size_t K, size_t N);  // floats launch_quantize's workspace must hold
int launch_pack(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, hipStream_t stream);
int launch_unpack(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, hipStream_t stream);
int launch_colmax(const void* w, int w_dtype, size_t K, size_t N, float* colmax, hipStream_t stream);
// int4 (W4A16): int4.hip
int launch_quantize_i4(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                       void* scales, float* colmax, hipStream_t stream);
int launch_pack_i4(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, hipStream_t stream);
int launch_unpack_i4(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, hipStream_t stream);
int launch_streamk_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                      hipStream_t stream);
int launch_gemv_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                   hipStream_t stream);
int launch_w4a16(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                 hipStream_t stream, int path = EETQ_PATH_AUTO);
int launch_gemv(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                hipStream_t stream, Prologue pro = Prologue{});
namespace gemv {
struct GroupedArgs;
}
bool gemv_grouped_supports(int K);
int  launch_gemv_grouped(const gemv::GroupedArgs& g, int K, int rows, hipStream_t stream);
int launch_gemm_mfma(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                     hipStream_t stream);
int launch_rmsnorm(const f16* x, const f16* gamma, f16* out, float eps, int rows, int cols, hipStream_t stream);
int launch_rotary(const int64_t* pos, f16* q, f16* k, const f16* cache, int tokens, int q_heads, int k_heads,
                  int head_size, int rot_dim, int q_stride, int k_stride, hipStream_t stream);

int launch_rotary_any(const int64_t* pos, void* q, void* k, const void* cache, int dtype, int tokens, int q_heads,
                      int k_heads, int head_size, int rot_dim, int q_stride, int k_stride, hipStream_t stream);

void set_attn_stamps(unsigned long long* buf);
int launch_rope_attn_decode(const int64_t* positions, const int64_t* slots, int slot_stride, const f16* q, const f16* k,
                            const f16* v, const f16* cos_sin, f16* kc, f16* vc, const f16* mask, f16* out, float* ws,
                            unsigned* tickets, int B, int H, int Hkv, int S, int D, int splits, float scaling,
                            const long* strides, const int64_t* kv_len, int kv_len_bias, int64_t* advance,
                            hipStream_t stream);
// causal attention over a prompt on MFMA (attn_prefill.hip); st: {q_b, q_token, q_head, k_b, k_head, k_row, v_b, v_head, v_row, out_b,
// out_token, out_head} in elements
bool prefill_attention_supports(int D);
int launch_prefill_attention(const f16* q, const f16* k, const f16* v, f16* out, int B, int H, int Hkv, int Tq, int Tk, int D,
                             int causal_offset, float scaling, const long* st, hipStream_t stream);
int launch_attn_decode(const f16* q, const f16* k, const f16* v, const f16* mask, f16* out, float* ws, int B, int H, int Hkv,
                       int S, int D, int splits, float scaling, const long* strides, const int64_t* kv_len, int kv_len_bias,
                       int64_t* advance, hipStream_t stream);

int launch_rotary_kvcache(const int64_t* pos, const int64_t* slots, int slot_stride, f16* q, const f16* k, const f16* v,
                          const f16* cache, f16* kcache, f16* vcache, int batch, int q_heads, int k_heads, int head_size, int rot_dim, long q_stride,
                          long k_stride, long v_stride, long c_sb, long c_sh, long c_ss, int max_pos, hipStream_t stream, int tokens = 0,
                          int first_row = 0);  // tokens > 0: the prefill form (batch * tokens blocks, rows first_row + t)

int launch_silu_mul(const f16* gu, f16* out, int rows, int inter, hipStream_t stream, bool glu8 = false);

// compute units of the current device (cached per device); 256 on MI355X
int device_cu_count();

int  launch_stream_read(const void* p, size_t bytes, unsigned* sink, hipStream_t stream);
int  launch_empty(unsigned* sink, int grid, int block, hipStream_t stream);
int  launch_clock_stamp(unsigned long long* out, int grid, hipStream_t stream);

int  stream_plan_query(int bits, int M, int N, int K, int ncu, int* form, int* nt, int* waves);
int  launch_streamk(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                    hipStream_t stream);

constexpr int kGemvMaxM   = 4;
constexpr int kStreamMaxM = 64;
constexpr int kMidMaxM    = 128;
constexpr int kSplitkMaxM = 1024;  // the split-K tile with row groups (round 5): up to 32 groups of <= 128 rows
int launch_gemm_mid(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                    hipStream_t stream);
// split-K form of the medium-batch tile (gemm_splitk.hip); force_nb / force_s (0 = planned) are tuning hooks; `env_plan`:
// honour EETQ_AMD_SPLITK_PLAN (only the explicitly forced path EETQ_PATH_SPLITK does: tests and tuning -- AUTO launches
// never read the environment)
int launch_gemm_splitk(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                       hipStream_t stream, int force_nb = 0, int force_s = 0, bool env_plan = false);
int launch_gemm_splitk_i4(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                          hipStream_t stream, bool env_plan = false);
// *r (optional): row groups the batch is cut into (gemm_splitk_kernel.hpp: R); nb / s / stages are planned for ceil(M / r) rows
void splitk_plan(int M, int N, int K, int* nb, int* s, int* stages, int* r = nullptr);
// true (and *r = row groups) when the row-group plan of the split-K tile applies: 97 <= M <= 1024, its workgroups fit the chip at
// once and the tiled kernel would not K-slice the shape (gemm_splitk.hip)
bool splitk_rows_plan(int M, int N, int K, int* r = nullptr);
// the calling stream's own split-K scratch region (gemm_splitk.hip): slabs (*slab_bytes of them), one ticket array per slice
// count (2 and 4), *max_tiles tickets each.  EETQ_ERR_UNSUPPORTED (no message) when the stream cannot have one right now.
int splitk_region(hipStream_t stream, float** slabs, size_t* slab_bytes, unsigned** tickets2, unsigned** tickets4, size_t* max_tiles);
// split-K form of the tiled MFMA GEMM (gemm.hip): K slices of the 128 x 64 tile when whole tiles leave CUs idle; falls back
// to launch_gemm_mfma when it does not apply.  *used_s (optional) reports the slice count that ran.
int launch_gemm_tile_splitk(const f16* x, const uint8_t* w, const f16* scales, Epilogue ep, f16* y, int M, int N, int K,
                            hipStream_t stream, int force_s = 0, int* used_s = nullptr, bool env_plan = false);
// slices launch_gemm_tile_splitk would use (1 = it would run the unsplit tiled kernel)
int tile_splitk_slices(int M, int N, int K);
// the same for the 128 x 128 tile (2 = two slices; 1 = unsplit)
int wide_tile_splitk_slices(int M, int N, int K);
// EETQ_PATH_* that W4A16 AUTO takes (int4.hip; MFMA = expansion to int8 tiles + the W8A16 kernels)
int w4a16_auto_path(int M, int N, int K);
// library-owned scratch (eetq_release_workspace): each frees its buffers on every device and adds the bytes to *freed
int launch_quantize_pack_i4_native(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_packed, void* scales,
                                   const float* colmax, hipStream_t stream);  // quant.hip; UNSUPPORTED -> int4.hip's own route
int release_splitk_workspace(size_t* freed);
int  release_splitk_region(hipStream_t stream);
int release_w4a16_workspace(size_t* freed);

}  // namespace eetq
