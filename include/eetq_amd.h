/*
 * eetq_amd.h -- C ABI of libeetq_amd.so: the MI355X (gfx950) implementation of EETQ's W8A16 hot path.
 *
 * Every entry point replaces one function of the reference's native boundary (the pybind module
 * `EETQ`, /root/reference/csrc/eetpy.cpp:7-19); the reference interface each one stands in for is cited
 * on the declaration.  Plain pointers and sizes only -- no torch types.  All device pointers refer to
 * the *current* HIP device; `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 * are asynchronous with respect to the host unless stated otherwise.
 *
 * Return value: 0 (EETQ_OK) on success, a negative EETQ_ERR_* code otherwise; the message for the most
 * recent failure on the calling thread is available from eetq_last_error().  The Python host layer
 * (eetq_amd/ops.py) turns a non-zero status into RuntimeError, which is what the reference's C++
 * exceptions become through pybind (csrc/utils/cuda_utils.h:40-60).
 */
#ifndef EETQ_AMD_H_
#define EETQ_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    EETQ_OK              = 0,
    EETQ_ERR_INVALID     = -1, /* bad argument (null pointer, unsupported shape, unknown enum) */
    EETQ_ERR_HIP         = -2, /* a HIP runtime call or kernel launch failed */
    EETQ_ERR_UNSUPPORTED = -3  /* valid request this build does not implement */
};

/* Element type of the weight handed to the quantiser and of the scales it returns
 * (reference: symmetric_quantize<half,half> / <float,float>, fpA_intB_gemm_wrapper.cu:78-95). */
enum { EETQ_DTYPE_F16 = 0, EETQ_DTYPE_F32 = 1, EETQ_DTYPE_F64 = 2 /* rotary only */ };

/* Byte layout of an int8 [K][N]-shaped weight tensor.
 *   ROW_MAJOR : raw two's-complement int8, element (k,n) at k*N+n -- the reference's "unprocessed" tensor.
 *   GFX950    : this library's native layout (DESIGN.md "HBM layout"): 1 KiB tiles of 16 columns x 64 k,
 *               uint8 = q+128, k-contiguous per column, dword bytes 1<->2 swapped.  What
 *               eetq_w8a16_gemm consumes.  Requires K % 64 == 0, N % 16 == 0.
 *   SM80      : the reference's processed layout for sm75..sm89 (cutlass_preprocessors.cc:497-534;
 *               ColumnMajorTileInterleave<64,2> + row permute + bias/byte swizzle), i.e. the bytes found
 *               in EETQ checkpoints written on NVIDIA GPUs.  Requires K % 64 == 0, N % 64 == 0. */
enum { EETQ_LAYOUT_ROW_MAJOR = 0, EETQ_LAYOUT_GFX950 = 1, EETQ_LAYOUT_SM80 = 2 };

/* Kernel selection for eetq_w8a16_gemm_ex (tests and tuning).  AUTO is what eetq_w8a16_gemm uses. */
enum { EETQ_PATH_AUTO = 0, EETQ_PATH_GEMV = 1 /* M <= 4 */, EETQ_PATH_MFMA = 2 /* LDS-tiled */, EETQ_PATH_STREAM = 3 /* M <= 64; AUTO uses it for 2..16 */,
       EETQ_PATH_MID = 4 /* M <= 128: 32-column tiles, 256-deep K steps; what AUTO runs for 17..128 under EETQ_AMD_SPLITK=0 */,
       EETQ_PATH_SPLITK = 5 /* W8A16: M <= 1024 -- split-K tiles: K slices with an in-launch deterministic reduction and / or row groups
                               of <= 128 rows (AUTO: 17 <= M <= 128, and the row-group plan on few-tile shapes up to M = 1024); W4A16: M <= 128 */,
       EETQ_PATH_TILESPLIT = 6 /* the LDS-tiled kernel with K slices per 128 x 64 tile (few tiles, deep K); unsplit when that does not apply */ };

/* Activation of the fused bias + activation epilogue (eetq_w8a16_gemm_act).  Reference: ActivationType
 * (csrc/utils/activation_types.h:23-38) as dispatched by CutlassFpAIntBGemmRunner::gemm_bias_act
 * (fpA_intB_gemm/fpA_intB_gemm_template.h:492-537): Relu, Gelu (tanh form), Silu, Identity. */
enum { EETQ_ACT_IDENTITY = 0, EETQ_ACT_RELU = 1, EETQ_ACT_GELU = 2, EETQ_ACT_SILU = 3 };

/* ---- quantise --------------------------------------------------------------------------------------
 * Replaces EETQ.quant_weights -> symmetric_quantize_last_axis_of_tensor
 * (csrc/cutlass_kernels/fpA_intB_gemm_wrapper.cu:28-107) -> ft::symmetric_quantize
 * (csrc/cutlass_kernels/cutlass_preprocessors.cc:581-678).
 * Per column n of the row-major [K][N] weight: s32 = max_k|w| * 2^-7 (fp32); scales[n] = (dtype)s32;
 * q = int8(clamp(round_half_away(w / s32), -128, 127)); bit-exact with the reference, including q = 127
 * for an all-zero column.  q_raw (ROW_MAJOR) and q_packed (in `layout`) may each be NULL.
 * All pointers are DEVICE pointers.  `scales` has dtype `w_dtype` and N elements.
 * Two launches: per-row-block column maxima (plain stores: no atomics, no zero fill), then quantise + pack (every pack
 * workgroup reduces the rows of maxima for its 64 columns).
 * eetq_quantize_i8_ws (ABI revision 2): `workspace` holds `workspace_floats` floats (device).
 *   >= eetq_quantize_workspace_floats(K, N) (= N * ceil(K / 128): one row of partial maxima per 128 weight rows): the
 *      two-launch route above;
 *   >= N: a zero fill + atomicMax maxima launch instead of the partial rows (same bytes out, a few us slower);
 *   <  N: EETQ_ERR_INVALID, nothing is launched.
 *   workspace = NULL: the library uses an internal buffer (one per device, allocated once and grown on demand, freed by
 *   eetq_release_workspace: calls that pass NULL must not overlap on one device -- concurrent streams bring their own
 *   workspace, as both Python bindings do); workspace_floats is ignored then.
 * eetq_quantize_i8 (ABI revision 1, kept): no size argument, and revision 1 documented the workspace as N floats -- a
 *   caller-provided workspace is therefore treated as exactly N floats (the atomicMax route); NULL as above.  Callers that
 *   allocate eetq_quantize_workspace_floats() floats should move to eetq_quantize_i8_ws to get the faster route. */
/* Revision history: 1 = round 1-2; 2 = eetq_quantize_i8_ws (sized workspace), eetq_release_stream_workspace, eetq_w4a16_gemm_ex;
 * 3 = eetq_diag_auto_path, EETQ_PATH_SPLITK accepts M <= 1024 (row groups); 4 = eetq_diag_splitk_plan; 5 =
 * eetq_rotary_neox_kvcache_prefill_f16, eetq_greedy_handover_f16, eetq_w8a16_gemm_glu8 at M > 16; 6 = eetq_prefill_attention_f16
 * (+ _supported).  Revisions only ADD entry points: a caller built against an older header keeps working. */
#define EETQ_AMD_ABI_VERSION 6
int eetq_abi_version(void);   /* EETQ_AMD_ABI_VERSION of the loaded library */
int eetq_quantize_i8_ws(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed,
                        int layout, void* scales, float* workspace, size_t workspace_floats, void* stream);
int eetq_quantize_i8(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed,
                     int layout, void* scales, float* workspace, void* stream);
size_t eetq_quantize_workspace_floats(size_t K, size_t N);

/* Same operation on HOST buffers (what the reference's CPU function receives): uploads w, runs the HIP
 * kernels, downloads the results and synchronises.  Blocking. */
int eetq_quantize_i8_host(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed,
                          int layout, void* scales);

/* ---- pack / unpack ---------------------------------------------------------------------------------
 * Replaces EETQ.preprocess_weights -> preprocess_weights_cuda (fpA_intB_gemm_wrapper.cu:109-128) ->
 * ft::preprocess_weights_for_mixed_gemm (cutlass_preprocessors.cc:497-534).
 * eetq_pack_i8: ROW_MAJOR int8 [K][N] -> `layout`.  eetq_unpack_i8: `layout` -> ROW_MAJOR (the reference has
 * no inverse; needed to load NVIDIA-written checkpoints).  Device pointers; src != dst. */
int eetq_pack_i8(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, void* stream);
int eetq_unpack_i8(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, void* stream);
/* Host-buffer variants (blocking), mirroring the CPU-tensor contract of preprocess_weights. */
int eetq_pack_i8_host(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout);
int eetq_unpack_i8_host(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout);

/* ---- fused dequant + GEMM --------------------------------------------------------------------------
 * Replaces EETQ.w8_a16_gemm / w8_a16_gemm_ -> w8_a16_gemm_forward_cuda(_)
 * (fpA_intB_gemm_wrapper.cu:130-202), i.e. weight_only_batched_gemv_launcher
 * (csrc/weightOnlyBatchedGemv/kernelLauncher.cu:122-232) for small M and ft::gemm_fp16_int
 * (csrc/cutlass_kernels/fpA_intB_gemm.cu:21-33) otherwise.
 *   y[m][n] = fp16( sum_k fp32(x[m][k]) * fp32( fp16( q[k][n] * scales[n] ) ) ),  fp32 accumulation.
 * x: fp16 [M][K] row-major; w_packed: GFX950 layout of the [K][N] int8 weight; scales: fp16 [N];
 * y: fp16 [M][N] row-major.  Device pointers, 16-byte aligned.  Requires K % 64 == 0, N % 16 == 0, M >= 1. */
int eetq_w8a16_gemm(const void* x, const int8_t* w_packed, const void* scales, void* y, int M, int N, int K,
                    void* stream);
/* As above with an explicit kernel path (EETQ_PATH_*); returns EETQ_ERR_UNSUPPORTED when the path cannot
 * run the shape (e.g. GEMV with M > 4, STREAM with M > 64). */
int eetq_w8a16_gemm_ex(const void* x, const int8_t* w_packed, const void* scales, void* y, int M, int N,
                       int K, int path, void* stream);

/* Fused bias epilogue: y = fp16(gemm) + bias, the fp16 add done after the fp16 rounding so the
 * result is bit-identical to the reference's separate `output + bias` (python/eetq/modules/qlinear.py:61), without the
 * extra elementwise kernel.  bias: fp16 [N], 8-byte aligned, or NULL.  `path` as in eetq_w8a16_gemm_ex. */
int eetq_w8a16_gemm_bias(const void* x, const int8_t* w_packed, const void* scales, const void* bias, void* y, int M,
                         int N, int K, int path, void* stream);

/* y = fp16(acc) [+ bias[n]] [+ residual[m][n]]: bias as above, then a residual tensor [M][N] (fp16, row stride N, 8-byte
 * aligned) added in fp16 after it -- bit-identical to a separate elementwise add of the GEMM output.  The reference's
 * counterpart is FT's bias / residual epilogue family (csrc/cutlass_kernels/fpA_intB_gemm.cu:35-97,
 * fpA_intB_gemm_template.h:492-537), which its Python layer never reaches.  `residual` may be `y` itself (in-place
 * accumulate into the residual stream); any other overlap is undefined.  Either pointer may be NULL. */
int eetq_w8a16_gemm_fused(const void* x, const int8_t* w_packed, const void* scales, const void* bias,
                          const void* residual, void* y, int M, int N, int K, int path, void* stream);

/* Bias + activation epilogue: replaces ft::gemm_fp16_int_bias_act (csrc/cutlass_kernels/fpA_intB_gemm.cu:35-62;
 * epilogues cutlass_extensions/.../epilogue_helpers.h:20-71), which the reference compiles but does not bind.
 *   act == EETQ_ACT_IDENTITY: exactly eetq_w8a16_gemm_fused (fp16 bias add after the fp16 rounding);
 *   otherwise               : y = fp16( act( acc + fp32(bias[n]) ) ) -- sum and activation in fp32, one rounding, as the
 *                             CUTLASS LinearCombinationRelu / Silu / Generic<GELU_taylor> epilogues compute it -- and the
 *                             optional residual is then added in fp16.  bias may be NULL. */
int eetq_w8a16_gemm_act(const void* x, const int8_t* w_packed, const void* scales, const void* bias, const void* residual,
                        void* y, int M, int N, int K, int path, int act, void* stream);

/* ---- int4 (W4A16) -----------------------------------------------------------------------------------
 * Replaces the quint4x2 branches of EETQ.quant_weights / preprocess_weights (fpA_intB_gemm_wrapper.cu:41-66,
 * cutlass_preprocessors.cc:360-418, 581-678 with PACKED_INT4_WEIGHT_ONLY) and the Int4b instantiations of the GEMV /
 * GEMM (weightOnlyBatchedGemv/kernel.h:68-116; fpA_intB_gemm.cu with uint4b_t).
 * Per column: s32 = max|w| * 2^-3; q = clamp(round_half_away(w / s32), -8, 7); ROW_MAJOR packing = two values per byte
 * along N (element 2j in the low nibble, 2j+1 in the high nibble), tensor [K][N/2] bytes.  N is the LOGICAL column count.
 * GFX950 layout: 1 KiB tiles of 16 columns x 128 k, unsigned nibbles q + 8.  Requires K % 128 == 0,
 * N % 16 == 0 (GFX950) / K % 64 == 0, N % 64 == 0 (SM80). */
int eetq_quantize_i4(const void* w, int w_dtype, size_t K, size_t N, int8_t* q_raw, int8_t* q_packed, int layout,
                     void* scales, float* workspace, void* stream);
int eetq_pack_i4(const int8_t* q_raw, size_t K, size_t N, int8_t* q_packed, int layout, void* stream);
int eetq_unpack_i4(const int8_t* q_packed, size_t K, size_t N, int8_t* q_raw, int layout, void* stream);
/* y = fp16( sum_k fp32(x) * fp32( fp16( q4[k][n] * scales[n] ) ) ) [+ bias] [+ residual]; w_packed in the GFX950 int4 layout. */
int eetq_w4a16_gemm(const void* x, const int8_t* w_packed, const void* scales, const void* bias, const void* residual,
                    void* y, int M, int N, int K, void* stream);
/* As above with an explicit kernel path: EETQ_PATH_AUTO (M = 1: dot2 GEMV on int4 tiles; 2..16: register-streaming MFMA
 * kernel; 17..128: split-K MFMA tile on int4 tiles; above: nibbles expanded to int8 tiles + the W8A16 kernels),
 * EETQ_PATH_GEMV (M <= 4), EETQ_PATH_STREAM (M <= 16), EETQ_PATH_SPLITK (M <= 128; honours EETQ_AMD_SPLITK_PLAN),
 * EETQ_PATH_MFMA (the expansion route at any M).  Other paths: EETQ_ERR_UNSUPPORTED. */
int eetq_w4a16_gemm_ex(const void* x, const int8_t* w_packed, const void* scales, const void* bias, const void* residual,
                       void* y, int M, int N, int K, int path, void* stream);

/* ---- side ops --------------------------------------------------------------------------------------
 * Replaces EETQ.layernorm_forward -> layernorm_forward_cuda (csrc/layernorm_kernels/layernorm.cu:98-113):
 * T5/RMS norm, out = clamp_fp16( x * rsqrt(mean(x^2) + eps) * gamma ), fp32 math, fp16 I/O.
 * Launches on `stream` (the reference uses the default stream, layernorm.cu:76). */
int eetq_rmsnorm_f16(const void* x, const void* gamma, void* out, float eps, int rows, int cols, void* stream);

/* Replaces EETQ.rotary_embedding_neox (csrc/embedding_kernels/pos_encoding_kernels.cu:55-87) for fp16:
 * in-place NeoX rotation of q and k ([tokens][heads][head_size]) by cos_sin_cache[positions[t]]
 * ([max_pos][rot_dim], cos half then sin half), every product/sum rounded to fp16 like the reference. */
int eetq_rotary_neox_f16(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                         int tokens, int heads, int head_size, int rot_dim, void* stream);

/* The same entry for every floating type the reference dispatches except bfloat16 (AT_DISPATCH_FLOATING_TYPES_AND2,
 * pos_encoding_kernels.cu:73-86; bf16 is outside this library's scope): dtype = EETQ_DTYPE_F16 / F32 / F64, shared by
 * query, key and cache.  F32 / F64 are plain IEEE products and sums of that type WITHOUT contraction (the reference's float
 * instantiation is whatever nvcc's default FMA contraction makes of `x * cos - y * sin`: agreement within one rounding). */
int eetq_rotary_neox(const int64_t* positions, void* query, void* key, const void* cos_sin_cache, int dtype,
                     int tokens, int heads, int head_size, int rot_dim, void* stream);

/* Same rotation on strided operands (no reference counterpart at the pybind boundary; it is what the reference's
 * EETLlamaAttention, python/eetq/modules/llama_modules.py:94-106, needs to rotate the q and k slices of a fused QKV
 * projection in place): q is [tokens][q_heads][head_size] with q_stride elements between tokens, k likewise with
 * k_heads (< q_heads for grouped-query models) and k_stride. */
int eetq_rotary_neox_strided_f16(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                                 int tokens, int q_heads, int k_heads, int head_size, int rot_dim, int q_stride,
                                 int k_stride, void* stream);

/* Grouped decode GEMV (extension; the reference's launcher takes one problem per call,
 * csrc/weightOnlyBatchedGemv/kernelLauncher.cu:122-232): `count` INDEPENDENT M = 1 problems -- no problem reads what
 * another one writes -- in as few dispatches as possible.  Problems of equal K (a multiple of 64 in [2048, 32768]) share
 * one dispatch per 32 problems: their tile rows form one grid, so the ~1.8 us of launch ramp / first-byte latency / tail
 * a 16 MiB GEMV pays per dispatch is paid once per group and the weight stream is as long as the group is.  Results are
 * bit-identical to `count` eetq_w8a16_gemm_fused calls (same kernel body, same summation order) wherever those take the
 * whole-tile-row GEMV kernel, and within one summation-order difference otherwise.  Problems the grouped kernel cannot
 * take (other K) are launched one by one.  `problems` is a HOST array, read during the call (the pointers inside are
 * device pointers, 16-byte aligned; bias / residual may be NULL); capturable in a HIP graph. */
typedef struct {
    const void*   x;        /* fp16 [K] */
    const int8_t* w_packed; /* GFX950 layout of the [K][N] int8 weight */
    const void*   scales;   /* fp16 [N] */
    void*         y;        /* fp16 [N] */
    const void*   bias;     /* fp16 [N] or NULL */
    const void*   residual; /* fp16 [N] or NULL */
    int           N, K;
} eetq_gemv_problem;
int eetq_w8a16_gemv_grouped(const eetq_gemv_problem* problems, int count, void* stream);

/* RMS-norm -> W8A16 GEMV as one launch, M = 1 (extension): y = fp16(sum_k fp32(xn[k]) * fp32(fp16(q*s))) [+ bias] [+ residual]
 * with xn = eetq_rmsnorm_f16(x, gamma, eps) computed while the activation vector is staged in LDS (same arithmetic; the
 * sum of squares is added in a different order, so xn may differ from the separate op by one fp16 ulp in rare elements). */
int eetq_w8a16_gemv_rmsnorm(const void* x, const void* gamma, float eps, const int8_t* w_packed, const void* scales,
                            const void* bias, const void* residual, void* y, int N, int K, void* stream);

/* Gated-MLP activation -> W8A16 GEMV as one launch, M = 1 (extension): gate_up is one row [gate(K) | up(K)] (16-byte
 * aligned, K % 8 == 0); y = W^T . (silu(gate) * up) [+ bias] [+ residual] with the activation computed while the vector is
 * staged in LDS, in the roundings of eetq_silu_mul_f16. */
int eetq_w8a16_gemv_silu_gated(const void* gate_up, const int8_t* w_packed, const void* scales, const void* bias,
                               const void* residual, void* y, int N, int K, void* stream);

/* Gated-MLP activation on a fused gate|up projection output (extension): out[r][i] = silu(gate_up[r][i]) *
 * gate_up[r][intermediate + i], gate_up [rows][2 * intermediate] dense, intermediate % 8 == 0.  fp32 silu rounded to fp16,
 * then an fp16 multiply. */
int eetq_silu_mul_f16(const void* gate_up, void* out, int rows, int intermediate, void* stream);

/* Gated MLP with the activation in the projection's epilogue (extension).  "glu8" column order of a fused gate|up weight:
 * columns in groups of 16 = gate columns 8t..8t+7 followed by up columns 8t..8t+7 (scales and bias in the same order), so
 * one 16-column tile holds both operands of 8 outputs.
 *   eetq_w8a16_gemv_glu8: M = 1; y[8t + c] = silu_mul(g, u) with g, u the projection's fp16 outputs (+ bias) for the pair,
 *     i.e. exactly eetq_w8a16_gemm (+ bias) followed by eetq_silu_mul_glu8_f16; y has N / 2 entries; gamma non-NULL adds
 *     the RMS-norm prologue of eetq_w8a16_gemv_rmsnorm.
 *   eetq_silu_mul_glu8_f16: out[r][8t + c] = silu_mul(gate_up[r][16t + c], gate_up[r][16t + 8 + c]) for the M > 1 GEMMs over
 *     such a weight; gate_up [rows][2 * intermediate] dense, intermediate % 8 == 0. */
int eetq_w8a16_gemv_glu8(const void* x, const void* gamma, float eps, const int8_t* w_packed, const void* scales,
                         const void* bias, void* y, int N, int K, void* stream);
/*   eetq_w8a16_gemm_glu8: the same for M > 1 rows (no norm prologue): y [M][N / 2].  2 <= M <= 16 (batched decode): the small-
 *     batch kernel's epilogue.  M > 16 (prompts; since ABI 5): where EETQ_PATH_AUTO runs the plain projection on the unsplit tiled
 *     MFMA kernel (eetq_diag_auto_path: EETQ_PATH_MFMA, or EETQ_PATH_TILESPLIT with one slice), that kernel writes silu_mul of its
 *     fp16 tile image -- the same bits as eetq_w8a16_gemm_act
 *     (+ bias) followed by eetq_silu_mul_glu8_f16, without the [M][N] round trip; any other shape returns EETQ_ERR_UNSUPPORTED
 *     and launches nothing (the caller runs those two). */
int eetq_w8a16_gemm_glu8(const void* x, const int8_t* w_packed, const void* scales, const void* bias, void* y, int M, int N,
                         int K, void* stream);
int eetq_silu_mul_glu8_f16(const void* gate_up, void* out, int rows, int intermediate, void* stream);

/* Decode-step rotary + KV-cache write (extension for the EET attention blocks): one new token per batch row b, rotated
 * by cos_sin_cache[positions[b]]; q [batch][q_heads][head_size] is rotated in place, k is rotated and written to
 * k_cache[b][head][slot][:], v is copied to v_cache[b][head][slot][:] (caches [batch][k_heads][max_positions][head_size]).
 * The cache row is slots[b * slot_stride] (DEVICE int64; slot_stride 0 = one slot for the whole batch, e.g. a static
 * cache's token counter; 1 = per row) or, with slots == NULL, positions[b].  The two differ for left-padded batches:
 * the rotary position counts real tokens, the slot counts cache rows.  Same fp16 arithmetic as eetq_rotary_neox_f16.
 * strides (elements): {q_b, k_b, v_b, cache_b, cache_head, cache_pos}.  Rows whose slot is outside [0, max_positions)
 * are left untouched. */
int eetq_rotary_neox_kvcache_f16(const int64_t* positions, const int64_t* slots, int slot_stride, void* query,
                                 const void* key, const void* value, const void* cos_sin_cache, void* k_cache,
                                 void* v_cache, int batch, int q_heads, int k_heads, int head_size, int rot_dim,
                                 const long* strides, int max_positions, void* stream);

/* Prefill form of the above (extension, ABI 5): `tokens` new tokens per batch row.  Token t of row b -- element b * tokens + t
 * of positions and of query / key / value, which advance by ONE stride per token (a fused QKV projection output [batch][tokens]
 * [(q_heads + 2 k_heads) * head_size]) -- is rotated by cos_sin_cache[positions[b * tokens + t]]; q in place, k into
 * k_cache[b][head][base + t][:], v copied to v_cache[b][head][base + t][:], base = *first_row_dev (DEVICE int64, e.g. a static
 * cache's token counter -- read, not advanced) or, with first_row_dev == NULL, first_row.  Replaces the rotary launch plus
 * the two index_copy launches (and their index arithmetic) of a static cache's update on a prompt.  Same fp16 arithmetic as
 * eetq_rotary_neox_f16.  strides (elements): {q_token, k_token, v_token, cache_b, cache_head, cache_pos}.  Host-known rows
 * must satisfy first_row + tokens <= max_positions (EETQ_ERR_INVALID, nothing launched); tokens whose device-side row falls
 * outside [0, max_positions) are left untouched and counted (eetq_decode_dropped_steps). */
int eetq_rotary_neox_kvcache_prefill_f16(const int64_t* positions, void* query, const void* key, const void* value,
                                         const void* cos_sin_cache, void* k_cache, void* v_cache, int batch, int tokens,
                                         const int64_t* first_row_dev, int first_row, int q_heads, int k_heads, int head_size,
                                         int rot_dim, const long* strides, int max_positions, void* stream);

/* Greedy decode hand-over (extension, ABI 5; the reference's recipe leaves this to transformers' generate loop,
 * examples/models/llama_transformers_example.py:68-79): for each of `batch` rows of fp16 logits [batch][vocab] (row_stride
 * elements apart) the index of the maximum -- the first one on ties, a NaN counts as the maximum: torch.argmax's answer -- is
 * written to out_tokens[b][*column] ([batch][out_cols] int64, out_stride elements apart; skipped when *column is outside
 * [0, out_cols)) and to next_token[b]; then *position += 1 and *column += 1 (DEVICE int64 scalars).  One launch, capturable:
 * what a HIP-graph decode loop does between two model steps. */
int eetq_greedy_handover_f16(const void* logits, long row_stride, int vocab, int batch, int64_t* out_tokens, long out_stride,
                             int out_cols, int64_t* column, int64_t* next_token, int64_t* position, void* stream);

/* Single-query (decode) attention over a KV cache; extension used by the EET attention blocks' decode step (the
 * reference delegates the attention product to flash-attn, python/eetq/modules/llama_modules.py:131-143).
 *   out[b][h][:] = softmax_j( scaling * q[b][h] . k[b][h / (heads/kv_heads)][j] + mask[b][j] ) @ v[...]   j < positions
 * fp16 operands, fp32 softmax/accumulation.  strides (in elements): {q_b, q_h, k_b, k_h, k_pos, v_b, v_h, v_pos, mask_b,
 * out_b, out_h}; head_dim (64 or 128) is the dense last dimension everywhere.  mask: additive fp16 [batch][positions]
 * rows (-inf = masked; mask_b may be 0 to share one row) or NULL.  workspace: batch * heads * splits * (head_dim + 4)
 * floats, 16-byte aligned.  A fully masked row gives 0.
 * kv_len (DEVICE int64 scalar or NULL): only cache rows j < min(positions, *kv_len + kv_len_bias) are attended -- the
 * valid length of a pre-allocated cache whose tail holds zeros or stale tokens.  advance (DEVICE int64 scalar or NULL): incremented by one
 * when the call's last kernel finishes (the cache's token counter; may alias kv_len). */
int eetq_decode_attention_f16(const void* q, const void* k_cache, const void* v_cache, const void* mask, void* out,
                              float* workspace, int batch, int heads, int kv_heads, int positions, int head_dim,
                              int splits, float scaling, const long* strides, const int64_t* kv_len, int kv_len_bias,
                              int64_t* advance, void* stream);

/* The chunk count both decode-attention entry points are tuned for (callers size the workspace with it): about two
 * workgroups per compute unit over batch * heads * splits, at most 8, at least 64 cache rows per chunk.  Measured at Llama-13B
 * shapes (tools/attn_bench.py): batch 1 -> 8, batch 2 -> 6, batch 4 -> 3. */
int eetq_decode_attention_splits(int batch, int heads, int positions);

/* Decode step of a pre-allocated KV cache as ONE launch (extension): eetq_rotary_neox_kvcache_f16 followed by
 * eetq_decode_attention_f16, bit-identical to that pair in the cache rows written and in the output.  The new token's q
 * [batch][heads][head_dim] and k [batch][kv_heads][head_dim] are rotated by cos_sin_cache[positions[b]] (rot_dim =
 * head_dim; q is NOT written back), the rotated k and v go to cache row slots[b * slot_stride] (slots NULL: positions[b]),
 * and the rows j < min(max_positions, *kv_len + kv_len_bias) are attended, the new row among them if it is in that range.
 * strides (elements): {q_b, k_b, v_b, kcache_b, kcache_head, kcache_pos, vcache_b, vcache_head, vcache_pos, mask_b,
 * out_b, out_head}.  workspace: batch * heads * splits * (head_dim + 4) floats, 16-byte aligned (contents irrelevant).  tickets: batch *
 * heads + 1 unsigned, ZERO before the first launch that uses them; the launch leaves them zero.  Launches that may run
 * concurrently need distinct workspaces and tickets.  advance: see eetq_decode_attention_f16 (may alias kv_len and slots:
 * it is written after every workgroup of the launch has read them). */
int eetq_rope_decode_attention_f16(const int64_t* positions, const int64_t* slots, int slot_stride, const void* query,
                                   const void* key, const void* value, const void* cos_sin_cache, void* k_cache,
                                   void* v_cache, const void* mask, void* out, float* workspace, unsigned* tickets,
                                   int batch, int heads, int kv_heads, int max_positions, int head_dim, int splits,
                                   float scaling, const long* strides, const int64_t* kv_len, int kv_len_bias,
                                   int64_t* advance, void* stream);

/* Causal attention over a PROMPT on the matrix cores (extension, ABI revision 6): out[b][t][h] = softmax_j<=t+causal_offset(scaling
 * q[b][t][h] . k[b][h / groups][j]) v[b][h / groups][j] over the first `keys` rows of a KV cache, fp16 in / out, fp32 scores, softmax
 * state and accumulation (probabilities rounded to fp16 for the second product, like flash-attn, which the reference's block calls
 * here: python/eetq/modules/llama_modules.py:131-143).  q: the rotated query rows, e.g. a view into the fused QKV projection's
 * output; k, v: cache tensors [batch][kv_heads][rows][head_dim].  strides (elements): {q_b, q_token, q_head, k_b, k_head, k_row,
 * v_b, v_head, v_row, out_b, out_token, out_head}, head_dim contiguous, q / k / v strides multiples of 8, out strides of 4.
 * causal_offset = keys - q_tokens for a prompt appended to `keys - q_tokens` cached rows (0 on an empty cache).  head_dim 64 or 128
 * (eetq_prefill_attention_supported); EETQ_ERR_UNSUPPORTED otherwise. */
int eetq_prefill_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads, int kv_heads, int q_tokens,
                               int keys, int head_dim, int causal_offset, float scaling, const long* strides, void* stream);
int eetq_prefill_attention_supported(int head_dim);

/* Diagnostics: while `stamps` (DEVICE, batch * heads * splits * 8 uint64) is set, every eetq_rope_decode_attention_f16
 * launch of this process records per workgroup the 100 MHz device clock at: 0 entry, 1 scalar reads done, 2 new token
 * rotated, 3 chunk done, 4 chunk record published, 5 ticket drawn, 6 (last workgroup of a head) merge done, 7 output
 * stored.  NULL switches it off.  Used by tools/attn_bench.py --stamps; not for production launches. */
int eetq_diag_attn_stamps(unsigned long long* stamps);

/* ---- profiling hook (no reference counterpart; used by bench.py) -------------------------------------
 * Between eetq_prof_begin(n) and eetq_prof_end() every kernel this library launches from the calling thread
 * carries a start/stop event pair on its dispatch packet; eetq_prof_end synchronises the device and returns
 * the kernel durations in microseconds, in launch order (*count = launches recorded, <= n).  Not capturable
 * into a HIP graph. */
int eetq_prof_begin(int max_launches);
int eetq_prof_end(float* durations_us, int capacity, int* count);
/* Diagnostic: a kernel that only reads `bytes` (multiple of 64 KiB) from p with the GEMV's load pattern; its
 * duration is the read floor for that many bytes on this chip.  `sink` = 4 writable device bytes. */
int eetq_diag_stream_read(const void* p, size_t bytes, void* sink, void* stream);
/* Diagnostic: a kernel of `grid` x `block` threads that touches no memory (the fixed cost of a dispatch, and the resolution
 * floor of the timing method it is measured with). */
int eetq_diag_empty(void* sink, int grid, int block, void* stream);
/* Diagnostic: `grid` one-wave workgroups each record {XCC_ID, HW_ID, s_memtime (shader cycles), s_memrealtime (100 MHz)} into
 * out[4 * workgroup .. +3] (DEVICE, grid * 4 uint64).  Two launches around a chain of kernels give the average shader clock
 * the chip held over the chain, per XCD: d(memtime) / d(memrealtime) x 100 MHz (bench.py `effective_clock_mhz`: the
 * evidence for "power-bound").  Graph-capturable. */
int eetq_diag_clock_stamp(unsigned long long* out, int grid, void* stream);
/* Diagnostic, host arithmetic only (no launch; with cus > 0 no device either): which plan the small-batch kernel (1 <= M <= 16 rows;
 * AUTO sends 2 <= M <= 16 there, except narrow deep W8A16 weights from M = 9: eetq_diag_auto_path) takes for a weight of `bits` (8 / 4), K x N, on a chip with `cus` compute units (<= 0: the current
 * device's).  *form = 0 activation fragments straight from L2 into registers, 1 rows copied once per workgroup into LDS, 2 per-wave
 * LDS-DMA ring; *tile_rows = 16-column tile rows per workgroup (1 / 2); *waves = waves per workgroup.  All plans give the same bits
 * at equal *waves.  The environment overrides (EETQ_AMD_I8_STREAM_PLAN ...) are not applied.  No reference counterpart: the
 * reference picks its CUTLASS tile by a timing sweep at run time (cutlass_heuristic.cc), this library by a rule -- this entry shows it. */
int eetq_diag_stream_plan(int bits, int M, int N, int K, int cus, int* form, int* tile_rows, int* waves);
/* Diagnostic, host arithmetic only (no launch): which kernel path EETQ_PATH_AUTO takes for an M x K activation against a K x N
 * weight of `bits` (8 / 4) on the current device.  *path = EETQ_PATH_* (for bits = 4: GEMV, STREAM, SPLITK, or MFMA = expansion to
 * int8 tiles + the W8A16 kernels); *detail (may be NULL) = K slices per tile when *path is EETQ_PATH_TILESPLIT (1 = the unsplit
 * tiled kernel), the row groups of the plan when *path is EETQ_PATH_SPLITK (0 = K slices only), else 0.  It calls the function the launchers call.  Replaces, as far as anything does, the reference's run-time
 * choice: the m <= 4 switch (fpA_intB_gemm_wrapper.cu:149-162) and the occupancy-scored tile pick
 * (cutlass_kernels/cutlass_heuristic.cc:123-206) -- one rule here, printed by bench.py's `config4` block per point and measured
 * against every forced path by tools/auto_regret.py. */
int eetq_diag_auto_path(int bits, int M, int N, int K, int* path, int* detail);

/* Diagnostic, host arithmetic only: the plan the split-K medium-batch tile (EETQ_PATH_SPLITK, W8A16 and W4A16 alike) runs an M x K
 * activation against a K x N weight with -- *column_blocks (32-column blocks per workgroup: 1 or 2), *k_slices (1, 2 or 4; > 1
 * needs the stream's scratch region), *ring (10 * activation stages + weight stages: 22 or 33), *row_groups (the batch cut along
 * M; M <= 32 * 4 * row_groups).  It calls the planner the launcher calls (gemm_splitk.hip::splitk_plan: one cost model over every
 * combination, its constants fitted on the measured time of every plan -- profiles/r05_splitk_plan_regret*.jsonl), so
 * tests/test_abi.py can hold the planner against those tables without a GPU.  1 <= M <= 1024, K % 64 == 0. */
int eetq_diag_splitk_plan(int M, int N, int K, int* column_blocks, int* k_slices, int* ring, int* row_groups);

/* Decode steps on a pre-allocated KV cache (eetq_rope_decode_attention_f16, eetq_rotary_neox_kvcache_f16) whose new token
 * was NOT written because its cache row lies outside the cache (slot >= rows: the cache is full; or a negative position).
 * The kernels skip the write instead of faulting (stock transformers raises an index error there); this entry synchronises
 * the current device and reports how many such steps its kernels have dropped since the last reset (batch rows count
 * individually), optionally resetting the count.  A non-zero count means the tokens generated after that point are wrong. */
int eetq_decode_dropped_steps(unsigned long long* count, int reset);

/* ---- library-owned scratch ---------------------------------------------------------------------------
 * The reference's operators take no workspace argument (fpA_intB_gemm_wrapper.cu:169-170 passes none), so the few
 * buffers the kernels need are owned by the library: the split-K partial-tile regions (40 MiB per launch stream that
 * ever ran a 17 <= M <= 128 GEMM, at most EETQ_AMD_SPLITK_REGIONS (default 16) per device; a stream that cannot get a
 * region of its own runs unsplit -- regions are never shared), the W4A16 prefill expansion buffers and the quantiser's
 * NULL-workspace buffer.  eetq_release_workspace synchronises the devices that hold any, frees all of it and reports the
 * bytes freed (bytes_freed may be NULL); later calls re-create what they need.  Do not call it while a HIP graph that
 * captured a split-K or W4A16 launch is still going to be replayed. */
int eetq_release_workspace(size_t* bytes_freed);
/* Gives the split-K region owned by `stream` (current device) back to the pool after synchronising the stream; call it
 * before destroying a stream that ran 17 <= M <= 128 GEMMs, unless HIP graphs captured on it are still going to be
 * replayed.  The library never reclaims a region on its own: once all regions are owned, further streams run the unsplit
 * kernels.  EETQ_OK also when the stream owns nothing. */
int eetq_release_stream_workspace(void* stream);

/* ---- misc ------------------------------------------------------------------------------------------ */
const char* eetq_last_error(void);   /* thread-local, never NULL */
const char* eetq_version(void);      /* "eetq_amd <version> gfx950" */
/* 1 if the current HIP device is a gfx950; 0 otherwise (kernels are built for gfx950 only). */
int eetq_device_supported(void);

#ifdef __cplusplus
}
#endif
#endif /* EETQ_AMD_H_ */
