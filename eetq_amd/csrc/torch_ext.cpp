// The compiled operator module `EETQ`: torch::Tensor in / torch::Tensor out over the C ABI of libeetq_amd.so.
//
// Drop-in for the reference's pybind module (csrc/eetpy.cpp:7-19): the six functions it binds keep their names, argument
// order, py::arg names and defaults, current-stream / device-guard behaviour and error type (C++ exception ->
// RuntimeError).  Host code only (g++): every kernel lives behind include/eetq_amd.h.  Optional trailing keyword arguments
// (layout=, path=, bias=, residual=, norm=, gated=) and five further functions are extensions of this library
// (include/eetq_amd.h says which reference interface, if any, each one stands in for).
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/eetq_amd.h"

namespace {

using torch::Tensor;
using OptTensor = std::optional<Tensor>;

void check(int status)
{
    if (status != EETQ_OK) {
        const char* msg = eetq_last_error();
        throw std::runtime_error(msg && *msg ? msg : "eetq_amd: error " + std::to_string(status));
    }
}

void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

int layout_id(const std::string& name)
{
    if (name == "gfx950" || name == "native") return EETQ_LAYOUT_GFX950;
    if (name == "sm80") return EETQ_LAYOUT_SM80;
    if (name == "row_major") return EETQ_LAYOUT_ROW_MAJOR;
    throw std::runtime_error("unknown weight layout '" + name + "' (expected 'gfx950', 'sm80' or 'row_major')");
}

int path_id(const std::string& name)
{
    if (name == "auto") return EETQ_PATH_AUTO;
    if (name == "gemv") return EETQ_PATH_GEMV;
    if (name == "mfma") return EETQ_PATH_MFMA;
    if (name == "stream") return EETQ_PATH_STREAM;
    if (name == "mid") return EETQ_PATH_MID;
    if (name == "splitk") return EETQ_PATH_SPLITK;
    if (name == "tilesplit") return EETQ_PATH_TILESPLIT;
    throw std::runtime_error("unknown GEMM path '" + name + "'");
}

c10::Device work_device(const Tensor& t)
{
    if (t.is_cuda()) return t.device();
    TORCH_CHECK(torch::cuda::is_available(), "eetq_amd: no HIP device available; the W8A16 path has no CPU implementation");
    return c10::Device(c10::kCUDA, c10::hip::current_device());
}

// ---- quantise (reference: symmetric_quantize_last_axis_of_tensor, fpA_intB_gemm_wrapper.cu:28-107) ----------------
std::vector<Tensor> quant_weights(const Tensor& weight, py::object quant_type, bool return_unprocessed_quantized_tensor,
                                  const std::string& layout)
{
    const at::ScalarType qt = torch::python::detail::py_object_to_dtype(quant_type);
    TORCH_CHECK(weight.is_contiguous(), "weight must be contiguous");
    TORCH_CHECK(weight.numel() != 0, "weight should not be empty tensor");
    TORCH_CHECK(weight.dim() == 2 || weight.dim() == 3, "Invalid dim. The dim of weight should be 2 or 3");
    const auto st = weight.scalar_type();
    TORCH_CHECK(st == at::kHalf || st == at::kFloat, "Invalid datatype. Weight must be FP16 or FP32");
    TORCH_CHECK(qt == at::kChar || qt == at::kQUInt4x2, "Must be int4 or int8 quantization");
    if (weight.dim() == 3) {
        // [E, K, N] expert stack.  The reference accepts it, allocates [E, K, N] / [E, N] outputs (fpA_intB_gemm_wrapper.cu:45-66)
        // and then hands symmetric_quantize the 2-D shape {num_rows, num_cols} (:82, :90): only expert 0 is quantised, the other
        // experts' outputs stay uninitialised.  Here every expert is quantised -- the same output shapes, expert 0 identical.
        std::vector<Tensor> parts[3];
        for (int64_t e = 0; e < weight.size(0); ++e) {
            auto r = quant_weights(weight.select(0, e), quant_type, return_unprocessed_quantized_tensor, layout);
            for (size_t i = 0; i < r.size(); ++i) parts[i].push_back(r[i]);
        }
        std::vector<Tensor> out;
        for (auto& p : parts)
            if (!p.empty()) out.push_back(torch::stack(p, 0));
        return out;
    }
    const bool   int4 = qt == at::kQUInt4x2;
    const int    lay  = layout_id(layout);
    const size_t K = weight.size(0), N = weight.size(1);
    const auto   dev = work_device(weight);
    c10::DeviceGuard guard(dev);
    Tensor       w_dev = weight.is_cuda() ? weight : weight.to(dev);
    const auto   i8    = torch::TensorOptions().dtype(at::kChar).device(dev);
    Tensor       raw, processed, scales = torch::empty({(int64_t)N}, weight.options().device(dev));
    // workspace of the quantiser (int8: one row of column maxima per 128 weight rows; int4: N floats)
    Tensor colmax = torch::empty({(int64_t)(int4 ? N : eetq_quantize_workspace_floats(K, N))},
                                 torch::TensorOptions().dtype(at::kFloat).device(dev));
    if (!int4) {
        if (return_unprocessed_quantized_tensor) raw = torch::empty({(int64_t)K, (int64_t)N}, i8);
        processed = torch::empty({(int64_t)K, (int64_t)N}, i8);
        check(eetq_quantize_i8_ws(w_dev.data_ptr(), st == at::kHalf ? EETQ_DTYPE_F16 : EETQ_DTYPE_F32, K, N,
                                  raw.defined() ? raw.data_ptr<int8_t>() : nullptr, processed.data_ptr<int8_t>(), lay,
                                  scales.data_ptr(), colmax.data_ptr<float>(), (size_t)colmax.numel(), stream_of(w_dev)));
    } else {
        // packed int4: two values per byte along N (reference output shape [K, N/2], fpA_intB_gemm_wrapper.cu:60-66)
        TORCH_CHECK(N % 2 == 0, "int4 quantization needs an even number of columns");
        if (return_unprocessed_quantized_tensor) raw = torch::empty({(int64_t)K, (int64_t)N / 2}, i8);
        processed = torch::empty({(int64_t)K, (int64_t)N / 2}, i8);
        check(eetq_quantize_i4(w_dev.data_ptr(), st == at::kHalf ? EETQ_DTYPE_F16 : EETQ_DTYPE_F32, K, N,
                               raw.defined() ? raw.data_ptr<int8_t>() : nullptr, processed.data_ptr<int8_t>(), lay,
                               scales.data_ptr(), colmax.data_ptr<float>(), stream_of(w_dev)));
    }
    if (!weight.is_cuda()) {  // CPU tensors in -> CPU tensors out, like the reference (:33)
        processed = processed.cpu();
        scales    = scales.cpu();
        if (raw.defined()) raw = raw.cpu();
    }
    if (return_unprocessed_quantized_tensor) return {raw, processed, scales};
    return {processed, scales};
}

Tensor relayout(const Tensor& src_in, const std::string& layout, bool pack, bool is_int4)
{
    TORCH_CHECK(src_in.scalar_type() == at::kChar, "expected an int8 tensor");
    if (src_in.dim() != 2) throw std::runtime_error("[FT][ERROR] Shape must be 2-D");
    Tensor       src = src_in.contiguous();
    const int    lay = layout_id(layout);
    const size_t K = src.size(0), Nb = src.size(1);
    const auto   dev = work_device(src);
    c10::DeviceGuard guard(dev);
    Tensor       s_dev = src.is_cuda() ? src : src.to(dev);
    Tensor       out   = torch::empty_like(s_dev);
    if (is_int4)
        check((pack ? eetq_pack_i4 : eetq_unpack_i4)(s_dev.data_ptr<int8_t>(), K, Nb * 2, out.data_ptr<int8_t>(), lay,
                                                     stream_of(s_dev)));
    else
        check((pack ? eetq_pack_i8 : eetq_unpack_i8)(s_dev.data_ptr<int8_t>(), K, Nb, out.data_ptr<int8_t>(), lay,
                                                     stream_of(s_dev)));
    return src.is_cuda() ? out : out.cpu();
}

// reference: preprocess_weights_cuda, fpA_intB_gemm_wrapper.cu:109-128
Tensor preprocess_weights(const Tensor& origin_weight, bool is_int4, const std::string& layout)
{
    return relayout(origin_weight, layout, true, is_int4);
}

Tensor unprocess_weights(const Tensor& processed_weight, const std::string& layout, bool is_int4)
{
    return relayout(processed_weight, layout, false, is_int4);
}

// ---- fused dequant + GEMM ---------------------------------------------------------------------------------------------
void layernorm_forward(const Tensor& input, const Tensor& gamma, Tensor& out, double eps);
// eetq_rotary_neox_kvcache_f16 + eetq_decode_attention_f16 as one launch (the decode step of a static cache)
Tensor rope_decode_attention(const Tensor& positions, const Tensor& query, const Tensor& key, const Tensor& value,
                             const Tensor& cos_sin_cache, Tensor& key_cache, Tensor& value_cache, Tensor& tickets,
                             const OptTensor& slots, const OptTensor& mask, std::optional<double> scaling,
                             std::optional<int64_t> splits_in, const OptTensor& kv_len, int64_t kv_len_bias,
                             const OptTensor& advance)
{
    for (const Tensor* t : std::initializer_list<const Tensor*>{&query, &key, &value, &cos_sin_cache, &key_cache, &value_cache})
        TORCH_CHECK(t->scalar_type() == at::kHalf, "rope_decode_attention: float16 tensors expected");
    TORCH_CHECK(positions.scalar_type() == at::kLong && positions.is_contiguous(),
                "rope_decode_attention: positions must be contiguous int64");
    TORCH_CHECK(query.dim() == 3 && key.dim() == 3 && value.dim() == 3 && key_cache.dim() == 4 &&
                    value_cache.sizes() == key_cache.sizes(),
                "rope_decode_attention: expected query [B, H, D], key / value [B, Hkv, D], caches [B, Hkv, S, D]");
    const int64_t B = query.size(0), H = query.size(1), D = query.size(2), Hkv = key.size(1), S = key_cache.size(2);
    TORCH_CHECK(key.size(0) == B && key.size(2) == D && value.sizes() == key.sizes() && key_cache.size(0) == B &&
                    key_cache.size(1) == Hkv && key_cache.size(3) == D && H % Hkv == 0 && positions.numel() == B,
                "rope_decode_attention: shape mismatch");
    for (const Tensor* t : std::initializer_list<const Tensor*>{&query, &key, &value})
        TORCH_CHECK(t->stride(-1) == 1 && t->stride(-2) == D, "rope_decode_attention: [heads, head_size] must be dense");
    TORCH_CHECK(key_cache.stride(-1) == 1 && value_cache.stride(-1) == 1 && cos_sin_cache.is_contiguous() &&
                    cos_sin_cache.size(-1) == D,
                "rope_decode_attention: cache rows must be dense and the rotation must cover the whole head (rot_dim == D)");
    TORCH_CHECK(tickets.scalar_type() == at::kInt && tickets.is_contiguous() && tickets.numel() >= B * H + 1 &&
                    tickets.device() == query.device(),
                "rope_decode_attention: tickets must be a zeroed int32 tensor of at least B * H + 1 elements on the device");
    int slot_stride = 0;
    if (slots) {
        const Tensor& s = *slots;
        TORCH_CHECK(s.scalar_type() == at::kLong && s.device() == query.device() && (s.numel() == 1 || s.numel() == B) &&
                        s.is_contiguous(),
                    "rope_decode_attention: slots must be contiguous int64 on the device, 1 or B elements");
        slot_stride = (s.numel() == B && B > 1) ? 1 : 0;
    }
    Tensor  mrow;
    int64_t m_sb = 0;
    if (mask) {
        mrow = mask->dim() != 2 ? mask->reshape({mask->size(0), -1}) : *mask;
        TORCH_CHECK(mrow.scalar_type() == at::kHalf && mrow.size(-1) >= S && mrow.stride(-1) == 1 && mrow.device() == query.device(),
                    "rope_decode_attention: mask must be additive float16 with a dense last dimension >= S");
        TORCH_CHECK(mrow.size(0) == 1 || mrow.size(0) == B,
                    "rope_decode_attention: the mask needs one row per batch entry (or a single shared row)");
        m_sb = (mrow.size(0) == B && B > 1) ? mrow.stride(0) : 0;
    }
    for (const OptTensor* t : std::initializer_list<const OptTensor*>{&kv_len, &advance})
        if (*t)
            TORCH_CHECK((*t)->scalar_type() == at::kLong && (*t)->numel() == 1 && (*t)->device() == query.device(),
                        "rope_decode_attention: kv_len / advance must be a one-element int64 tensor on the query's device");
    const double sc = scaling ? *scaling : 1.0 / std::sqrt((double)D);
    int64_t      splits = splits_in ? *splits_in
                                    : (int64_t)eetq_decode_attention_splits((int)B, (int)H, (int)S);
    Tensor       out = torch::empty({B, H, D}, query.options());
    Tensor       ws  = torch::empty({B * H * splits * (D + 4)}, query.options().dtype(at::kFloat));
    const long   strides[12] = {(long)query.stride(0),       (long)key.stride(0),         (long)value.stride(0),
                                (long)key_cache.stride(0),   (long)key_cache.stride(1),   (long)key_cache.stride(2),
                                (long)value_cache.stride(0), (long)value_cache.stride(1), (long)value_cache.stride(2),
                                (long)m_sb,                  (long)out.stride(0),         (long)out.stride(1)};
    c10::DeviceGuard guard(query.device());
    check(eetq_rope_decode_attention_f16(
        positions.data_ptr<int64_t>(), slots ? slots->data_ptr<int64_t>() : nullptr, slot_stride, query.data_ptr(),
        key.data_ptr(), value.data_ptr(), cos_sin_cache.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        mrow.defined() ? mrow.data_ptr() : nullptr, out.data_ptr(), ws.data_ptr<float>(),
        reinterpret_cast<unsigned*>(tickets.data_ptr<int32_t>()), (int)B, (int)H, (int)Hkv, (int)S, (int)D, (int)splits,
        (float)sc, strides, kv_len ? kv_len->data_ptr<int64_t>() : nullptr, (int)kv_len_bias,
        advance ? advance->data_ptr<int64_t>() : nullptr, stream_of(query)));
    return out;
}

// causal attention over a prompt on the matrix cores (eetq_prefill_attention_f16): query [B, T, H, D] (any token / head strides, D
// dense), key / value cache views [B, Hkv, rows >= keys, D]; returns [B, T, H, D] contiguous
bool prefill_attention_supported(int64_t head_dim) { return eetq_prefill_attention_supported((int)head_dim) != 0; }

Tensor prefill_attention(const Tensor& query, const Tensor& key, const Tensor& value, int64_t keys, std::optional<double> scaling,
                         std::optional<int64_t> causal_offset)
{
    for (const Tensor* t : std::initializer_list<const Tensor*>{&query, &key, &value})
        TORCH_CHECK(t->scalar_type() == at::kHalf && t->is_cuda() && t->dim() == 4 && t->stride(-1) == 1 && t->device() == query.device(),
                    "prefill_attention: float16 CUDA tensors [B, T, H, D] / [B, Hkv, S, D] with a dense last dimension expected");
    const int64_t B = query.size(0), T = query.size(1), H = query.size(2), D = query.size(3), Hkv = key.size(1);
    TORCH_CHECK(key.size(0) == B && key.size(3) == D && value.sizes() == key.sizes() && Hkv > 0 && H % Hkv == 0 && keys > 0 &&
                    keys <= key.size(2) && T > 0,
                "prefill_attention: shape mismatch");
    TORCH_CHECK(prefill_attention_supported(D), "prefill_attention: unsupported head_dim");
    const double sc  = scaling ? *scaling : 1.0 / std::sqrt((double)D);
    const int64_t off = causal_offset ? *causal_offset : keys - T;
    Tensor       out = torch::empty({B, T, H, D}, query.options());
    const long   st[12] = {(long)query.stride(0), (long)query.stride(1), (long)query.stride(2), (long)key.stride(0), (long)key.stride(1),
                           (long)key.stride(2),   (long)value.stride(0), (long)value.stride(1), (long)value.stride(2),
                           (long)out.stride(0),   (long)out.stride(1),   (long)out.stride(2)};
    c10::DeviceGuard guard(query.device());
    check(eetq_prefill_attention_f16(query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(), (int)B, (int)H, (int)Hkv, (int)T,
                                     (int)keys, (int)D, (int)off, (float)sc, st, stream_of(query)));
    return out;
}

Tensor silu_mul(const Tensor& gate_up, bool glu8 = false);

void check_epilogue(const Tensor& input, const OptTensor& bias, const OptTensor& residual, int64_t m, int64_t n)
{
    if (bias) {
        const Tensor& b = *bias;
        TORCH_CHECK(b.scalar_type() == at::kHalf && b.device() == input.device() && b.numel() == n && b.is_contiguous(),
                    "w8_a16_gemm: bias must be a contiguous float16 [N] tensor on the input's device");
    }
    if (residual) {
        const Tensor& r = *residual;
        TORCH_CHECK(r.scalar_type() == at::kHalf && r.device() == input.device() && r.numel() == m * n &&
                        r.is_contiguous() && r.size(-1) == n,
                    "w8_a16_gemm: residual must be a contiguous float16 [..., N] tensor with the output's element count, "
                    "on the input's device");
    }
}

Tensor& gemm_launch(const Tensor& input, const Tensor& weight, const Tensor& scale, Tensor& output, int64_t m, int64_t n,
                    int64_t k, int path, const OptTensor& bias, const OptTensor& residual, int act)
{
    TORCH_CHECK(input.scalar_type() == at::kHalf, "w8_a16_gemm: input must be float16 (got ", input.scalar_type(), ")");
    TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
    TORCH_CHECK(weight.scalar_type() == at::kChar && scale.scalar_type() == at::kHalf,
                "w8_a16_gemm: weight must be int8 and scale float16");
    TORCH_CHECK(weight.device() == input.device() && scale.device() == input.device() && output.device() == input.device(),
                "w8_a16_gemm: input, weight, scale and output must be on the same device");
    TORCH_CHECK(weight.is_contiguous() && scale.is_contiguous() && output.is_contiguous(),
                "w8_a16_gemm: weight, scale and output must be contiguous");
    check_epilogue(input, bias, residual, m, n);
    Tensor           x = input.contiguous();
    c10::DeviceGuard guard(input.device());
    const bool       int4 = weight.size(-1) * 2 == n && weight.size(-1) != n;  // packed int4: [K, N/2] bytes
    TORCH_CHECK(!int4 || act == EETQ_ACT_IDENTITY, "w8_a16_gemm: int4 weights take no activation epilogue");
    if (int4)
        check(eetq_w4a16_gemm_ex(x.data_ptr(), weight.data_ptr<int8_t>(), scale.data_ptr(), bias ? bias->data_ptr() : nullptr,
                                 residual ? residual->data_ptr() : nullptr, output.data_ptr(), (int)m, (int)n, (int)k, path,
                                 stream_of(input)));
    else
        check(eetq_w8a16_gemm_act(x.data_ptr(), weight.data_ptr<int8_t>(), scale.data_ptr(),
                                  bias ? bias->data_ptr() : nullptr, residual ? residual->data_ptr() : nullptr,
                                  output.data_ptr(), (int)m, (int)n, (int)k, path, act, stream_of(input)));
    return output;
}

int act_id(const std::string& name)
{
    if (name.empty() || name == "identity" || name == "none") return EETQ_ACT_IDENTITY;
    if (name == "relu") return EETQ_ACT_RELU;
    if (name == "gelu") return EETQ_ACT_GELU;
    if (name == "silu") return EETQ_ACT_SILU;
    throw std::runtime_error("unknown activation '" + name + "' (identity, relu, gelu, silu; silu_glu8 for gated weights)");
}

std::vector<int64_t> out_shape(const Tensor& input, int64_t n)
{
    std::vector<int64_t> s(input.sizes().begin(), input.sizes().end());
    s.back() = n;
    return s;
}

// reference: w8_a16_gemm_forward_cuda, fpA_intB_gemm_wrapper.cu:130-173 (fresh output, current stream, asynchronous)
Tensor w8_a16_gemm(const Tensor& input_in, const Tensor& weight, const Tensor& scale, const std::string& path,
                   const OptTensor& bias, const OptTensor& residual, const std::optional<std::tuple<Tensor, double>>& norm,
                   bool gated, const std::string& activation)
{
    Tensor  input = input_in;
    int64_t n     = scale.numel();  // [N]; for packed int4 weights the byte tensor is [K, N/2]
    TORCH_CHECK(input.dim() >= 1 && weight.dim() == 2, "w8_a16_gemm: expected input [..., K] and weight [K, N]");
    const int64_t kw = weight.size(0);
    if (activation == "silu_glu8") {
        // gated MLP over a weight in "glu8" column order: [..., N/2] = silu_mul of the column pairs.  One launch for a single
        // row (activation in the GEMV epilogue, RMS-norm in its prologue); projection + eetq_silu_mul_glu8_f16 otherwise.
        TORCH_CHECK(!gated && !residual && n % 16 == 0 && weight.size(1) == n,
                    "w8_a16_gemm: silu_glu8 takes an int8 [K, N] weight with N % 16 == 0, no residual");
        const int64_t rows = kw ? input.numel() / kw : 0;
        const bool    fusable_norm = !norm || (std::get<0>(*norm).scalar_type() == at::kHalf &&
                                            std::get<0>(*norm).is_contiguous() && std::get<0>(*norm).numel() == kw &&
                                            std::get<0>(*norm).device() == input.device());
        if (rows == 1 && path == "auto" && input.size(-1) == kw && input.is_cuda() && input.scalar_type() == at::kHalf &&
            fusable_norm) {
            TORCH_CHECK(weight.scalar_type() == at::kChar && scale.scalar_type() == at::kHalf && weight.is_contiguous() &&
                            weight.device() == input.device() && scale.device() == input.device(),
                        "w8_a16_gemm: weight must be contiguous int8 and scale float16, on the input's device");
            check_epilogue(input, bias, residual, 1, n);
            Tensor           x = input.contiguous();
            Tensor           output = torch::empty(out_shape(input, n / 2), input.options());
            c10::DeviceGuard guard(input.device());
            check(eetq_w8a16_gemv_glu8(x.data_ptr(), norm ? std::get<0>(*norm).data_ptr() : nullptr,
                                       norm ? (float)std::get<1>(*norm) : 0.f, weight.data_ptr<int8_t>(), scale.data_ptr(),
                                       bias ? bias->data_ptr() : nullptr, output.data_ptr(), (int)n, (int)kw, stream_of(input)));
            return output;
        }
        if (rows >= 2 && rows <= 16 && path == "auto" && input.size(-1) == kw && input.is_cuda() &&
            input.scalar_type() == at::kHalf && weight.scalar_type() == at::kChar && scale.scalar_type() == at::kHalf &&
            weight.is_contiguous() && weight.device() == input.device() && scale.device() == input.device()) {
            // batched decode: the activation in the stream kernel's epilogue (the norm, if any, as its own launch first)
            Tensor xin = input.contiguous();
            if (norm) {
                Tensor normed = torch::empty_like(xin);
                layernorm_forward(xin, std::get<0>(*norm), normed, std::get<1>(*norm));
                xin = normed;
            }
            check_epilogue(input, bias, residual, rows, n);
            Tensor           output = torch::empty(out_shape(input, n / 2), input.options());
            c10::DeviceGuard guard(input.device());
            check(eetq_w8a16_gemm_glu8(xin.data_ptr(), weight.data_ptr<int8_t>(), scale.data_ptr(),
                                       bias ? bias->data_ptr() : nullptr, output.data_ptr(), (int)rows, (int)n, (int)kw,
                                       stream_of(input)));
            return output;
        }
        if (rows > 16 && path == "auto" && input.size(-1) == kw && input.is_cuda() && input.scalar_type() == at::kHalf &&
            weight.scalar_type() == at::kChar && scale.scalar_type() == at::kHalf && weight.is_contiguous() &&
            weight.device() == input.device() && scale.device() == input.device()) {
            // prompts: the tiled MFMA kernel writes the activation out of its fp16 tile image where AUTO runs the shape on it
            Tensor xin = input.contiguous();
            if (norm) {
                Tensor normed = torch::empty_like(xin);
                layernorm_forward(xin, std::get<0>(*norm), normed, std::get<1>(*norm));
                xin = normed;
            }
            check_epilogue(input, bias, residual, rows, n);
            Tensor           output = torch::empty(out_shape(input, n / 2), input.options());
            c10::DeviceGuard guard(input.device());
            const int        st = eetq_w8a16_gemm_glu8(xin.data_ptr(), weight.data_ptr<int8_t>(), scale.data_ptr(),
                                                       bias ? bias->data_ptr() : nullptr, output.data_ptr(), (int)rows, (int)n,
                                                       (int)kw, stream_of(input));
            if (st != EETQ_ERR_UNSUPPORTED) {
                check(st);
                return output;
            }
            return silu_mul(w8_a16_gemm(xin, weight, scale, path, bias, residual, std::nullopt, false, std::string()), true);
        }
        return silu_mul(w8_a16_gemm(input_in, weight, scale, path, bias, residual, norm, false, std::string()), true);
    }
    if (gated) {
        TORCH_CHECK(input.size(-1) == 2 * kw, "w8_a16_gemm: gated input must be [..., 2K] for a [K, N] weight");
        const int64_t rows = input.numel() / input.size(-1);
        if (rows == 1 && path == "auto" && !norm && input.is_cuda() && input.scalar_type() == at::kHalf &&
            input.is_contiguous() && kw % 8 == 0 && weight.size(1) == n && activation.empty()) {
            TORCH_CHECK(weight.scalar_type() == at::kChar && scale.scalar_type() == at::kHalf && weight.is_contiguous(),
                        "w8_a16_gemm: weight must be contiguous int8 and scale float16");
            Tensor output = torch::empty(out_shape(input, n), input.options());
            check_epilogue(input, bias, residual, 1, n);
            c10::DeviceGuard guard(input.device());
            check(eetq_w8a16_gemv_silu_gated(input.data_ptr(), weight.data_ptr<int8_t>(), scale.data_ptr(),
                                             bias ? bias->data_ptr() : nullptr, residual ? residual->data_ptr() : nullptr,
                                             output.data_ptr(), (int)n, (int)kw, stream_of(input)));
            return output;
        }
        input = silu_mul(input.contiguous());
    }
    const int64_t k = input.size(-1);
    TORCH_CHECK(kw == k, "w8_a16_gemm: weight is [", kw, ", ", n, "] but input has K=", k);
    const int64_t m      = k ? input.numel() / k : 0;
    Tensor        output = torch::empty(out_shape(input, n), input.options());
    if (m == 0) return output;
    if (norm) {
        const Tensor& gamma = std::get<0>(*norm);
        const double  eps   = std::get<1>(*norm);
        if (m == 1 && path == "auto" && gamma.scalar_type() == at::kHalf && gamma.is_contiguous() && gamma.numel() == k &&
            weight.size(1) == n && activation.empty()) {
            TORCH_CHECK(input.scalar_type() == at::kHalf && input.is_cuda(), "w8_a16_gemm: input must be a float16 CUDA tensor");
            TORCH_CHECK(weight.scalar_type() == at::kChar && scale.scalar_type() == at::kHalf && weight.is_contiguous(),
                        "w8_a16_gemm: weight must be contiguous int8 and scale float16");
            TORCH_CHECK(gamma.device() == input.device() && weight.device() == input.device() &&
                            scale.device() == input.device(),
                        "w8_a16_gemm: all tensors must be on the input's device");
            check_epilogue(input, bias, residual, 1, n);
            Tensor           x = input.contiguous();
            c10::DeviceGuard guard(input.device());
            check(eetq_w8a16_gemv_rmsnorm(x.data_ptr(), gamma.data_ptr(), (float)eps, weight.data_ptr<int8_t>(),
                                          scale.data_ptr(), bias ? bias->data_ptr() : nullptr,
                                          residual ? residual->data_ptr() : nullptr, output.data_ptr(), (int)n, (int)k,
                                          stream_of(input)));
            return output;
        }
        Tensor xin    = input.contiguous();
        Tensor normed = torch::empty_like(xin);
        layernorm_forward(xin, gamma, normed, eps);
        input = normed;
    }
    return gemm_launch(input, weight, scale, output, m, n, k, path_id(path), bias, residual, act_id(activation));
}

// reference: w8_a16_gemm_forward_cuda_, fpA_intB_gemm_wrapper.cu:176-202 (writes into `output`, returns it)
Tensor w8_a16_gemm_(const Tensor& input, const Tensor& weight, const Tensor& scale, Tensor& output, int64_t m, int64_t n,
                    int64_t k)
{
    return gemm_launch(input, weight, scale, output, m, n, k, EETQ_PATH_AUTO, std::nullopt, std::nullopt, EETQ_ACT_IDENTITY);
}

// ---- side ops ---------------------------------------------------------------------------------------------------------
// reference: layernorm_forward_cuda, layernorm.cu:98-113 (returns void; current stream here, default stream there)
void layernorm_forward(const Tensor& input, const Tensor& gamma, Tensor& out, double eps)
{
    TORCH_CHECK(input.scalar_type() == at::kHalf && gamma.scalar_type() == at::kHalf && out.scalar_type() == at::kHalf,
                "layernorm_forward: expected scalar type Half");
    TORCH_CHECK(input.is_cuda() && gamma.is_cuda() && out.is_cuda(), "layernorm_forward: tensors must be CUDA tensors");
    TORCH_CHECK(input.is_contiguous() && gamma.is_contiguous() && out.is_contiguous(),
                "layernorm_forward: tensors must be contiguous");
    const int64_t cols = input.size(-1);
    const int64_t rows = cols ? input.numel() / cols : 0;
    TORCH_CHECK(gamma.numel() == cols && out.numel() == input.numel(), "layernorm_forward: shape mismatch");
    c10::DeviceGuard guard(input.device());
    check(eetq_rmsnorm_f16(input.data_ptr(), gamma.data_ptr(), out.data_ptr(), (float)eps, (int)rows, (int)cols,
                           stream_of(input)));
}

// reference: rotary_embedding_neox, pos_encoding_kernels.cu:55-87 (float / double / half; bf16 is out of scope)
void rotary_embedding_neox(const Tensor& positions, Tensor& query, Tensor& key, int64_t head_size, const Tensor& cos_sin_cache)
{
    const auto st = query.scalar_type();
    TORCH_CHECK(st == at::kHalf || st == at::kFloat || st == at::kDouble,
                "eetq_amd: rotary_embedding_neox is implemented for float16, float32 and float64");
    TORCH_CHECK(key.scalar_type() == st && cos_sin_cache.scalar_type() == st,
                "rotary_embedding_neox: query, key and cos_sin_cache must share one dtype");
    TORCH_CHECK(positions.scalar_type() == at::kLong, "rotary_embedding_neox: positions must be int64");
    TORCH_CHECK(query.is_contiguous() && key.is_contiguous() && cos_sin_cache.is_contiguous() && positions.is_contiguous(),
                "rotary_embedding_neox: tensors must be contiguous");
    TORCH_CHECK(query.dim() >= 3, "rotary_embedding_neox: query must be [batch, seq, heads, head_size] or [tokens, heads, head_size]");
    const int64_t tokens = positions.numel();
    const int64_t heads  = query.size(-2);
    c10::DeviceGuard guard(query.device());
    const int dt = st == at::kHalf ? EETQ_DTYPE_F16 : (st == at::kFloat ? EETQ_DTYPE_F32 : EETQ_DTYPE_F64);
    check(eetq_rotary_neox(positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(), cos_sin_cache.data_ptr(), dt,
                           (int)tokens, (int)heads, (int)head_size, (int)cos_sin_cache.size(1), stream_of(query)));
}

// tokens / heads / token stride of a [..., heads, head_size] view whose leading dimensions collapse to one stride
std::tuple<int64_t, int64_t, int64_t> token_view(const Tensor& t, int64_t head_size)
{
    TORCH_CHECK(t.dim() >= 2, "rotary_embedding_neox_strided: expected [..., heads, head_size]");
    const int64_t heads = t.size(-2), hs = t.size(-1);
    TORCH_CHECK(hs == head_size && t.stride(-1) == 1 && t.stride(-2) == hs,
                "rotary_embedding_neox_strided: the last two dimensions must be dense [heads, head_size]");
    int64_t tokens = 1, stride = -1, expect = -1;
    for (int64_t d = t.dim() - 3; d >= 0; --d) {
        tokens *= t.size(d);
        if (t.size(d) == 1) continue;
        if (stride < 0) {
            stride = t.stride(d);
            expect = stride * t.size(d);
        } else {
            TORCH_CHECK(t.stride(d) == expect, "rotary_embedding_neox_strided: leading dimensions do not collapse to one stride");
            expect = t.stride(d) * t.size(d);
        }
    }
    return {tokens, heads, stride >= 0 ? stride : heads * hs};
}

void rotary_embedding_neox_strided(const Tensor& positions, Tensor& query, Tensor& key, int64_t head_size,
                                   const Tensor& cos_sin_cache)
{
    TORCH_CHECK(query.scalar_type() == at::kHalf && key.scalar_type() == at::kHalf && cos_sin_cache.scalar_type() == at::kHalf,
                "eetq_amd: rotary_embedding_neox is implemented for float16 only");
    TORCH_CHECK(positions.scalar_type() == at::kLong && positions.is_contiguous() && cos_sin_cache.is_contiguous(),
                "rotary_embedding_neox_strided: positions must be contiguous int64, the cache contiguous");
    auto [tq, hq, sq] = token_view(query, head_size);
    auto [tk, hk, sk] = token_view(key, head_size);
    TORCH_CHECK(tq == tk && positions.numel() == tq,
                "rotary_embedding_neox_strided: query, key and positions disagree on the token count");
    c10::DeviceGuard guard(query.device());
    check(eetq_rotary_neox_strided_f16(positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(),
                                       cos_sin_cache.data_ptr(), (int)tq, (int)hq, (int)hk, (int)head_size,
                                       (int)cos_sin_cache.size(1), (int)sq, (int)sk, stream_of(query)));
}

void rotary_embedding_neox_kvcache(const Tensor& positions, Tensor& query, const Tensor& key, const Tensor& value,
                                   int64_t head_size, const Tensor& cos_sin_cache, Tensor& key_cache, Tensor& value_cache,
                                   const OptTensor& slots)
{
    for (const Tensor* t : std::initializer_list<const Tensor*>{&query, &key, &value, &cos_sin_cache, &key_cache, &value_cache})
        TORCH_CHECK(t->scalar_type() == at::kHalf, "rotary_embedding_neox_kvcache: float16 tensors expected");
    TORCH_CHECK(positions.scalar_type() == at::kLong && positions.is_contiguous(),
                "rotary_embedding_neox_kvcache: positions must be contiguous int64");
    TORCH_CHECK(query.dim() == 3 && key.dim() == 3 && value.dim() == 3 && key_cache.dim() == 4,
                "rotary_embedding_neox_kvcache: shape mismatch");
    const int64_t B = query.size(0), H = query.size(1), D = query.size(2), Hkv = key.size(1);
    TORCH_CHECK(key.size(0) == B && key.size(2) == D && value.sizes() == key.sizes() && D == head_size &&
                    key_cache.size(0) == B && key_cache.size(1) == Hkv && key_cache.size(3) == D &&
                    value_cache.sizes() == key_cache.sizes() && value_cache.strides() == key_cache.strides() &&
                    positions.numel() == B,
                "rotary_embedding_neox_kvcache: shape mismatch");
    for (const Tensor* t : std::initializer_list<const Tensor*>{&query, &key, &value})
        TORCH_CHECK(t->stride(-1) == 1 && t->stride(-2) == D, "rotary_embedding_neox_kvcache: [heads, head_size] must be dense");
    TORCH_CHECK(key_cache.stride(-1) == 1 && cos_sin_cache.is_contiguous(), "rotary_embedding_neox_kvcache: cache rows must be dense");
    int slot_stride = 0;
    if (slots) {
        const Tensor& s = *slots;
        TORCH_CHECK(s.scalar_type() == at::kLong && s.device() == query.device() && (s.numel() == 1 || s.numel() == B) &&
                        s.is_contiguous(),
                    "rotary_embedding_neox_kvcache: slots must be contiguous int64 on the device, 1 or B elements");
        slot_stride = (s.numel() == B && B > 1) ? 1 : 0;
    }
    const long strides[6] = {(long)query.stride(0), (long)key.stride(0), (long)value.stride(0), (long)key_cache.stride(0),
                             (long)key_cache.stride(1), (long)key_cache.stride(2)};
    c10::DeviceGuard guard(query.device());
    check(eetq_rotary_neox_kvcache_f16(positions.data_ptr<int64_t>(), slots ? slots->data_ptr<int64_t>() : nullptr,
                                       slot_stride, query.data_ptr(), key.data_ptr(), value.data_ptr(),
                                       cos_sin_cache.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)B, (int)H,
                                       (int)Hkv, (int)head_size, (int)cos_sin_cache.size(1), strides, (int)key_cache.size(2),
                                       stream_of(query)));
}

// Prefill on a pre-allocated KV cache (extension): query [B, T, H, D] rotated in place, key [B, T, Hkv, D] rotated into
// key_cache[b, :, base + t], value copied to value_cache[b, :, base + t], base = first_row_dev (device int64) or first_row:
// eetq_rotary_neox_kvcache_prefill_f16
void rotary_embedding_neox_kvcache_prefill(const Tensor& positions, Tensor& query, const Tensor& key, const Tensor& value,
                                           int64_t head_size, const Tensor& cos_sin_cache, Tensor& key_cache,
                                           Tensor& value_cache, int64_t first_row, const OptTensor& first_row_dev)
{
    const char* name = "rotary_embedding_neox_kvcache_prefill: ";
    for (const Tensor* t : std::initializer_list<const Tensor*>{&query, &key, &value, &cos_sin_cache, &key_cache, &value_cache}) {
        TORCH_CHECK(t->scalar_type() == at::kHalf, name, "float16 tensors expected");
        TORCH_CHECK(t->is_cuda() && t->device() == query.device(), name, "all tensors must be on one CUDA device");
    }
    TORCH_CHECK(positions.scalar_type() == at::kLong && positions.is_contiguous() && positions.device() == query.device(),
                name, "positions must be contiguous int64 on the device");
    TORCH_CHECK(query.dim() == 4 && key.dim() == 4 && value.dim() == 4 && key_cache.dim() == 4, name, "shape mismatch");
    const int64_t B = query.size(0), T = query.size(1), H = query.size(2), D = query.size(3), Hkv = key.size(2);
    TORCH_CHECK(key.size(0) == B && key.size(1) == T && key.size(3) == D && value.sizes() == key.sizes() && D == head_size &&
                    key_cache.size(0) == B && key_cache.size(1) == Hkv && key_cache.size(3) == D &&
                    value_cache.sizes() == key_cache.sizes() && value_cache.strides() == key_cache.strides() &&
                    positions.numel() == B * T && first_row >= 0 && (first_row_dev || first_row + T <= key_cache.size(2)),
                name, "shape mismatch");
    if (first_row_dev)
        TORCH_CHECK(first_row_dev->scalar_type() == at::kLong && first_row_dev->numel() == 1 &&
                        first_row_dev->device() == query.device(),
                    name, "first_row_dev must be one int64 on the device");
    for (const Tensor* t : std::initializer_list<const Tensor*>{&query, &key, &value})
        TORCH_CHECK(t->stride(-1) == 1 && t->stride(-2) == D && (B == 1 || T == 1 || t->stride(0) == T * t->stride(1)), name,
                    "[heads, head_size] must be dense and the tokens of all rows one stride apart");
    const int tdim = T == 1 ? 0 : 1;  // (a size-1 dimension's stride is arbitrary)
    TORCH_CHECK(key_cache.stride(-1) == 1 && cos_sin_cache.is_contiguous(), name, "cache rows must be dense");
    const long strides[6] = {(long)query.stride(tdim), (long)key.stride(tdim), (long)value.stride(tdim), (long)key_cache.stride(0),
                             (long)key_cache.stride(1), (long)key_cache.stride(2)};
    c10::DeviceGuard guard(query.device());
    check(eetq_rotary_neox_kvcache_prefill_f16(positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(), value.data_ptr(),
                                               cos_sin_cache.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)B,
                                               (int)T, first_row_dev ? first_row_dev->data_ptr<int64_t>() : nullptr,
                                               (int)first_row, (int)H, (int)Hkv, (int)head_size,
                                               (int)cos_sin_cache.size(1), strides, (int)key_cache.size(2), stream_of(query)));
}

// Greedy decode hand-over (extension): argmax of logits [B, V] (fp16, dense rows) -> out_tokens[:, column] and next_token, then
// position += 1, column += 1, all on the device in one launch (eetq_greedy_handover_f16)
void greedy_handover(const Tensor& logits, Tensor& out_tokens, Tensor& column, Tensor& next_token, Tensor& position)
{
    const char* name = "greedy_handover: ";
    TORCH_CHECK(logits.is_cuda() && logits.scalar_type() == at::kHalf && logits.dim() == 2 && logits.stride(1) == 1, name,
                "logits must be a float16 CUDA tensor [B, V] with dense rows");
    const int64_t B = logits.size(0), V = logits.size(1);
    for (const Tensor* t : std::initializer_list<const Tensor*>{&out_tokens, &column, &next_token, &position})
        TORCH_CHECK(t->scalar_type() == at::kLong && t->device() == logits.device(), name, "int64 tensors on the logits' device expected");
    TORCH_CHECK(out_tokens.dim() == 2 && out_tokens.size(0) == B && out_tokens.stride(1) == 1 && next_token.numel() == B &&
                    next_token.is_contiguous() && column.numel() == 1 && position.numel() == 1 && V > 0,
                name, "shape mismatch");
    c10::DeviceGuard guard(logits.device());
    check(eetq_greedy_handover_f16(logits.data_ptr(), (long)logits.stride(0), (int)V, (int)B, out_tokens.data_ptr<int64_t>(),
                                   (long)out_tokens.stride(0), (int)out_tokens.size(1), column.data_ptr<int64_t>(),
                                   next_token.data_ptr<int64_t>(), position.data_ptr<int64_t>(), stream_of(logits)));
}

Tensor decode_attention(const Tensor& query, const Tensor& key_cache, const Tensor& value_cache, const OptTensor& mask,
                        std::optional<double> scaling, std::optional<int64_t> splits_in, const OptTensor& kv_len,
                        int64_t kv_len_bias, const OptTensor& advance)
{
    TORCH_CHECK(query.scalar_type() == at::kHalf && key_cache.scalar_type() == at::kHalf && value_cache.scalar_type() == at::kHalf,
                "decode_attention: query and caches must be float16");
    TORCH_CHECK(query.dim() == 3 && key_cache.dim() == 4 && value_cache.sizes() == key_cache.sizes(),
                "decode_attention: expected query [B, H, D] and caches [B, Hkv, S, D]");
    const int64_t B = query.size(0), H = query.size(1), D = query.size(2);
    const int64_t Hkv = key_cache.size(1), S = key_cache.size(2);
    TORCH_CHECK(key_cache.size(0) == B && key_cache.size(3) == D && H % Hkv == 0 && query.stride(-1) == 1 &&
                    key_cache.stride(-1) == 1 && value_cache.stride(-1) == 1,
                "decode_attention: shape / stride mismatch");
    Tensor  mrow;
    int64_t m_sb = 0;
    if (mask) {
        mrow = mask->dim() != 2 ? mask->reshape({mask->size(0), -1}) : *mask;
        TORCH_CHECK(mrow.scalar_type() == at::kHalf && mrow.size(-1) >= S && mrow.stride(-1) == 1 && mrow.device() == query.device(),
                    "decode_attention: mask must be additive float16 with a dense last dimension >= S");
        TORCH_CHECK(mrow.size(0) == 1 || mrow.size(0) == B,
                    "decode_attention: the mask needs one row per batch entry (or a single shared row)");
        m_sb = (mrow.size(0) == B && B > 1) ? mrow.stride(0) : 0;
    }
    for (const OptTensor* t : std::initializer_list<const OptTensor*>{&kv_len, &advance})
        if (*t)
            TORCH_CHECK((*t)->scalar_type() == at::kLong && (*t)->numel() == 1 && (*t)->device() == query.device(),
                        "decode_attention: kv_len / advance must be a one-element int64 tensor on the query's device");
    const double  sc = scaling ? *scaling : 1.0 / std::sqrt((double)D);
    int64_t       splits = splits_in ? *splits_in
                                     : (int64_t)eetq_decode_attention_splits((int)B, (int)H, (int)S);
    Tensor        out = torch::empty({B, H, D}, query.options());
    Tensor        ws  = torch::empty({B * H * splits * (D + 4)}, query.options().dtype(at::kFloat));
    const long    strides[11] = {(long)query.stride(0),       (long)query.stride(1),       (long)key_cache.stride(0),
                                 (long)key_cache.stride(1),   (long)key_cache.stride(2),   (long)value_cache.stride(0),
                                 (long)value_cache.stride(1), (long)value_cache.stride(2), (long)m_sb,
                                 (long)out.stride(0),         (long)out.stride(1)};
    c10::DeviceGuard guard(query.device());
    check(eetq_decode_attention_f16(query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                                    mrow.defined() ? mrow.data_ptr() : nullptr, out.data_ptr(), ws.data_ptr<float>(), (int)B,
                                    (int)H, (int)Hkv, (int)S, (int)D, (int)splits, (float)sc, strides,
                                    kv_len ? kv_len->data_ptr<int64_t>() : nullptr, (int)kv_len_bias,
                                    advance ? advance->data_ptr<int64_t>() : nullptr, stream_of(query)));
    return out;
}

Tensor silu_mul(const Tensor& gate_up, bool glu8)
{
    TORCH_CHECK(gate_up.scalar_type() == at::kHalf && gate_up.is_cuda() && gate_up.is_contiguous(),
                "silu_mul: expected a contiguous float16 CUDA tensor");
    const int64_t inter = gate_up.size(-1) / 2;
    TORCH_CHECK(gate_up.size(-1) == 2 * inter && inter % 8 == 0, "silu_mul: last dimension must be 2*I with I a multiple of 8");
    Tensor        out  = torch::empty(out_shape(gate_up, inter), gate_up.options());
    const int64_t rows = inter ? out.numel() / inter : 0;
    c10::DeviceGuard guard(gate_up.device());
    if (glu8)
        check(eetq_silu_mul_glu8_f16(gate_up.data_ptr(), out.data_ptr(), (int)rows, (int)inter, stream_of(gate_up)));
    else
        check(eetq_silu_mul_f16(gate_up.data_ptr(), out.data_ptr(), (int)rows, (int)inter, stream_of(gate_up)));
    return out;
}

}  // namespace

// One decode step (one new token per batch row) of an accelerated Llama decoder layer over a pre-allocated KV cache, as ONE
// call: the six launches the Python blocks issue (eetq_amd/modules/llama_modules.py: EETLlamaAttention.forward +
// EETLlamaMLP.forward under replace_with_eet_fused_residual), in the same order with the same arguments -- so the result is
// bit-identical -- without ~100 us of interpreter time per layer.  Host-side only; nothing here touches the device directly.
//   qkv  = W8A16(rmsnorm(hidden))                      (norm inside the GEMV launch for a single row)
//   attn = rope_decode_attention(qkv)                  (rotary + cache write + attention, advances `counter`)
//   h    = hidden + W8A16_o(attn)                      (residual in the epilogue)
//   out  = h + W8A16_down(silu(gate) * up),  gate|up = W8A16(rmsnorm(h))
using NormArg = std::tuple<Tensor, double>;
Tensor llama_decode_layer(const Tensor& hidden, const NormArg& input_norm, const Tensor& qkv_w, const Tensor& qkv_s,
                          const OptTensor& qkv_b, const Tensor& positions, const Tensor& cos_sin_cache, Tensor& key_cache,
                          Tensor& value_cache, Tensor& tickets, Tensor& counter, const OptTensor& mask, double scaling,
                          int64_t heads, int64_t kv_heads, const Tensor& o_w, const Tensor& o_s, const OptTensor& o_b,
                          const NormArg& post_norm, const Tensor& gu_w, const Tensor& gu_s, const OptTensor& gu_b,
                          const Tensor& down_w, const Tensor& down_s, const OptTensor& down_b, bool glu8)
{
    TORCH_CHECK(hidden.dim() == 3 && hidden.size(1) == 1, "llama_decode_layer: hidden must be [B, 1, C]");
    const int64_t B = hidden.size(0), total = heads + 2 * kv_heads;
    if (B == 1) {
        // batch 1: straight onto the C ABI -- one device guard, one stream query, one scratch allocation; the same six
        // launches with the same arguments as the generic composition below (bit-identical by construction)
        const int64_t C = hidden.size(2), NQ = qkv_s.numel(), I2 = gu_s.numel(), I = I2 / 2;
        const Tensor &g1 = std::get<0>(input_norm), &g2 = std::get<0>(post_norm);
        TORCH_CHECK(total > 0 && NQ % total == 0, "llama_decode_layer: the QKV width is not (heads + 2 kv_heads) * D");
        const int64_t D = NQ / total, S = key_cache.size(2);
        for (const Tensor* t : std::initializer_list<const Tensor*>{&hidden, &g1, &g2, &qkv_s, &o_s, &gu_s, &down_s, &cos_sin_cache,
                                                                    &key_cache, &value_cache})
            TORCH_CHECK(t->scalar_type() == at::kHalf && t->is_cuda() && t->is_contiguous() && t->device() == hidden.device(),
                        "llama_decode_layer: float16 contiguous tensors on the hidden state's device expected");
        for (const Tensor* t : std::initializer_list<const Tensor*>{&qkv_w, &o_w, &gu_w, &down_w})
            TORCH_CHECK(t->scalar_type() == at::kChar && t->is_contiguous() && t->dim() == 2 && t->device() == hidden.device(),
                        "llama_decode_layer: weights must be contiguous int8 [K, N] on the hidden state's device");
        for (const OptTensor* t : std::initializer_list<const OptTensor*>{&qkv_b, &o_b, &gu_b, &down_b})
            TORCH_CHECK(!*t || ((*t)->scalar_type() == at::kHalf && (*t)->is_contiguous() && (*t)->device() == hidden.device()),
                        "llama_decode_layer: biases must be contiguous float16 on the hidden state's device");
        TORCH_CHECK(qkv_w.size(0) == C && qkv_w.size(1) == NQ && g1.numel() == C && g2.numel() == C && o_w.size(0) == heads * D &&
                        o_w.size(1) == C && o_s.numel() == C && gu_w.size(0) == C && gu_w.size(1) == I2 && I2 == 2 * I &&
                        I % 8 == 0 && down_w.size(0) == I && down_w.size(1) == C && down_s.numel() == C &&
                        (!qkv_b || qkv_b->numel() == NQ) && (!o_b || o_b->numel() == C) && (!gu_b || gu_b->numel() == I2) &&
                        (!down_b || down_b->numel() == C),
                    "llama_decode_layer: weight shapes do not match the hidden size / head geometry");
        TORCH_CHECK(key_cache.dim() == 4 && key_cache.size(0) == 1 && key_cache.size(1) == kv_heads && key_cache.size(3) == D &&
                        value_cache.sizes() == key_cache.sizes() && cos_sin_cache.size(-1) == D && heads % kv_heads == 0,
                    "llama_decode_layer: cache [1, kv_heads, S, D] / rotation table [positions, D] expected");
        TORCH_CHECK(positions.scalar_type() == at::kLong && positions.numel() == 1 && positions.device() == hidden.device() &&
                        counter.scalar_type() == at::kLong && counter.numel() == 1 && counter.device() == hidden.device() &&
                        tickets.scalar_type() == at::kInt && tickets.is_contiguous() && tickets.numel() >= heads + 1 &&
                        tickets.device() == hidden.device(),
                    "llama_decode_layer: positions / counter must be one-element int64, tickets int32 [>= heads + 1], on the device");
        const void* mrow = nullptr;
        if (mask) {
            TORCH_CHECK(mask->scalar_type() == at::kHalf && mask->size(-1) >= S && mask->stride(-1) == 1 &&
                            mask->numel() == mask->size(-1) && mask->device() == hidden.device(),
                        "llama_decode_layer: mask must be one additive float16 row of at least S entries");
            mrow = mask->data_ptr();
        }
        const int64_t splits = eetq_decode_attention_splits(1, (int)heads, (int)S);
        // fp16 scratch: qkv | attention output | gate|up | activation ; fp32 scratch: the attention chunk records
        const int64_t n16 = NQ + heads * D + I2 + I, n32 = heads * splits * (D + 4);
        Tensor  scratch = torch::empty({n16 * 2 + n32 * 4 + 16}, hidden.options().dtype(at::kByte));
        Tensor  h = torch::empty_like(hidden), out = torch::empty_like(hidden);
        char*   base = static_cast<char*>(scratch.data_ptr());
        void *  qkv = base, *att = base + NQ * 2, *gu = base + (NQ + heads * D) * 2, *act = base + (NQ + heads * D + I2) * 2;
        float*  ws  = reinterpret_cast<float*>(base + ((n16 * 2 + 15) / 16) * 16);
        const long st[12] = {(long)NQ, (long)NQ, (long)NQ, (long)key_cache.stride(0), (long)key_cache.stride(1),
                             (long)key_cache.stride(2), (long)value_cache.stride(0), (long)value_cache.stride(1),
                             (long)value_cache.stride(2), 0, (long)(heads * D), (long)D};
        auto optp = [](const OptTensor& t) -> const void* { return t ? t->data_ptr() : nullptr; };
        c10::DeviceGuard guard(hidden.device());
        void*            stream = stream_of(hidden);
        int64_t*         cnt    = counter.data_ptr<int64_t>();
        check(eetq_w8a16_gemv_rmsnorm(hidden.data_ptr(), g1.data_ptr(), (float)std::get<1>(input_norm), qkv_w.data_ptr<int8_t>(),
                                      qkv_s.data_ptr(), optp(qkv_b), nullptr, qkv, (int)NQ, (int)C, stream));
        const char* q = static_cast<const char*>(qkv);
        check(eetq_rope_decode_attention_f16(positions.data_ptr<int64_t>(), cnt, 0, q, q + heads * D * 2,
                                             q + (heads + kv_heads) * D * 2, cos_sin_cache.data_ptr(), key_cache.data_ptr(),
                                             value_cache.data_ptr(), mrow, att, ws,
                                             reinterpret_cast<unsigned*>(tickets.data_ptr<int32_t>()), 1, (int)heads,
                                             (int)kv_heads, (int)S, (int)D, (int)splits, (float)scaling, st, cnt, 1, cnt, stream));
        check(eetq_w8a16_gemm_act(att, o_w.data_ptr<int8_t>(), o_s.data_ptr(), optp(o_b), hidden.data_ptr(), h.data_ptr(), 1,
                                  (int)C, (int)(heads * D), EETQ_PATH_AUTO, EETQ_ACT_IDENTITY, stream));
        if (glu8) {  // gate|up in "glu8" column order: the activation rides in the projection's epilogue
            check(eetq_w8a16_gemv_glu8(h.data_ptr(), g2.data_ptr(), (float)std::get<1>(post_norm), gu_w.data_ptr<int8_t>(),
                                       gu_s.data_ptr(), optp(gu_b), act, (int)I2, (int)C, stream));
        } else {
            check(eetq_w8a16_gemv_rmsnorm(h.data_ptr(), g2.data_ptr(), (float)std::get<1>(post_norm), gu_w.data_ptr<int8_t>(),
                                          gu_s.data_ptr(), optp(gu_b), nullptr, gu, (int)I2, (int)C, stream));
            check(eetq_silu_mul_f16(gu, act, 1, (int)I, stream));
        }
        check(eetq_w8a16_gemm_act(act, down_w.data_ptr<int8_t>(), down_s.data_ptr(), optp(down_b), h.data_ptr(), out.data_ptr(),
                                  1, (int)C, (int)I, EETQ_PATH_AUTO, EETQ_ACT_IDENTITY, stream));
        return out;
    }
    const std::string autop = "auto", none;
    const std::optional<NormArg> n1(input_norm), n2(post_norm), no_norm;
    const OptTensor no_tensor;
    Tensor qkv = w8_a16_gemm(hidden, qkv_w, qkv_s, autop, qkv_b, no_tensor, n1, false, none);  // [B, 1, (H + 2 Hkv) D]
    TORCH_CHECK(total > 0 && qkv.size(-1) % total == 0, "llama_decode_layer: the QKV width is not (heads + 2 kv_heads) * D");
    const int64_t D = qkv.size(-1) / total;
    Tensor rows = qkv.view({B, total, D});
    Tensor attn = rope_decode_attention(positions, rows.narrow(1, 0, heads), rows.narrow(1, heads, kv_heads),
                                        rows.narrow(1, heads + kv_heads, kv_heads), cos_sin_cache, key_cache, value_cache,
                                        tickets, counter, mask, scaling, std::nullopt, counter, 1, counter);
    Tensor h  = w8_a16_gemm(attn.view({B, 1, heads * D}), o_w, o_s, autop, o_b, hidden, no_norm, false, none);
    Tensor act = glu8 ? w8_a16_gemm(h, gu_w, gu_s, autop, gu_b, no_tensor, n2, false, std::string("silu_glu8"))
                      : silu_mul(w8_a16_gemm(h, gu_w, gu_s, autop, gu_b, no_tensor, n2, false, none));
    return w8_a16_gemm(act, down_w, down_s, autop, down_b, h, no_norm, false, none);
}

// Grouped decode GEMV (extension): independent single-row problems in as few dispatches as possible (eetq_w8a16_gemv_grouped).
// inputs[i]: fp16 with K_i elements (any shape with one row); weights[i]: processed int8 [K_i, N_i]; scales[i]: fp16 [N_i].
// Returns fresh outputs shaped like inputs[i] with the last dimension replaced by N_i.
std::vector<Tensor> w8_a16_gemv_grouped(const std::vector<Tensor>& inputs, const std::vector<Tensor>& weights,
                                        const std::vector<Tensor>& scales, const std::optional<std::vector<OptTensor>>& biases,
                                        const std::optional<std::vector<OptTensor>>& residuals)
{
    const size_t n = inputs.size();
    TORCH_CHECK(weights.size() == n && scales.size() == n, "w8_a16_gemv_grouped: inputs, weights and scales must have one entry per problem");
    TORCH_CHECK(!biases || biases->size() == n, "w8_a16_gemv_grouped: one bias entry (or None) per problem");
    TORCH_CHECK(!residuals || residuals->size() == n, "w8_a16_gemv_grouped: one residual entry (or None) per problem");
    std::vector<Tensor>            outs, keep;
    std::vector<eetq_gemv_problem> probs(n);
    if (n == 0) return outs;
    const auto dev = inputs[0].device();
    for (size_t i = 0; i < n; ++i) {
        const Tensor& x = inputs[i];
        const Tensor& w = weights[i];
        const Tensor& s = scales[i];
        TORCH_CHECK(x.scalar_type() == at::kHalf, "w8_a16_gemm: input must be float16 (got ", x.scalar_type(), ")");
        TORCH_CHECK(x.is_cuda(), "input must be a CUDA tensor");
        TORCH_CHECK(w.scalar_type() == at::kChar && s.scalar_type() == at::kHalf, "w8_a16_gemm: weight must be int8 and scale float16");
        TORCH_CHECK(w.dim() == 2 && w.is_contiguous() && s.is_contiguous(), "w8_a16_gemm: weight [K, N] and scale must be contiguous");
        TORCH_CHECK(x.device() == dev && w.device() == dev && s.device() == dev, "w8_a16_gemv_grouped: all tensors must be on one device");
        const int64_t K = w.size(0), N = w.size(1);
        TORCH_CHECK(x.numel() == K && x.size(-1) == K, "w8_a16_gemv_grouped: every input must be ONE row of K elements");
        TORCH_CHECK(s.numel() == N, "w8_a16_gemm: scale must have N elements");
        Tensor xc = x.is_contiguous() ? x : x.contiguous();
        keep.push_back(xc);
        auto shape = x.sizes().vec();
        shape.back() = N;
        Tensor y = torch::empty(shape, x.options());
        outs.push_back(y);
        probs[i] = eetq_gemv_problem{xc.data_ptr(), w.data_ptr<int8_t>(), s.data_ptr(), y.data_ptr(), nullptr, nullptr, (int)N, (int)K};
        const OptTensor b = biases ? (*biases)[i] : OptTensor();
        const OptTensor r = residuals ? (*residuals)[i] : OptTensor();
        check_epilogue(xc, b, r, 1, N);
        if (b) probs[i].bias = b->data_ptr();
        if (r) probs[i].residual = r->data_ptr();
    }
    c10::DeviceGuard guard(dev);
    check(eetq_w8a16_gemv_grouped(probs.data(), (int)n, stream_of(inputs[0])));
    return outs;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "EETQ operator module on libeetq_amd.so (MI355X / gfx950)";
    // ---- the reference's six functions (csrc/eetpy.cpp:9-19): names, order, py::arg names and defaults kept ----------
    m.def("w8_a16_gemm", &w8_a16_gemm, "Weight only gemm", py::arg("input"), py::arg("weight"), py::arg("scale"),
          py::arg("path") = "auto", py::arg("bias") = py::none(), py::arg("residual") = py::none(),
          py::arg("norm") = py::none(), py::arg("gated") = false, py::arg("activation") = "");
    m.def("w8_a16_gemm_", &w8_a16_gemm_, "Weight only gemm inplace", py::arg("input"), py::arg("weight"), py::arg("scale"),
          py::arg("output"), py::arg("m"), py::arg("n"), py::arg("k"));
    m.def("preprocess_weights", &preprocess_weights, "transform_int8_weights_for_cutlass", py::arg("origin_weight"),
          py::arg("is_int4") = false, py::arg("layout") = "gfx950");
    m.def("quant_weights", &quant_weights, "quantize weight", py::arg("origin_weight"), py::arg("quant_type"),
          py::arg("return_unprocessed_quantized_tensor") = false, py::arg("layout") = "gfx950");
    m.def("rotary_embedding_neox", &rotary_embedding_neox, "Apply GPT-NeoX style rotary embedding to query and key",
          py::arg("positions"), py::arg("query"), py::arg("key"), py::arg("head_size"), py::arg("cos_sin_cache"));
    m.def("layernorm_forward", &layernorm_forward, "LayerNorm kernel", py::arg("input"), py::arg("gamma"), py::arg("out"),
          py::arg("eps"));
    // ---- extensions of this library ----------------------------------------------------------------------------------
    m.def("unprocess_weights", &unprocess_weights, "inverse of preprocess_weights", py::arg("processed_weight"),
          py::arg("layout") = "gfx950", py::arg("is_int4") = false);
    m.def("rotary_embedding_neox_strided", &rotary_embedding_neox_strided, "rotary embedding on strided q/k views",
          py::arg("positions"), py::arg("query"), py::arg("key"), py::arg("head_size"), py::arg("cos_sin_cache"));
    m.def("rotary_embedding_neox_kvcache", &rotary_embedding_neox_kvcache, "decode-step rotary + KV-cache write",
          py::arg("positions"), py::arg("query"), py::arg("key"), py::arg("value"), py::arg("head_size"),
          py::arg("cos_sin_cache"), py::arg("key_cache"), py::arg("value_cache"), py::arg("slots") = py::none());
    m.def("rotary_embedding_neox_kvcache_prefill", &rotary_embedding_neox_kvcache_prefill, "prompt rotary + KV-cache write",
          py::arg("positions"), py::arg("query"), py::arg("key"), py::arg("value"), py::arg("head_size"),
          py::arg("cos_sin_cache"), py::arg("key_cache"), py::arg("value_cache"), py::arg("first_row") = 0,
          py::arg("first_row_dev") = py::none());
    m.def("greedy_handover", &greedy_handover, "argmax + token hand-over of a greedy decode step", py::arg("logits"),
          py::arg("out_tokens"), py::arg("column"), py::arg("next_token"), py::arg("position"));
    m.def("decode_attention", &decode_attention, "single-query attention over a KV cache", py::arg("query"),
          py::arg("key_cache"), py::arg("value_cache"), py::arg("mask") = py::none(), py::arg("scaling") = py::none(),
          py::arg("splits") = py::none(), py::arg("kv_len") = py::none(), py::arg("kv_len_bias") = 0,
          py::arg("advance") = py::none());
    m.def("rope_decode_attention", &rope_decode_attention, py::arg("positions"), py::arg("query"), py::arg("key"),
          py::arg("value"), py::arg("cos_sin_cache"), py::arg("key_cache"), py::arg("value_cache"), py::arg("tickets"),
          py::arg("slots") = py::none(), py::arg("mask") = py::none(), py::arg("scaling") = py::none(),
          py::arg("splits") = py::none(), py::arg("kv_len") = py::none(), py::arg("kv_len_bias") = 0,
          py::arg("advance") = py::none());
    m.def("prefill_attention", &prefill_attention, py::arg("query"), py::arg("key"), py::arg("value"), py::arg("keys"),
          py::arg("scaling") = py::none(), py::arg("causal_offset") = py::none(),
          "causal attention of a prompt's query rows [B, T, H, D] over the first `keys` rows of a KV cache [B, Hkv, S, D] (MFMA)");
    m.def("prefill_attention_supported", &prefill_attention_supported, py::arg("head_dim"));
    m.def("llama_decode_layer", &llama_decode_layer, "one decode step of an accelerated Llama decoder layer (six launches, one call)",
          py::arg("hidden"), py::arg("input_norm"), py::arg("qkv_weight"), py::arg("qkv_scale"), py::arg("qkv_bias"),
          py::arg("positions"), py::arg("cos_sin_cache"), py::arg("key_cache"), py::arg("value_cache"), py::arg("tickets"),
          py::arg("counter"), py::arg("mask"), py::arg("scaling"), py::arg("heads"), py::arg("kv_heads"), py::arg("o_weight"),
          py::arg("o_scale"), py::arg("o_bias"), py::arg("post_norm"), py::arg("gate_up_weight"), py::arg("gate_up_scale"),
          py::arg("gate_up_bias"), py::arg("down_weight"), py::arg("down_scale"), py::arg("down_bias"), py::arg("glu8") = false);
    m.def("silu_mul", &silu_mul, "silu(gate) * up on a fused gate|up block (glu8: columns in groups of 8 gate + 8 up)",
          py::arg("gate_up"), py::arg("glu8") = false);
    m.def("w8_a16_gemv_grouped", &w8_a16_gemv_grouped, "independent single-row W8A16 problems in as few dispatches as possible",
          py::arg("inputs"), py::arg("weights"), py::arg("scales"), py::arg("biases") = py::none(),
          py::arg("residuals") = py::none());
    m.attr("__eetq_amd_version__") = eetq_version();
}
