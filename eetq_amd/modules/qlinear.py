"""Quantised linear modules over the W8A16 operators.

Host-side mirror of /root/reference/python/eetq/modules/qlinear.py: ``quantize_and_preprocess_weights``
(:14-24), ``W8A16Linear`` (:27-62), ``EetqLinearMMFunction`` (:64-94) and ``EetqLinear`` (:96-124) keep
their names, constructor arguments, buffer names/shapes/dtypes and forward semantics.  ``W8A16LoraLinear``
(:127-186) is dead code in the reference (never constructed successfully) and is not carried over.

Layouts.  In memory the int8 buffer (``qweight`` / ``weight``) holds this library's ``gfx950`` layout -- NOT the bytes a
CUDA build of the reference keeps there.  State dicts are interchangeable all the same: every module re-encodes its int8
buffer to the reference's processed layout (``sm80``) when ``state_dict()`` is taken and back when ``load_state_dict()``
runs (eetq_amd/checkpoint.py), so a checkpoint written here loads in CUDA-EETQ and an NVIDIA-written one loads
here.  Code that copies tensors straight into the buffers (bypassing ``load_state_dict``) must convert them itself:
``eetq_amd.utils.convert_model_layout_(model, "sm80")``.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ..ops import preprocess_weights, quant_weights, w8_a16_gemm
from ..checkpoint import install_layout_hooks

__all__ = ["quantize_and_preprocess_weights", "W8A16Linear", "W4A16Linear", "EetqLinearMMFunction", "EetqLinear"]


def quantize_and_preprocess_weights(weight, scales=None):
    """nn.Linear weight [out, in] -> (processed int8 [in, out], scales [out]).

    fp16 weights are quantised (per output channel, symmetric); int8 weights (e.g. bitsandbytes) are only
    re-laid-out and need caller-provided ``scales``.  Anything else raises ValueError, like the reference.
    Unlike the reference (which always round-trips through host memory, qlinear.py:16), a weight that
    already lives on the GPU is quantised in place on that GPU.
    """
    kn = torch.t(weight).contiguous()  # [K = in_features, N = out_features]
    if weight.dtype == torch.int8:
        assert scales is not None, "int8 weights need their scales"
        return preprocess_weights(kn), scales
    if weight.dtype == torch.float16:
        processed, scales = quant_weights(kn, torch.int8, False)
        return processed, scales
    raise ValueError("Unsupported data type: {}".format(weight.dtype))


class W8A16Linear(nn.Module):
    """Inference-only linear layer: int8 weight ``qweight`` [in, out], fp16 ``weight_scales`` [out]."""

    def __init__(self, in_features, out_features, bias=True, dev="cuda:0"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("qweight", torch.zeros((in_features, out_features), dtype=torch.int8, device=dev))
        self.register_buffer("weight_scales", torch.zeros((out_features,), dtype=torch.float16, device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features,), dtype=torch.float16, device=dev))
        else:
            self.bias = None
        # state dicts carry the reference's layout: re-encode on save / load (see the module docstring)
        self.checkpoint_layout = None   # None = the process-wide wire layout; "sm80" / "gfx950" pins what load expects
        install_layout_hooks(self, "qweight")

    @classmethod
    def from_torch(cls, linear, scales=None, init_only=False):
        dev = linear.weight.device
        mod = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, dev=dev)
        if init_only:  # buffers only; weights arrive later through load_state_dict
            return mod
        if linear.bias is not None:
            mod.bias = linear.bias.clone().half()
        qweight, scales = quantize_and_preprocess_weights(linear.weight, scales)
        mod.qweight = qweight.to(dev)
        mod.weight_scales = scales.half().to(dev)
        return mod

    @torch.no_grad()
    def forward(self, input, residual=None, norm=None, gated=False, activation=""):
        # bias is fused into the kernel epilogue: same bits as the reference's `output + self.bias` (qlinear.py:61);
        # `residual` (extension) is added after it in the same epilogue: the decoder block's `residual + proj(x)`
        # `norm=(gamma, eps)` (extension): RMS-norm of the input, fused into the launch for single-row inputs
        # `gated=True` (extension): input is a fused gate|up block and the projection runs on silu(gate) * up
        # `activation` (extension): "relu" / "gelu" / "silu" epilogues, "silu_glu8" for a gate|up weight in glu8 column order
        return w8_a16_gemm(input, self.qweight, self.weight_scales, bias=self.bias, residual=residual, norm=norm,
                           gated=gated, activation=activation)

    def extra_repr(self):
        return "in_features={}, out_features={}, bias={}".format(self.in_features, self.out_features,
                                                                 self.bias is not None)


class W4A16Linear(nn.Module):
    """int4 weight-only linear layer (extension; the reference binds int4 quantisation -- ``quant_weights(w, torch.quint4x2)``
    -- but no int4 GEMM): packed ``qweight`` int8 [in, out / 2] (two values per byte) in the gfx950 int4 layout, fp16
    ``weight_scales`` [out].  Needs in_features % 128 == 0 and out_features % 16 == 0."""

    def __init__(self, in_features, out_features, bias=True, dev="cuda:0"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("qweight", torch.zeros((in_features, out_features // 2), dtype=torch.int8, device=dev))
        self.register_buffer("weight_scales", torch.zeros((out_features,), dtype=torch.float16, device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features,), dtype=torch.float16, device=dev))
        else:
            self.bias = None

    @classmethod
    def from_torch(cls, linear, init_only=False):
        dev = linear.weight.device
        mod = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, dev=dev)
        if init_only:
            return mod
        if linear.weight.dtype != torch.float16:
            raise ValueError("Unsupported data type: {}".format(linear.weight.dtype))
        if linear.bias is not None:
            mod.bias = linear.bias.clone().half()
        qweight, scales = quant_weights(torch.t(linear.weight).contiguous(), torch.quint4x2, False)
        mod.qweight = qweight.to(dev)
        mod.weight_scales = scales.half().to(dev)
        return mod

    @torch.no_grad()
    def forward(self, input, residual=None):
        return w8_a16_gemm(input, self.qweight, self.weight_scales, bias=self.bias, residual=residual)

    def extra_repr(self):
        return "in_features={}, out_features={}, bias={}, bits=4".format(self.in_features, self.out_features,
                                                                        self.bias is not None)


class EetqLinearMMFunction(Function):
    """Autograd wrapper: forward = fused dequant GEMM; backward dequantises W by multiplying an identity
    (exact: every output element is a single product) and returns grad_input only."""

    @staticmethod
    def forward(ctx, x, weight, scales, bias=None):
        ctx.save_for_backward(x, weight, scales, bias)
        return w8_a16_gemm(x, weight, scales, bias=bias)

    @staticmethod
    def backward(ctx, grad_output):
        x, weight, scales, _bias = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            eye = torch.eye(weight.shape[0], device=weight.device, dtype=x.dtype)
            w_deq = w8_a16_gemm(eye, weight, scales)  # fp16 [K, N] == fp16(q * s)
            grad_input = grad_output.squeeze(0).matmul(w_deq.transpose(0, 1)).unsqueeze(0)
        return grad_input, None, None, None


class EetqLinear(nn.Module):
    """The module shape transformers/TGI instantiate: buffer ``weight`` int8 [in, out], ``weight_scales``
    registered later via :meth:`register_scale`, optional fp16 ``bias``."""

    def __init__(self, in_features, out_features, bias=True, device="cuda:0"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("weight", torch.zeros((in_features, out_features), dtype=torch.int8, device=device))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features,), dtype=torch.float16, device=device))
        else:
            self.bias = None
        self.checkpoint_layout = None
        install_layout_hooks(self, "weight")

    def register(self, buffer_name, tensor):
        self.register_buffer(buffer_name, tensor)

    def register_scale(self, device):
        n = self.weight.shape[-1]
        self.register_buffer("weight_scales", torch.zeros((n,), dtype=torch.float16, device=device))

    def forward(self, input):
        if self.training:
            return EetqLinearMMFunction.apply(input, self.weight, self.weight_scales, self.bias)
        with torch.no_grad():
            return EetqLinearMMFunction.apply(input, self.weight, self.weight_scales, self.bias)
