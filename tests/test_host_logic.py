"""Host-side logic that needs no GPU: the module/quantiser mirrors of the reference's Python layer and the
argument checks of the operator surface."""
import pytest
import torch
import torch.nn as nn


def test_alias_packages_expose_reference_names():
    import EETQ
    import eetq
    from eetq.modules.qlinear import W8A16Linear  # examples/layers/test_qlinear.py:10
    from eetq.utils import eet_quantize  # README.md:72-73
    for name in ("w8_a16_gemm", "w8_a16_gemm_", "preprocess_weights", "quant_weights", "rotary_embedding_neox",
                 "layernorm_forward"):  # csrc/eetpy.cpp:9-19
        assert callable(getattr(EETQ, name))
    assert W8A16Linear is eetq.W8A16Linear and callable(eet_quantize)


def test_w8a16linear_buffers_match_reference_contract():
    from eetq_amd.modules.qlinear import EetqLinear, W8A16Linear
    m = W8A16Linear(64, 32, bias=True, dev="cpu")
    sd = m.state_dict()
    assert set(sd) == {"qweight", "weight_scales", "bias"}
    assert sd["qweight"].shape == (64, 32) and sd["qweight"].dtype == torch.int8        # [in, out]
    assert sd["weight_scales"].shape == (32,) and sd["weight_scales"].dtype == torch.float16
    assert sd["bias"].dtype == torch.float16
    assert W8A16Linear(64, 32, bias=False, dev="cpu").bias is None
    lin = nn.Linear(64, 32, bias=True).half()
    init = W8A16Linear.from_torch(lin, init_only=True)                                     # no kernel call
    assert init.qweight.abs().sum() == 0 and init.in_features == 64 and init.out_features == 32
    e = EetqLinear(64, 32, bias=False, device="cpu")
    assert set(e.state_dict()) == {"weight"}
    e.register_scale("cpu")
    assert e.state_dict()["weight_scales"].shape == (32,)
    e.register("extra", torch.zeros(1))
    assert "extra" in e.state_dict()


def test_quantize_and_preprocess_rejects_other_dtypes():
    from eetq_amd.modules.qlinear import quantize_and_preprocess_weights
    with pytest.raises(ValueError):
        quantize_and_preprocess_weights(torch.zeros(4, 4, dtype=torch.float32))
    with pytest.raises(AssertionError):
        quantize_and_preprocess_weights(torch.zeros(4, 4, dtype=torch.int8))


def test_find_layers_and_set_op_by_name():
    from eetq_amd.utils.quantizer import find_layers, get_named_linears, set_op_by_name

    class Sub(nn.Linear):
        pass

    model = nn.Sequential()
    model.add_module("blocks", nn.ModuleList([nn.Sequential(nn.Linear(4, 4), nn.ReLU()), nn.Sequential(Sub(4, 4))]))
    model.add_module("lm_head", nn.Linear(4, 8))
    found = find_layers(model)
    assert list(found) == ["blocks.0.0"]                       # exact type match, lm_head excluded
    assert set(get_named_linears(model)) == {"blocks.0.0", "blocks.1.0"}
    assert set(find_layers(model, exclude=[])) == {"blocks.0.0", "lm_head"}
    new = nn.Identity()
    set_op_by_name(model, "blocks.0.0", new)
    assert model.blocks[0][0] is new
    set_op_by_name(model, "lm_head", new)
    assert model.lm_head is new


def test_eet_quantize_rejects_fp32_model():
    from eetq_amd.utils.quantizer import eet_quantize
    with pytest.raises(ValueError):
        eet_quantize(nn.Sequential(nn.Linear(64, 64)))


def test_ops_validate_before_touching_the_gpu():
    from eetq_amd import ops
    w = torch.zeros(64, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="int4 or int8"):
        ops.quant_weights(w, torch.float16)
    with pytest.raises(RuntimeError, match="not implemented"):
        ops.quant_weights(w, torch.quint4x2)
    with pytest.raises(RuntimeError, match="FP16 or FP32"):
        ops.quant_weights(w.to(torch.bfloat16), torch.int8)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.quant_weights(torch.zeros(64, 128, dtype=torch.float16)[:, ::2], torch.int8)
    with pytest.raises(RuntimeError, match="empty"):
        ops.quant_weights(torch.zeros(0, 64, dtype=torch.float16), torch.int8)
    with pytest.raises(RuntimeError, match="dim"):
        ops.quant_weights(torch.zeros(64, dtype=torch.float16), torch.int8)
    with pytest.raises(RuntimeError, match="2-D"):
        ops.quant_weights(torch.zeros(2, 64, 64, dtype=torch.float16), torch.int8)
    with pytest.raises(RuntimeError):
        ops.preprocess_weights(torch.zeros(64, 64, dtype=torch.int8), True)
    with pytest.raises(RuntimeError, match="unknown weight layout"):
        ops.preprocess_weights(torch.zeros(64, 64, dtype=torch.int8), False, "sm90")
    x = torch.zeros(1, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.w8_a16_gemm(x, torch.zeros(64, 64, dtype=torch.int8), torch.zeros(64, dtype=torch.float16))
    with pytest.raises(RuntimeError, match="float16"):
        ops.w8_a16_gemm(x.float(), torch.zeros(64, 64, dtype=torch.int8), torch.zeros(64, dtype=torch.float16))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):  # no silent CPU fallback
            ops.quant_weights(w, torch.int8)
