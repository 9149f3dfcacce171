"""The caller the north star names: transformers' own EETQ integration (`transformers/integrations/eetq.py`,
`quantizers/quantizer_eetq.py`; the reference's README.md:55-102 sends users there) running UNMODIFIED on this repo's
compiled `EETQ` module -- `from_pretrained(..., quantization_config=EetqConfig("int8"))`, transformers' `EetqLinear`
(int8 / fp16 nn.Parameters, not buffers), `EetqQuantize.convert` (CPU tensor in) and `EetqLinearMMFunction` forward and
backward.  eetq_amd.utils.hf.use_with_transformers() only redirects the kernel-hub lookup.  Checked against the oracle
(bit-exact bytes; tier A on the outputs), against fp16 nn.Linear (the reference's atol = 1e-2,
examples/layers/test_qlinear.py:36) and, for the wire format, against the oracle's sm80 writer."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
transformers = pytest.importorskip("transformers")


def _config():
    return transformers.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                    num_key_value_heads=4, vocab_size=1000, max_position_embeddings=256)


@pytest.fixture(scope="module")
def fp16_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tiny_llama_fp16"))
    torch.manual_seed(11)
    model = transformers.LlamaForCausalLM(_config()).half().eval()
    model.save_pretrained(d)
    return d


@pytest.fixture(scope="module")
def fp16_model(fp16_dir):
    return transformers.AutoModelForCausalLM.from_pretrained(fp16_dir, dtype=torch.float16).eval()


@pytest.fixture(scope="module")
def eetq_model(fp16_dir):
    from eetq_amd.utils.hf import use_with_transformers
    use_with_transformers()
    cfg = transformers.EetqConfig("int8")
    model = transformers.AutoModelForCausalLM.from_pretrained(fp16_dir, quantization_config=cfg, device_map=DEV,
                                                              dtype=torch.float16)
    return model.eval()


def _eetq_linears(model):
    from transformers.integrations.eetq import EetqLinear
    return {n: m for n, m in model.named_modules() if isinstance(m, EetqLinear)}


def test_quantize_on_load_is_bit_exact(oracle, eetq_model, fp16_model):
    """EetqQuantize.convert hands quant_weights a CPU [K, N] tensor (as the reference demands, fpA_intB_gemm_wrapper.cu:33);
    what lands in transformers' int8 nn.Parameter is the oracle's quantisation in the native layout, scales bit for bit."""
    mods = _eetq_linears(eetq_model)
    assert len(mods) == 2 * 7 and not any("lm_head" in n for n in mods)     # q k v o gate up down per layer
    ref = dict(fp16_model.named_modules())
    for name, mod in mods.items():
        assert isinstance(mod.weight, torch.nn.Parameter) and mod.weight.dtype == torch.int8 and mod.weight.is_cuda
        assert isinstance(mod.weight_scales, torch.nn.Parameter) and mod.weight_scales.dtype == torch.float16
        w_kn = ref[name].weight.detach().t().contiguous().numpy()
        q, s = oracle.quantize(w_kn)
        assert tuple(mod.weight.shape) == w_kn.shape
        assert np.array_equal(mod.weight.detach().cpu().numpy(), oracle.gfx950_pack(q)), name
        assert mod.weight_scales.detach().cpu().numpy().tobytes() == s.tobytes(), name


@pytest.mark.parametrize("rows", [1, 7, 40])
def test_their_linear_forward_vs_oracle_and_fp16_linear(oracle, eetq_model, fp16_model, rows):
    mods = _eetq_linears(eetq_model)
    ref = dict(fp16_model.named_modules())
    for name in ("model.layers.0.self_attn.q_proj", "model.layers.1.mlp.down_proj", "model.layers.0.mlp.gate_proj"):
        mod, lin = mods[name], ref[name]
        K = lin.in_features
        torch.manual_seed(rows)
        x = torch.rand(1, rows, K, dtype=torch.float16)
        with torch.no_grad():
            y = mod(x.to(DEV)).cpu().numpy().astype(np.float32)
            y_lin = lin(x).numpy().astype(np.float32)
        q, s = oracle.quantize(lin.weight.detach().t().contiguous().numpy())
        want = oracle.w8a16_gemm(x.view(rows, K).numpy(), q, s).astype(np.float32).reshape(y.shape)
        assert np.all(np.abs(y - want) <= 1e-3 * np.abs(want).max() + 2e-3 * np.abs(want)), name       # tier A
        assert np.abs(y - y_lin).max() <= 1e-2, (name, np.abs(y - y_lin).max())                          # reference's atol


def test_their_autograd_function_backward(oracle, eetq_model, fp16_model):
    """EetqLinearMMFunction.backward dequantises with an identity GEMM (M = K rows through w8_a16_gemm) and multiplies the
    incoming gradient by its transpose (python/eetq/modules/qlinear.py:80-94).  grad_input must be grad @ fp16(q.s)^T."""
    mods = _eetq_linears(eetq_model)
    name = "model.layers.0.mlp.up_proj"
    mod = mods[name]
    lin = dict(fp16_model.named_modules())[name]
    K, N = lin.in_features, lin.out_features
    torch.manual_seed(3)
    x = torch.rand(1, 9, K, dtype=torch.float16, device=DEV, requires_grad=True)
    y = mod(x)
    assert y.requires_grad
    g = torch.randn(1, 9, N, dtype=torch.float16, device=DEV) * 0.1
    y.backward(g)
    assert x.grad is not None and x.grad.shape == x.shape
    q, s = oracle.quantize(lin.weight.detach().t().contiguous().numpy())
    w_deq = oracle.dequant(q, s).astype(np.float32)                         # [K, N] fp16(q * s), exact
    want = g.view(9, N).cpu().numpy().astype(np.float32) @ w_deq.T         # [9, K]
    got = x.grad.view(9, K).cpu().numpy().astype(np.float32)
    assert np.all(np.abs(got - want) <= 2e-3 * np.abs(want).max() + 1e-2 * np.abs(want))
    assert mod.weight.grad is None                                          # int8 parameter: requires_grad False


def test_whole_model_logits_close_to_fp16(eetq_model, fp16_model):
    ids = torch.randint(0, 1000, (2, 24), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        a = eetq_model(ids.to(DEV)).logits.float().cpu()
        b = fp16_model(ids).logits.float()
    assert torch.isfinite(a).all()
    assert (a - b).norm() / b.norm() < 3e-2


def test_save_pretrained_writes_reference_bytes_and_reloads(oracle, eetq_model, fp16_model, tmp_path):
    """save_pretrained of the transformers-quantised model writes the REFERENCE's processed layout (sm80: what an NVIDIA
    GPU running EETQ would have written -- checked against the oracle's writer), and loading that directory as a
    pre-quantised checkpoint -- transformers assigns the tensors to the parameters, no load_state_dict -- gives the native
    bytes back and bit-identical logits."""
    from safetensors import safe_open
    d = str(tmp_path / "eetq_int8")
    eetq_model.save_pretrained(d)
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["quantization_config"]["quant_method"] == "eetq"
    ref = dict(fp16_model.named_modules())
    seen = 0
    for fn in os.listdir(d):
        if not fn.endswith(".safetensors"):
            continue
        with safe_open(os.path.join(d, fn), "pt") as f:
            keys = set(f.keys())
            for k in keys:
                if k.endswith(".weight") and (k + "_scales") in keys:
                    q, s = oracle.quantize(ref[k[:-len(".weight")]].weight.detach().t().contiguous().numpy())
                    assert np.array_equal(f.get_tensor(k).numpy(), oracle.sm80_pack(q)), k
                    assert f.get_tensor(k + "_scales").numpy().tobytes() == s.tobytes()
                    seen += 1
    assert seen == 14
    again = transformers.AutoModelForCausalLM.from_pretrained(d, device_map=DEV, dtype=torch.float16).eval()
    a, b = _eetq_linears(eetq_model), _eetq_linears(again)
    assert set(a) == set(b) and len(b) == 14
    for n in a:
        assert torch.equal(a[n].weight.data, b[n].weight.data), n          # gfx950 bytes in memory on both
        assert torch.equal(a[n].weight_scales.data, b[n].weight_scales.data)
    ids = torch.randint(0, 1000, (1, 16), generator=torch.Generator().manual_seed(6)).to(DEV)
    with torch.no_grad():
        assert torch.equal(eetq_model(ids).logits, again(ids).logits)


def test_layout_tag_is_honoured_and_written_through_transformers(oracle, eetq_model, tmp_path):
    """Round-4 ADVICE (medium): the transformers loader must take the source layout from the checkpoint's own tag, like
    models.from_quantized does -- a directory rewritten by convert_checkpoint(dst="gfx950") carries
    quantization_config["layout"] = "gfx950" and must NOT be re-encoded sm80 -> gfx950 a second time; and save_pretrained under
    a non-sm80 wire layout must write the tag (transformers' EetqConfig drops unknown keys)."""
    from eetq_amd.checkpoint import checkpoint_layout, convert_checkpoint, wire_layout
    d_sm80, d_native, d_saved = str(tmp_path / "sm80"), str(tmp_path / "native"), str(tmp_path / "saved_native")
    eetq_model.save_pretrained(d_sm80)
    assert "layout" not in json.load(open(os.path.join(d_sm80, "config.json")))["quantization_config"]   # the reference's config
    assert convert_checkpoint(d_sm80, d_native, dst="gfx950") == 14
    assert checkpoint_layout(json.load(open(os.path.join(d_native, "config.json")))) == "gfx950"
    ids = torch.randint(0, 1000, (1, 16), generator=torch.Generator().manual_seed(6)).to(DEV)
    with torch.no_grad():
        want = eetq_model(ids).logits
    a = _eetq_linears(eetq_model)
    for d in (d_native, d_sm80):
        again = transformers.AutoModelForCausalLM.from_pretrained(d, device_map=DEV, dtype=torch.float16).eval()
        b = _eetq_linears(again)
        for n in a:
            assert torch.equal(a[n].weight.data, b[n].weight.data), (d, n)    # gfx950 bytes in memory whatever the disk held
        with torch.no_grad():
            assert torch.equal(again(ids).logits, want), d
    # saving under a gfx950 wire layout: native bytes on disk AND the tag in config.json; reloads bit-identically
    with wire_layout("gfx950"):
        eetq_model.save_pretrained(d_saved)
    assert json.load(open(os.path.join(d_saved, "config.json")))["quantization_config"]["layout"] == "gfx950"
    from safetensors import safe_open
    with safe_open(os.path.join(d_saved, "model.safetensors"), "pt") as f:
        k = "model.layers.0.self_attn.q_proj.weight"
        assert torch.equal(f.get_tensor(k), a["model.layers.0.self_attn.q_proj"].weight.data.cpu())
    again = transformers.AutoModelForCausalLM.from_pretrained(d_saved, device_map=DEV, dtype=torch.float16).eval()
    with torch.no_grad():
        assert torch.equal(again(ids).logits, want)


def test_tgi_layer_call_pattern(oracle):
    """text-generation-inference's EETQ layer (server/text_generation_server/layers/eetq.py, `from EETQ import quant_weights,
    w8_a16_gemm`): fp16 weight -> `torch.t(w).contiguous().cpu()` -> `quant_weights(w, torch.int8, False)` -> `.cuda(device)`;
    forward = `w8_a16_gemm(input, weight, scale)` (+ bias).  The same lines on this repo's EETQ module, against the oracle."""
    from EETQ import quant_weights, w8_a16_gemm

    class EETQLinear(torch.nn.Module):          # the call sequence of TGI's layer, nothing else
        def __init__(self, weight, bias):
            super().__init__()
            device = weight.device
            if weight.dtype != torch.float16:
                weight = weight.to(dtype=torch.float16)
            weight = torch.t(weight).contiguous().cpu()
            weight, scale = quant_weights(weight, torch.int8, False)
            self.weight = weight.cuda(device)
            self.scale = scale.cuda(device)
            self.bias = bias.cuda(device) if bias is not None else None

        def forward(self, input):
            output = w8_a16_gemm(input, self.weight, self.scale)
            return output + self.bias if self.bias is not None else output

    torch.manual_seed(8)
    lin = torch.nn.Linear(512, 1024, bias=True, dtype=torch.float16)
    layer = EETQLinear(lin.weight.detach().to(DEV), lin.bias.detach().to(DEV))
    q, s = oracle.quantize(lin.weight.detach().t().contiguous().numpy())
    assert np.array_equal(layer.weight.cpu().numpy(), oracle.gfx950_pack(q)) and layer.scale.cpu().numpy().tobytes() == s.tobytes()
    for shape in ((1, 512), (3, 7, 512), (130, 512)):       # decode row, [batch, tokens, hidden], a prefill block
        x = torch.rand(*shape, dtype=torch.float16)
        y = layer(x.to(DEV))
        assert y.shape == shape[:-1] + (1024,)
        want = oracle.w8a16_gemm(x.reshape(-1, 512).numpy(), q, s).astype(np.float32) + lin.bias.detach().float().numpy()
        got = y.float().cpu().numpy().reshape(-1, 1024)
        assert np.all(np.abs(got - want) <= 1e-3 * np.abs(want).max() + 2e-3 * np.abs(want) + 1e-3)
        with torch.no_grad():
            assert np.abs(got - lin(x).float().numpy().reshape(-1, 1024)).max() <= 1e-2
