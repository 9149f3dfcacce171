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
    if not torch.cuda.is_available():
        # int4 is a valid request through the compiled module (it still needs the GPU); the ctypes twin refuses it
        with pytest.raises(RuntimeError, match="no HIP device" if ops.BOUNDARY == "ext" else "int4"):
            ops.quant_weights(w, torch.quint4x2)
    with pytest.raises(RuntimeError, match="FP16 or FP32"):
        ops.quant_weights(w.to(torch.bfloat16), torch.int8)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.quant_weights(torch.zeros(64, 128, dtype=torch.float16)[:, ::2], torch.int8)
    with pytest.raises(RuntimeError, match="empty"):
        ops.quant_weights(torch.zeros(0, 64, dtype=torch.float16), torch.int8)
    with pytest.raises(RuntimeError, match="dim"):
        ops.quant_weights(torch.zeros(64, dtype=torch.float16), torch.int8)
    with pytest.raises(RuntimeError, match="dim"):
        ops.quant_weights(torch.zeros(2, 2, 64, 64, dtype=torch.float16), torch.int8)
    if not torch.cuda.is_available():
        # a 3-D expert stack [E, K, N] is a valid request (every expert is quantised): it reaches the device check
        with pytest.raises(RuntimeError, match="no HIP device"):
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


def test_eet_accelerator_rewrites_llama_attention_on_cpu():
    """Structure of eet_accelerator(fused_attn=True) (reference accelerator.py:22-51) without touching a GPU: every
    LlamaAttention becomes an EETLlamaAttention over ONE nn.Linear holding [q; k; v] rows; grouped-query geometry and the
    rotary cache follow the config; other rope types are refused rather than silently mis-rotated."""
    transformers = pytest.importorskip("transformers")
    from eetq.utils import eet_accelerator                      # README.md:80-81
    from eetq_amd.modules.llama_modules import EETLlamaAttention, EETRotaryEmbedding, _rope_base
    cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=32, max_position_embeddings=48)
    torch.manual_seed(0)
    model = transformers.LlamaForCausalLM(cfg).half()
    q0 = model.model.layers[0].self_attn.q_proj.weight.detach().clone()
    v0 = model.model.layers[0].self_attn.v_proj.weight.detach().clone()
    assert eet_accelerator(model, quantize=False, fused_attn=True) is model
    attn = model.model.layers[0].self_attn
    assert isinstance(attn, EETLlamaAttention) and isinstance(model.model.layers[1].self_attn, EETLlamaAttention)
    assert attn.num_heads == 4 and attn.num_key_value_heads == 2 and attn.head_dim == 16 and attn.layer_idx == 0
    assert model.model.layers[1].self_attn.layer_idx == 1
    w = attn.qkv_proj.weight
    assert w.shape == ((4 + 2 * 2) * 16, 64) and torch.equal(w[:64], q0) and torch.equal(w[96:], v0)
    rope = attn.rotary_emb
    assert isinstance(rope, EETRotaryEmbedding) and rope.cos_sin_cache.shape == (48, 16)
    assert rope.cos_sin_cache.dtype == torch.float16 and rope.base == _rope_base(cfg)
    # cache layout of the reference (llama_modules.py:33-46): cos half then sin half, position-major
    t = torch.arange(48, dtype=torch.float32)
    inv = 1.0 / (rope.base ** (torch.arange(0, 16, 2, dtype=torch.float32) / 16))
    assert torch.equal(rope.cos_sin_cache[:, :8], torch.outer(t, inv).cos().half())
    assert torch.equal(rope.cos_sin_cache[:, 8:], torch.outer(t, inv).sin().half())

    class Scaled:
        rope_parameters = {"rope_type": "llama3", "rope_theta": 1e4}
    with pytest.raises(NotImplementedError):
        _rope_base(Scaled())


def test_replace_with_eet_qlinear_rejects_unmapped_models():
    from eetq_amd.utils.accelerator import replace_with_eet_qlinear
    with pytest.raises(ValueError):
        replace_with_eet_qlinear(nn.Sequential(), target_model="opt")


def _ref_python_golden():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return json.load(open(os.path.join(here, "ref_python_layer.json"))), here


def test_python_layer_matches_the_reference_python_layer():
    """tests/golden/ref_python_layer.json was produced by the REFERENCE's python/eetq package, imported unmodified in the
    build container on top of this repo's `EETQ` module (tests/golden/make_reference_python_golden.py).  The mirrors here
    must reproduce its buffer contracts, layer discovery and replacement results."""
    from eetq_amd.modules.qlinear import EetqLinear, W8A16Linear
    from eetq_amd.utils.quantizer import eet_quantize, find_layers, get_named_linears, set_op_by_name
    g, _ = _ref_python_golden()

    def sd_contract(m):
        return {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}

    def toy():
        class Sub(nn.Linear):
            pass
        model = nn.Sequential()
        model.add_module("blocks", nn.ModuleList([nn.Sequential(nn.Linear(8, 8), nn.ReLU()), nn.Sequential(Sub(8, 8))]))
        model.add_module("head", nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 4)))
        model.add_module("lm_head", nn.Linear(8, 8))
        return model

    assert sd_contract(W8A16Linear(64, 32, bias=True, dev="cpu")) == g["w8a16linear_bias"]
    assert sd_contract(W8A16Linear(64, 32, bias=False, dev="cpu")) == g["w8a16linear_nobias"]
    init = W8A16Linear.from_torch(nn.Linear(64, 32, bias=True).half(), init_only=True)
    assert {"in_features": init.in_features, "out_features": init.out_features, "state": sd_contract(init),
            "qweight_all_zero": bool(init.qweight.abs().sum() == 0)} == g["from_torch_init_only"]
    e = EetqLinear(64, 32, bias=False, device="cpu")
    assert sd_contract(e) == g["eetqlinear_before_register_scale"]
    e.register_scale("cpu")
    assert sd_contract(e) == g["eetqlinear_after_register_scale"]
    m = toy()
    assert list(find_layers(m)) == g["find_layers_default"]
    assert list(find_layers(m, exclude=[])) == g["find_layers_no_exclude"]
    assert list(get_named_linears(m)) == g["get_named_linears"]
    set_op_by_name(m, "head.0", nn.Identity())
    ours = {n: type(s).__name__ for n, s in m.named_modules() if n.startswith("head.")}
    assert ours == dict(g["after_set_op_by_name_head0"])
    # deliberate difference: the reference's delattr + setattr moves the replaced child to the END of its container
    # (its recorded order is head.1, head.0 -- it reorders an nn.Sequential); ours keeps the position
    assert [n for n, _ in m.named_modules() if n.startswith("head.")] == ["head.0", "head.1"]
    assert [p[0] for p in g["after_set_op_by_name_head0"]] == ["head.1", "head.0"]
    m = toy().half()
    eet_quantize(m, init_only=True)
    assert [[n, type(s).__name__] for n, s in m.named_modules()
            if isinstance(s, (nn.Linear, W8A16Linear))] == g["eet_quantize_init_only_types"]


def test_rotary_cache_is_bit_identical_to_the_reference_module():
    """cos|sin cache of the reference's EETRotaryEmbedding(64, 96, 10000), generated by running the reference module."""
    import os
    import numpy as np
    from eetq_amd.modules.llama_modules import EETRotaryEmbedding
    g, here = _ref_python_golden()
    ref = np.load(os.path.join(here, "ref_rotary_cache_d64_p96.npy"))
    rc = g["rotary_cache"]
    ours = EETRotaryEmbedding(rc["dim"], max_position_embeddings=rc["max_position_embeddings"], base=rc["base"])
    assert list(ours.cos_sin_cache.shape) == rc["shape"] and str(ours.cos_sin_cache.dtype) == rc["dtype"]
    assert ours.cos_sin_cache.numpy().tobytes() == ref.tobytes()


def test_error_behaviour_matches_the_reference_python_layer():
    """Exception types and messages recorded from the reference's python layer (same fixture)."""
    from eetq_amd.modules.qlinear import quantize_and_preprocess_weights
    from eetq_amd.utils.quantizer import eet_quantize
    g, _ = _ref_python_golden()

    def raised(fn):
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            return [type(e).__name__, str(e)[:80]]
        return None

    ours = {
        "quantize_and_preprocess_fp32": raised(lambda: quantize_and_preprocess_weights(torch.zeros(4, 4, dtype=torch.float32))),
        "quantize_and_preprocess_int8_without_scales": raised(
            lambda: quantize_and_preprocess_weights(torch.zeros(4, 4, dtype=torch.int8))),
        "eet_quantize_fp32_model": raised(lambda: eet_quantize(nn.Sequential(nn.Linear(4, 4)), init_only=True)),
    }
    for key, (etype, msg) in g["errors"].items():
        assert ours[key] is not None and ours[key][0] == etype, (key, ours[key], etype)
        if msg:
            assert ours[key][1] == msg, (key, ours[key][1], msg)


def test_glu8_fuse_layout_matches_packing_the_interleaved_matrix():
    """fuse_w8a16_linears(glu8=True) permutes the PROCESSED bytes of gate and up; the result must be the processed form of
    the raw [K, 2I] matrix whose columns are interleaved in groups of 8 gate + 8 up (oracle packer), scales likewise."""
    import numpy as np
    import oracle
    from eetq_amd.utils.fuse import _glu8_interleave_columns, _glu8_interleave_tiles
    rng = np.random.default_rng(0)
    for K, I in ((64, 16), (192, 48), (256, 160)):
        G = rng.integers(-128, 128, (K, I), dtype=np.int8)
        U = rng.integers(-128, 128, (K, I), dtype=np.int8)
        fused = _glu8_interleave_tiles(torch.from_numpy(oracle.gfx950_pack(G)), torch.from_numpy(oracle.gfx950_pack(U)), K)
        raw = _glu8_interleave_columns(torch.from_numpy(G), torch.from_numpy(U)).numpy()
        for t in range(I // 8):
            assert np.array_equal(raw[:, 16 * t: 16 * t + 8], G[:, 8 * t: 8 * t + 8])
            assert np.array_equal(raw[:, 16 * t + 8: 16 * t + 16], U[:, 8 * t: 8 * t + 8])
        assert np.array_equal(fused.reshape(K, 2 * I).numpy(), oracle.gfx950_pack(np.ascontiguousarray(raw)))
    s = _glu8_interleave_columns(torch.arange(32.), 100 + torch.arange(32.))
    assert s[:16].tolist() == list(range(8)) + list(range(100, 108))


def test_quantised_modules_pickle_whole():
    """torch.save(model) / torch.load of whole modules (the reference's examples do this: examples/layers/test_qlinear.py:44,
    llama_transformers_example.py:49) -- the layout hooks are module-level callables, not closures."""
    import io
    from eetq_amd.modules.qlinear import EetqLinear, W8A16Linear
    for mod in (W8A16Linear(64, 128, bias=True, dev="cpu"), EetqLinear(64, 128, bias=False, device="cpu")):
        buf = io.BytesIO()
        torch.save(mod, buf)
        buf.seek(0)
        back = torch.load(buf, weights_only=False)
        assert type(back) is type(mod) and back.in_features == 64 and back.out_features == 128
        assert len(back._state_dict_hooks) == 1 and len(back._load_state_dict_pre_hooks) == 1   # hooks travel with it
        name = "qweight" if isinstance(mod, W8A16Linear) else "weight"
        assert torch.equal(getattr(back, name), getattr(mod, name))


def test_state_dict_layout_tag_is_in_band_and_wins_on_load():
    """The save hook records the layout of the int8 bytes in the state dict's _metadata; the load hook trusts the tag
    before the module pin and the process-wide setting (host logic only: shapes whose bytes pass through unchanged)."""
    from eetq_amd.checkpoint import _LoadHook, _SaveHook
    from eetq_amd.modules.qlinear import W8A16Linear
    mod = W8A16Linear(48, 16, bias=False, dev="cpu")     # K % 64 != 0: the reference has no layout -> bytes pass through
    mod.qweight.copy_(torch.arange(48 * 16).reshape(48, 16).to(torch.int8))
    sd = mod.state_dict()
    assert sd._metadata[""]["eetq_layout"] == "gfx950" and sd._metadata[""]["version"] == 1
    other = W8A16Linear(48, 16, bias=False, dev="cpu")
    other.load_state_dict(sd)
    assert torch.equal(other.qweight, mod.qweight)
    # an unknown tag is an error, not a silent pass-through
    sd._metadata[""]["eetq_layout"] = "sm90"
    with pytest.raises(ValueError):
        W8A16Linear(48, 16, bias=False, dev="cpu").load_state_dict(sd)
    assert isinstance(mod._state_dict_hooks[next(iter(mod._state_dict_hooks))], _SaveHook)
    assert any(isinstance(getattr(h, "hook", h), _LoadHook) or isinstance(h, _LoadHook)
               for h in mod._load_state_dict_pre_hooks.values())


def test_eetq_py_shim_serves_both_boundaries(monkeypatch):
    """EETQ.py (executed only when EETQ.cpython-*.so is missing: extension modules win the import) hands out the compiled
    module when it can be built and the ctypes binding of the same C ABI otherwise."""
    import importlib.util
    import os
    import sys
    from eetq_amd import _ext
    names = ["w8_a16_gemm", "w8_a16_gemm_", "preprocess_weights", "quant_weights", "rotary_embedding_neox",
             "layernorm_forward"]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "EETQ.py")

    def run_shim(alias):
        spec = importlib.util.spec_from_file_location(alias, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[alias] = mod
        try:
            spec.loader.exec_module(mod)
            return sys.modules[alias]
        finally:
            sys.modules.pop(alias, None)

    def no_compiler(*a, **k):
        raise RuntimeError("no C++ compiler")
    monkeypatch.setattr(_ext, "build", no_compiler)
    fallback = run_shim("EETQ_shim_fallback")
    assert all(callable(getattr(fallback, n)) for n in names) and not hasattr(fallback, "__eetq_amd_version__")
    monkeypatch.undo()
    compiled = run_shim("EETQ_shim_compiled")
    assert all(callable(getattr(compiled, n)) for n in names) and hasattr(compiled, "__eetq_amd_version__")


def test_transformers_hook_redirects_the_kernel_lookup():
    """eetq_amd.utils.hf.use_with_transformers(): transformers' replace_with_eetq_linear fetches its kernel module with
    get_kernel("kernels-community/quantization-eetq") (needs the `kernels` package and a network); after the hook that
    lookup, the module-level handle and the quantizer's environment check resolve to this repo's EETQ module, other kernels
    are still looked up the stock way, and every EetqLinear carries the wire-format save hook.  No GPU needed: the modules
    are created on the meta device, as transformers does."""
    transformers = pytest.importorskip("transformers")
    import torch
    from eetq_amd.utils.hf import HUB_KERNEL_NAME, use_with_transformers
    mod = use_with_transformers()
    assert use_with_transformers() is mod                       # idempotent
    import transformers.integrations.eetq as hf_eetq
    import transformers.integrations.hub_kernels as hub
    import transformers.quantizers.quantizer_eetq as hf_q
    assert hub.get_kernel(HUB_KERNEL_NAME, version=1) is mod
    assert hf_q.is_kernels_available() is True
    for name in ("quant_weights", "w8_a16_gemm"):               # the two functions transformers calls
        assert callable(getattr(mod, name))
    with pytest.raises(Exception):
        hub.get_kernel("kernels-community/some-other-kernel")   # still the stock path (no `kernels` package here)
    cfg = transformers.LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                                   num_key_value_heads=2, vocab_size=100)
    with torch.device("meta"):
        model = transformers.LlamaForCausalLM(cfg)
    model = hf_eetq.replace_with_eetq_linear(model, modules_to_not_convert=["lm_head"])
    assert hf_eetq.eetq_kernels_hub is mod
    lins = [m for m in model.modules() if isinstance(m, hf_eetq.EetqLinear)]
    assert len(lins) == 7 and all(getattr(m, "_eetq_layout_hooks", False) for m in lins)
    assert isinstance(model.lm_head, torch.nn.Linear)


def test_eetq_package_exports_the_reference_surface():
    """`from eetq import AutoEETQForCausalLM` (python/eetq/__init__.py:1-2) resolves lazily; `import eetq` itself does not pull
    transformers in.  The class refuses direct construction like the reference's (auto.py:20-23)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import eetq; assert 'transformers' not in sys.modules; "
            "from eetq import AutoEETQForCausalLM, eet_quantize, eet_accelerator, W8A16Linear; "
            "from eetq.models import AutoEETQForCausalLM as A2, BaseEETQForCausalLM; assert A2 is AutoEETQForCausalLM; print('ok')")
    from conftest import ROOT
    out = subprocess.run([sys.executable, "-c", code % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    from eetq import AutoEETQForCausalLM
    with pytest.raises(EnvironmentError):
        AutoEETQForCausalLM()
    for name in ("from_pretrained", "from_quantized"):
        assert callable(getattr(AutoEETQForCausalLM, name))
