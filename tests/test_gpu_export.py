"""``from eetq import AutoEETQForCausalLM`` -- the reference's README recipe (README.md:55-68): from_pretrained -> quantize(dir)
writes a checkpoint with the reference's keys and BYTES (checked against the oracle's sm80 writer), and from_quantized (a stub
in the reference) loads it -- or an "NVIDIA-written" one built from the oracle's bytes -- back into W8A16Linear modules that
reproduce the in-memory quantised model bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
transformers = pytest.importorskip("transformers")


@pytest.fixture(scope="module")
def fp16_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tiny_llama_fp16_export"))
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=1000, max_position_embeddings=256)
    torch.manual_seed(21)
    transformers.LlamaForCausalLM(cfg).half().eval().save_pretrained(d)
    return d


def test_readme_recipe_quantize_save_reload(oracle, fp16_dir, tmp_path):
    from safetensors import safe_open
    from eetq import AutoEETQForCausalLM
    from eetq.modules.qlinear import W8A16Linear
    with pytest.raises(EnvironmentError):
        AutoEETQForCausalLM()
    model = AutoEETQForCausalLM.from_pretrained(fp16_dir, device_map=DEV)
    assert model.is_quantized is False and model.model_type == "llama"
    ref = transformers.AutoModelForCausalLM.from_pretrained(fp16_dir, dtype=torch.float16).eval()
    out = str(tmp_path / "quant")
    with pytest.raises(ValueError):
        model.quantize(out, tp=2)
    model.quantize(out)
    assert model.is_quantized
    cfg = json.load(open(os.path.join(out, "config.json")))
    assert cfg["quantization_config"] == {"quant_method": "eetq", "zero_point": False, "bits": 8}       # _config.py:80-85
    mods = dict(ref.named_modules())
    seen = 0
    for fn in os.listdir(out):
        if fn.endswith(".safetensors"):
            with safe_open(os.path.join(out, fn), "pt") as f:
                keys = set(f.keys())
                assert not any(k.startswith("lm_head") and k.endswith("qweight") for k in keys)           # lm_head stays fp16
                for k in keys:
                    if k.endswith(".qweight"):
                        name = k[:-len(".qweight")]
                        q, s = oracle.quantize(mods[name].weight.detach().t().contiguous().numpy())
                        assert np.array_equal(f.get_tensor(k).numpy(), oracle.sm80_pack(q)), k            # the reference's bytes
                        assert f.get_tensor(name + ".weight_scales").numpy().tobytes() == s.tobytes()
                        seen += 1
    assert seen == 14
    ids = torch.randint(0, 1000, (2, 12), generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        want = model(ids).logits
    again = AutoEETQForCausalLM.from_quantized(out)
    assert again.is_quantized
    lin = [m for m in again.model.modules() if isinstance(m, W8A16Linear)]
    assert len(lin) == 14
    with torch.no_grad():
        got = again(ids).logits
    assert torch.equal(got, want)
    g = again.generate(ids[:1], max_new_tokens=4, do_sample=False)
    assert g.shape == (1, 16)
    # quality: close to the fp16 model
    with torch.no_grad():
        b = ref(ids.cpu()).logits.float()
    assert (want.float().cpu() - b).norm() / b.norm() < 3e-2


def test_unsupported_model_type_is_refused(tmp_path):
    from eetq import AutoEETQForCausalLM
    d = str(tmp_path / "gpt2ish")
    cfg = transformers.GPT2Config(n_embd=64, n_layer=1, n_head=2, vocab_size=50)
    transformers.GPT2LMHeadModel(cfg).save_pretrained(d)
    with pytest.raises(TypeError, match="isn't supported yet"):
        AutoEETQForCausalLM.from_pretrained(d)
