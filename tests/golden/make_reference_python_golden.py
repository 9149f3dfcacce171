#!/usr/bin/env python
"""Golden vectors from the reference's own PYTHON layer, run in the build container.

`/root/reference/python/eetq` imports unmodified once a module named `EETQ` is importable -- here that is this repo's
alias package (EETQ/__init__.py -> eetq_amd.ops), so the run below is the reference's Python code sitting on OUR operator
module.  Only CPU-side behaviour is recorded (module construction, buffer contracts, layer discovery / replacement,
the rotary cos|sin cache): nothing here launches a kernel.  Outputs (data only) are committed next to this script:
    ref_python_layer.json, ref_rotary_cache_d64_p96.npy
Run from the repo root:  python tests/golden/make_reference_python_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/python"
sys.path.insert(0, ROOT)   # provides `EETQ` (ours)
sys.path.insert(0, REF)    # provides `eetq` (the reference's package) ahead of this repo's alias package

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import EETQ  # noqa: E402
import eetq  # noqa: E402

assert EETQ.__file__.startswith(ROOT) and eetq.__file__.startswith(REF), (EETQ.__file__, eetq.__file__)
from eetq.modules.llama_modules import EETRotaryEmbedding  # noqa: E402
from eetq.modules.qlinear import EetqLinear, W8A16Linear  # noqa: E402
from eetq.utils import eet_quantize  # noqa: E402
from eetq.utils.base import find_layers, get_named_linears, set_op_by_name  # noqa: E402


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


out = {"generated_by": "tests/golden/make_reference_python_golden.py (reference python/eetq run on this repo's EETQ module)"}
rope = EETRotaryEmbedding(64, max_position_embeddings=96, base=10000)
np.save(os.path.join(HERE, "ref_rotary_cache_d64_p96.npy"), rope.cos_sin_cache.numpy())
out["rotary_cache"] = {"dim": 64, "max_position_embeddings": 96, "base": 10000, "shape": list(rope.cos_sin_cache.shape),
                       "dtype": str(rope.cos_sin_cache.dtype)}
out["w8a16linear_bias"] = sd_contract(W8A16Linear(64, 32, bias=True, dev="cpu"))
out["w8a16linear_nobias"] = sd_contract(W8A16Linear(64, 32, bias=False, dev="cpu"))
lin = nn.Linear(64, 32, bias=True).half()
init = W8A16Linear.from_torch(lin, init_only=True)
out["from_torch_init_only"] = {"in_features": init.in_features, "out_features": init.out_features,
                               "state": sd_contract(init), "qweight_all_zero": bool(init.qweight.abs().sum() == 0)}
e = EetqLinear(64, 32, bias=False, device="cpu")
out["eetqlinear_before_register_scale"] = sd_contract(e)
e.register_scale("cpu")
out["eetqlinear_after_register_scale"] = sd_contract(e)
m = toy()
out["find_layers_default"] = list(find_layers(m))
out["find_layers_no_exclude"] = list(find_layers(m, exclude=[]))
out["get_named_linears"] = list(get_named_linears(m))
set_op_by_name(m, "head.0", nn.Identity())
out["after_set_op_by_name_head0"] = [[n, type(s).__name__] for n, s in m.named_modules() if n.startswith("head.")]
m = toy().half()
eet_quantize(m, init_only=True)
out["eet_quantize_init_only_types"] = [[n, type(s).__name__] for n, s in m.named_modules()
                                       if isinstance(s, (nn.Linear, W8A16Linear))]
from eetq.modules.qlinear import quantize_and_preprocess_weights  # noqa: E402


def raised(fn):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        return [type(e).__name__, str(e)[:80]]
    return None


out["errors"] = {
    "quantize_and_preprocess_fp32": raised(lambda: quantize_and_preprocess_weights(torch.zeros(4, 4, dtype=torch.float32))),
    "quantize_and_preprocess_int8_without_scales": raised(
        lambda: quantize_and_preprocess_weights(torch.zeros(4, 4, dtype=torch.int8))),
    "eet_quantize_fp32_model": raised(lambda: eet_quantize(nn.Sequential(nn.Linear(4, 4)), init_only=True)),
}
with open(os.path.join(HERE, "ref_python_layer.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:1500])
