"""Generates the committed fixtures under tests/golden/.  Run from the repo root:  python tests/golden/make_golden.py

Provenance (stated per file in MANIFEST.json):
  * ``linear_*``  : outputs of CPU ``torch.nn.Linear`` in float16 -- the reference forward BASELINE.json names as
    configs[0] and the comparator of the reference's own examples/layers/test_qlinear.py:20-36 (seed 1 recipes).
    Inputs are regenerated from the seed; a CRC of the inputs is stored so RNG drift is detected, and the
    small case stores inputs in full.
  * ``quant_*``   : inputs (hand-built edge cases + seeded random) and the outputs of oracle/ (C restatement of
    cutlass_preprocessors.cc).  The reference quantiser itself cannot be compiled in this image (needs
    CUTLASS + CUDA headers), so these are ORACLE-GENERATED pins, not reference-generated vectors.
"""
import json
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def linear_case(name, M, K, N, store_inputs, row_step=1):
    # recipe of examples/layers/test_qlinear.py:20-26: seed 1; nn.Linear(K, N, bias=False, fp16); x = rand(M, K)
    torch.manual_seed(1)
    lin = torch.nn.Linear(K, N, bias=False, dtype=torch.float16)
    x = torch.rand(M, K, dtype=torch.float16)
    with torch.no_grad():
        y = lin(x)
    w = lin.weight.detach().numpy()
    out = {"y": y.numpy()[::row_step]}  # every row_step-th token row keeps the fixture small
    if store_inputs:
        out["w"] = w
        out["x"] = x.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return {"file": name + ".npz", "M": M, "K": K, "N": N, "seed": 1, "row_step": row_step, "w_crc32": crc(w), "x_crc32": crc(x.numpy()),
            "provenance": "torch %s CPU nn.Linear float16 forward" % torch.__version__}


def quant_edge_inputs():
    rng = np.random.default_rng(20240307)
    K, N = 128, 64
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float16)
    w[:, 0] = 0                      # all-zero column -> scale 0, q = 127 (0/0 = NaN -> min(127, NaN))
    w[:, 1] = 0; w[5, 1] = -1.0      # single negative spike -> -128 exactly, everything else 0
    w[:, 2] = 0; w[7, 2] = 1.0       # single positive spike -> +128 clipped to 127
    # exact ties: scale = 1 (amax = 128) and values k + 0.5 -> half away from zero
    w[:, 3] = 0; w[0, 3] = 128.0
    ties = np.array([0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 126.5, -126.5, 127.5, -127.5], np.float16)
    w[1:1 + len(ties), 3] = ties
    w[:, 4] = np.float16(6.0e-8)     # subnormal fp16 column (scale is an fp32 subnormal-free tiny number)
    w[:, 5] = 0; w[3, 5] = np.float16(65504.0); w[4, 5] = np.float16(-65504.0)  # fp16 max
    w[:, 6] = 0; w[2, 6] = np.float16(np.inf); w[9, 6] = 1.0                  # inf column: scale inf
    w[:, 7] = (rng.standard_normal(K) * 0.05).astype(np.float16); w[11, 7] = np.float16(np.nan)  # NaN element
    return w


def quant_case(name, w):
    q, s = oracle.quantize(w)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), w=w, q=q, s=s,
                        gfx950=oracle.gfx950_pack(q), sm80=oracle.sm80_pack(q))
    return {"file": name + ".npz", "K": int(w.shape[0]), "N": int(w.shape[1]), "dtype": str(w.dtype),
            "provenance": "oracle/eetq_oracle.c (restatement; NOT a reference build)"}


def main():
    manifest = {"linear": [], "quant": []}
    manifest["linear"].append(linear_case("linear_small_m4_k256_n128", 4, 256, 128, True))
    manifest["linear"].append(linear_case("linear_qlinear_recipe_m128_k1024_n4096", 128, 1024, 4096, False, row_step=8))
    manifest["linear"].append(linear_case("linear_config0_m1_k4096_n4096", 1, 4096, 4096, False))
    manifest["quant"].append(quant_case("quant_edge_f16_k128_n64", quant_edge_inputs()))
    rng = np.random.default_rng(7)
    manifest["quant"].append(quant_case("quant_rand_f16_k192_n256", (rng.standard_normal((192, 256)) * 0.02).astype(np.float16)))
    manifest["quant"].append(quant_case("quant_rand_f32_k64_n64", (rng.standard_normal((64, 64)) * 3.0).astype(np.float32)))
    # sections other scripts / hands own (reference_python_layer: make_reference_python_golden.py; machine_code: the hazard
    # fixture of tools/check_store_hazard.py) are carried over
    path = os.path.join(HERE, "MANIFEST.json")
    if os.path.exists(path):
        with open(path) as f:
            for key, value in json.load(f).items():
                manifest.setdefault(key, value)
    with open(path, "w") as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
