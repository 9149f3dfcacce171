"""The reference's layer check (examples/layers/test_qlinear.py:20-36) on this library: seed 1, nn.Linear(1024 -> 4096, fp16, no
bias), x = rand(128, 1024); W8A16Linear.from_torch vs the fp16 layer, atol = 1e-2; then the state-dict round trip the reference
script does with torch.save (here into a temporary directory).  Prints True / False like the reference; exit status 1 on False.
usage: python examples/layers/test_qlinear.py"""
import os
import random
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eetq.modules.qlinear import W8A16Linear  # noqa: E402  (the reference's import path)


def set_random_seed(seed):
    random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


if __name__ == "__main__":
    set_random_seed(1)
    M, N, K = 128, 4096, 1024
    torch_linear = nn.Linear(K, N, bias=False, dtype=torch.float16).cuda()   # (the layer's device decides where the module lives)
    eet_linear = W8A16Linear.from_torch(torch_linear, scales=None, init_only=False)
    x = torch.rand(M, K, dtype=torch.float16).cuda()
    output = eet_linear(x)
    output_torch = torch_linear(x)
    ok = torch.allclose(output, output_torch, atol=1e-2)
    print("eet out: ", output)
    print("out torch: ", output_torch)
    print(ok, "max abs err %.3e" % (output - output_torch).abs().max().item())
    with tempfile.TemporaryDirectory() as d:
        torch.save(eet_linear.state_dict(), os.path.join(d, "eet_linear.pt"))     # the reference's bytes (sm80 layout) on disk
        torch.save(eet_linear, os.path.join(d, "eet_linear_model.pt"))
        fresh = W8A16Linear.from_torch(torch_linear, init_only=True)
        fresh.load_state_dict(torch.load(os.path.join(d, "eet_linear.pt")))
        whole = torch.load(os.path.join(d, "eet_linear_model.pt"), weights_only=False)
        same = torch.equal(fresh(x), output) and torch.equal(whole(x), output)
    print("state dict / whole-module round trip bit-identical:", same)
    sys.exit(0 if ok and same else 1)
