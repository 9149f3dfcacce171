"""quant_weights per call on the three routes: native layout only (quant_pack_kernel), native + row-major copy and the sm80
wire layout (strip_quant_kernel).  usage: python tools/quant_paths.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eetq_amd import ops
dev = "cuda:0"
K = N = 4096
ws = [((torch.rand(K, N, device=dev) * 2 - 1) / K ** 0.5).half() for _ in range(10)]
def timed(fn):
    for _ in range(3): fn(ws[0])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(50): fn(ws[i % 10])
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) * 1e3 / 50, 1)
print(json.dumps({"K": K, "N": N, "dtype": "fp16",
                  "native_us": timed(lambda w: ops.quant_weights(w, torch.int8, False)),
                  "native_plus_row_major_us": timed(lambda w: ops.quant_weights(w, torch.int8, True)),
                  "sm80_layout_us": timed(lambda w: ops.quant_weights(w, torch.int8, False, layout="sm80"))}))
