"""Time of quant_weights (column max + quantise + pack, GPU tensors in and out) per layer, and the kernel split from a
short rocprofv3-free event loop.  usage: python tools/quant_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from eetq_amd import ops

dev = "cuda:0"
for K, N, dt in [(4096, 4096, torch.float16), (4096, 11008, torch.float16), (13824, 5120, torch.float16), (4096, 4096, torch.float32)]:
    ws = [((torch.rand(K, N, device=dev) * 2 - 1) / K ** 0.5).to(dt) for _ in range(6)]   # > 256 MiB of inputs in rotation
    for _ in range(3):
        ops.quant_weights(ws[0], torch.int8, False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    a.record()
    for i in range(reps):
        ops.quant_weights(ws[i % len(ws)], torch.int8, False)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    # the same calls as a HIP graph: device time without the host's share (allocations, binding, launch calls)
    graph_us = float("nan")
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.quant_weights(ws[0], torch.int8, False)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = [ops.quant_weights(ws[i % len(ws)], torch.int8, False) for i in range(12)]
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        graph_us = (time.perf_counter() - t0) * 1e6 / (5 * 12)
        del g, keep
    except Exception as e:  # noqa: BLE001
        print("# graph form not available: %s" % str(e)[:100], file=sys.stderr)
    moved = K * N * ws[0].element_size() * 2 + K * N          # two reads of w + one write of the packed bytes
    print(json.dumps({"K": K, "N": N, "dtype": str(dt), "quant_weights_us": round(us, 1), "graph_replayed_us": round(graph_us, 1),
                      "bytes_moved_GBps": round(moved / us / 1e3), "min_traffic_GBps": round((K * N * (ws[0].element_size() + 1)) / us / 1e3)}))
    del ws
