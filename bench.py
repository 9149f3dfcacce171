#!/usr/bin/env python
"""Benchmark of the W8A16 hot path on MI355X.  Contract: python bench.py --gpus N --steps K --warmup W
(for N > 1 launched by torch.distributed.run, one rank per GPU); rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): w8a16 GEMV, M=1, N=K=4096.
A "step" is one w8_a16_gemm call (one pass of the decode hot path over one batch) on the next of NBUF distinct
(weight, scale) sets -- NBUF*16 MiB >= 512 MiB, so the weights come from HBM, not the 256 MB Infinity Cache.

  value      whole-job GB/s: algorithmic bytes (K*N + 2*M*K + 2*N + 2*M*N = 16 801 792 B/step) x steps x replicas / wall
             time of the timed region.  K steps are captured as HIP graphs of dependent launches -- as many graphs as it
             takes for their concatenation to visit every weight set equally (K = 20, NBUF = 40: two graphs) -- and the
             timed region replays them round-robin until it is at least --min-timed-ms long (default 50 ms) whatever K is,
             with barrier + synchronize on both sides and the MAX over ranks.  ms_per_step = region / (replays x K).
             Includes the ~1.5-1.9 us dependent-kernel boundary of every step.
  roofline   dominant kernel (gemv_kernel): algorithmic bytes / mean kernel duration, measured live with a HIP start/stop
             event pair attached to every dispatch (hipExtLaunchKernelGGL via eetq_prof_begin/_end) on the launch stream,
             over a pass through all weight sets.  `method_floor_us` is the same method on an EMPTY kernel of the GEMV's
             launch geometry (the method cannot read anything shorter); `read_only_floor` the same on a kernel that only
             loads the 16 MiB.  The rocprofv3 --kernel-trace --stats summary of this very command is committed by
             tools/profile_bench.sh under profiles/ (rNN_bench_kernel_stats.csv) together with the PMC traffic pass that
             `traffic` is read from (profiles/pmc_traffic.json: bytes per launch, method and date inside).
             roofline.gemm_m1024: the other half of the metric, fused dequant-GEMM at M=1024 (MFMA roofline).
  cpu_baseline  the oracle's scalar C port of the same GEMV on one host core (bounded sample); beside it
             (cpu_linear_fp16) the north star's CPU torch.nn.Linear fp16 forward, best over a sweep of thread counts.
  config5    BASELINE configs[4] on the same box (skip with --no-config5): Llama-2-13B shapes, prompt 1024 + 50 new tokens,
             one replica per GPU, whole-job tokens/s.  Reported beside the headline, never as `value`.
Multi-GPU: replicas only (model replicated, no data-path collective); rank 0 fans the activations out with one
broadcast, results are checked to be bit-identical across replicas.  scaling = "weak".
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA


def gemv_bytes(M, N, K):
    return K * N + 2 * M * K + 2 * N + 2 * M * N


def make_weight_sets(ops, nbuf, K, N, dev):
    """set 0 = nn.Linear default init, seed 1 (the recipe of examples/layers/test_qlinear.py); the others are
    U(+-1/sqrt(K)) drawn on the GPU.  All quantised by the HIP quantiser on the GPU."""
    sets = []
    torch.manual_seed(1)
    lin = torch.nn.Linear(K, N, bias=False, dtype=torch.float16)
    w0 = lin.weight.detach().t().contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    bound = 1.0 / (K ** 0.5)
    for i in range(nbuf):
        if i == 0:
            w = w0.to(dev)
        else:
            w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) * bound).half()
        processed, scales = ops.quant_weights(w, torch.int8, False)
        sets.append((processed, scales))
        del w
    return sets, w0, lin


def capture_graphs(fn, nsteps, nbuf):
    """Graphs of `nsteps` dependent launches each; graph g runs steps [g*nsteps, (g+1)*nsteps).  Enough graphs that their
    concatenation is a whole number of passes over the nbuf weight sets."""
    count = nbuf // math.gcd(nsteps, nbuf)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0, 3)  # warm the capture stream / lazy init outside capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graphs = []
    for gi in range(count):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn(gi * nsteps, nsteps)
        graphs.append(g)
    return graphs


def timed_replays(grp, graphs, nsteps, min_seconds):
    """Replays the graphs round-robin for >= min_seconds (a whole number of rounds); returns (seconds, replays)."""
    for g in graphs:  # one untimed replay each: graph upload
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / len(graphs), 1e-6)      # seconds per replay (incl. launch latency: an upper bound)
    rounds = max(1, int(math.ceil(min_seconds / est / len(graphs))))
    if grp.world_size > 1:  # every rank must time the same amount of work
        rounds = int(grp.max_over_ranks(float(rounds)))

    def run():
        for _ in range(rounds):
            for g in graphs:
                g.replay()
    seconds = grp.timed(run)
    if seconds < min_seconds:  # the estimate included host latency: top up once
        rounds = int(math.ceil(rounds * min_seconds / seconds * 1.1))
        if grp.world_size > 1:
            rounds = int(grp.max_over_ranks(float(rounds)))
        seconds = grp.timed(run)
    return seconds, rounds * len(graphs)


def dispatch_kernel_time(run, nlaunches):
    """Mean/median/min kernel duration (seconds) of `nlaunches` launches issued by run(): every launch carries a HIP
    start/stop event pair on its dispatch packet (eetq_prof_begin/_end -> hipExtLaunchKernelGGL), i.e. the kernel's own
    begin/end timestamps -- the quantity rocprofv3 --kernel-trace reports -- on the stream the kernel is launched on."""
    from eetq_amd import _lib
    L = _lib.lib()
    _lib.check(L.eetq_prof_begin(nlaunches))
    run()
    buf = (ctypes.c_float * nlaunches)()
    cnt = ctypes.c_int(0)
    _lib.check(L.eetq_prof_end(buf, nlaunches, ctypes.byref(cnt)))
    us = np.array(buf[:cnt.value], dtype=np.float64)
    assert cnt.value == nlaunches, (cnt.value, nlaunches)
    return float(us.mean()) * 1e-6, float(np.median(us)) * 1e-6, float(us.min()) * 1e-6


def cpu_gemv_baseline(oracle, x, q, s, budget_s=10.0):
    """Oracle port (scalar C, one core) of the same GEMV; bounded sample."""
    t_end = time.perf_counter() + budget_s
    oracle.w8a16_gemm_f32acc(x, q, s)  # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() < t_end or n < 3:
        oracle.w8a16_gemm_f32acc(x, q, s)
        n += 1
    dt = time.perf_counter() - t0
    return n, dt


def cpu_linear_sweep(lin, x, runs, thread_counts):
    """CPU nn.Linear fp16 forward (north-star baseline): median of `runs` per thread count; returns {threads: seconds}."""
    out = {}
    keep = torch.get_num_threads()
    try:
        with torch.no_grad():
            for t in thread_counts:
                torch.set_num_threads(t)
                for _ in range(2):
                    lin(x)
                ts = []
                for _ in range(runs):
                    t0 = time.perf_counter()
                    lin(x)
                    ts.append(time.perf_counter() - t0)
                out[t] = float(np.median(ts))
    finally:
        torch.set_num_threads(keep)
    return out


def load_traffic():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path))
    except Exception:
        return {}


def config5_leg(grp, prompt_len=1024, new_tokens=50):
    """BASELINE configs[4] on every replica: random-init Llama-2-13B shapes (no checkpoints offline), eet_accelerator with
    W8A16 everywhere, identical prompt fanned out from rank 0, greedy decode of 50 tokens through the HIP-graph decoder.
    Whole-job tokens/s = replicas x 50 / MAX over ranks of the end-to-end time (prefill + decode).  None when transformers
    is not importable."""
    try:
        import transformers
        from eetq_amd.utils import GraphDecoder, eet_accelerator
    except Exception as e:  # noqa: BLE001
        return {"skipped": "transformers / accelerator not importable: %s" % (str(e)[:80],)}
    dev = grp.device
    cfg = transformers.LlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                                   num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)
    torch.manual_seed(0)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        with torch.device(dev):
            model = transformers.LlamaForCausalLM(cfg).eval()
    finally:
        torch.set_default_dtype(old)
    eet_accelerator(model, quantize=True, fused_attn=True, fused_mlp=True, fused_norm=True, fused_residual=True)
    prompt = torch.randint(0, 32000, (1, prompt_len), generator=torch.Generator().manual_seed(1)).to(dev)
    grp.fan_out(prompt)
    holder = {}
    with torch.no_grad():
        dec = GraphDecoder(model, 1, prompt_len + new_tokens + 8)
        dec.generate(prompt[:, :64], 4)   # warm-up (allocations, kernel selection)
        torch.cuda.synchronize()

        def run():
            holder["out"] = dec.generate(prompt, new_tokens)
        secs = grp.timed(run)
        t_prefill = grp.timed(lambda: model(prompt))
    crcs = grp.gather_checksums(holder["out"][:, prompt_len:].to(torch.int32))
    res = {"workload": "Llama-2-13B shapes (random init), eet_accelerator W8A16, prompt %d + %d new tokens, batch 1 per replica, "
                       "HIP-graph greedy decode" % (prompt_len, new_tokens),
           "tokens_per_s": round(grp.world_size * new_tokens / secs, 2), "end_to_end_s": round(secs, 4),
           "prefill_s": round(t_prefill, 4), "replicas": grp.world_size, "replicas_identical_tokens": len(set(crcs)) == 1}
    del dec, model
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--nbuf", type=int, default=40, help="distinct weight sets rotated per step (x16 MiB)")
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="minimum length of every timed region")
    ap.add_argument("--gemm-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the BASELINE configs[4] leg (Llama-2-13B shapes, prompt 1024 + 50 new tokens, one replica per GPU)")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = ap.parse_args()

    from eetq_amd import _lib, ops
    from eetq_amd.utils.replicas import ReplicaGroup

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the W8A16 path has no CPU implementation)")
    grp = ReplicaGroup()
    if grp.world_size != args.gpus and grp.rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, grp.world_size), file=sys.stderr)
    dev = grp.device
    M, N, K = 1, 4096, 4096
    steps, warmup, nbuf = args.steps, args.warmup, args.nbuf
    min_s = args.min_timed_ms * 1e-3

    sets, w0_cpu, lin_cpu = make_weight_sets(ops, nbuf, K, N, dev)
    # identical activations on every replica: rank 0 draws them, RCCL broadcast fans them out
    torch.manual_seed(1)
    x = torch.rand(M, K, dtype=torch.float16).to(dev)
    grp.fan_out(x)
    outs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(8)]

    def gemv_steps(first, count):
        for i in range(first, first + count):
            w, s = sets[i % nbuf]
            ops.w8_a16_gemm_(x, w, s, outs[i % len(outs)], M, N, K)

    # ---- parity of this run (rank 0 checks against the oracle; all ranks must agree bit for bit) ----
    parity = {}
    y0 = torch.empty(M, N, dtype=torch.float16, device=dev)
    ops.w8_a16_gemm_(x, sets[0][0], sets[0][1], y0, M, N, K)
    torch.cuda.synchronize()
    crcs = grp.gather_checksums(y0)
    parity["replicas_bit_identical"] = len(set(crcs)) == 1
    oracle = None
    if grp.rank == 0:
        import oracle as _oracle
        oracle = _oracle
        q, s = oracle.quantize(w0_cpu.numpy())
        ref = oracle.w8a16_gemm(x.cpu().numpy(), q, s).astype(np.float32)
        got = y0.cpu().numpy().astype(np.float32)
        parity["tier_a_max_abs_err_vs_oracle"] = float(np.abs(got - ref).max())
        parity["tier_a_ok"] = bool(np.all(np.abs(got - ref) <= 1e-3 * np.abs(ref).max() + 2e-3 * np.abs(ref)))
        with torch.no_grad():
            y_lin = lin_cpu(x.cpu()).numpy().astype(np.float32)
        parity["tier_b_max_abs_err_vs_cpu_linear_fp16"] = float(np.abs(got - y_lin).max())
        parity["packed_bit_exact"] = bool(np.array_equal(sets[0][0].cpu().numpy(), oracle.gfx950_pack(q)))

    # ---- timed region: W warm-up steps, then K-step graphs replayed for >= min_timed_ms ----
    gemv_steps(0, warmup)
    torch.cuda.synchronize()
    graphs = capture_graphs(gemv_steps, steps, nbuf)
    seconds, replays = timed_replays(grp, graphs, steps, min_s)
    timed_steps = replays * steps
    step_bytes = gemv_bytes(M, N, K)
    value = grp.world_size * timed_steps * step_bytes / seconds / 1e9

    # ---- roofline of the dominant kernel: event pair around every launch, whole passes over the weight sets ----
    n_ev = max(nbuf, (min(max(steps, 400), 4000) // nbuf) * nbuf)
    gemv_steps(0, nbuf)
    torch.cuda.synchronize()
    k_mean, k_med, k_min = dispatch_kernel_time(lambda: gemv_steps(0, n_ev), n_ev)
    achieved = step_bytes / k_mean / 1e9
    sink = torch.zeros(16, dtype=torch.int32, device=dev)
    stream_ptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = _lib.lib()

    def read_only():
        for i in range(n_ev):
            _lib.check(L.eetq_diag_stream_read(ctypes.c_void_p(sets[i % nbuf][0].data_ptr()), K * N,
                                               ctypes.c_void_p(sink.data_ptr()), stream_ptr))

    def empty():
        for i in range(n_ev):
            _lib.check(L.eetq_diag_empty(ctypes.c_void_p(sink.data_ptr()), N // 16, 1024, stream_ptr))
    read_only()
    empty()
    torch.cuda.synchronize()
    f_mean, f_med, f_min = dispatch_kernel_time(read_only, n_ev)
    e_mean, e_med, e_min = dispatch_kernel_time(empty, n_ev)
    traffic_doc = load_traffic()
    roofline = {"kernel": "gemv_kernel<M=1,16 waves x 4 tiles,exact,xreg>", "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic_doc.get("gemv_hbm_bytes_per_launch"),
                "traffic_source": traffic_doc.get("source", "profiles/pmc_traffic.json (rocprofv3 --pmc pass, tools/profile_bench.sh)"),
                "algorithmic_bytes_per_launch": step_bytes, "launches_timed": n_ev,
                "kernel_us_mean": round(k_mean * 1e6, 3), "kernel_us_median": round(k_med * 1e6, 3),
                "kernel_us_min": round(k_min * 1e6, 3),
                "method": "HIP start/stop events on each dispatch packet (begin->end of the dispatch, as rocprofv3 "
                          "--kernel-trace); rocprofv3 summary of this command: profiles/*_bench_kernel_stats.csv",
                "method_floor_us": round(e_mean * 1e6, 3),
                "read_only_floor": {"what": "kernel that only loads the same 16 MiB (16 B/lane nt loads), same timing method",
                                    "kernel_us_mean": round(f_mean * 1e6, 3), "gbps": round(K * N / f_mean / 1e9, 1),
                                    "frac_of_peak": round(K * N / f_mean / 1e9 / HBM_PEAK_GBPS, 4)},
                "whole_step": {"what": "graph-replayed step incl. the dependent-launch boundary (= value)",
                               "us": round(seconds * 1e6 / timed_steps, 3),
                               "frac_of_peak": round(step_bytes / (seconds / timed_steps) / 1e9 / HBM_PEAK_GBPS, 4)}}

    # ---- the other half of the metric: fused dequant-GEMM, M = 1024 ----
    Mg = 1024
    torch.manual_seed(2)
    xg = torch.rand(Mg, K, dtype=torch.float16).to(dev)
    yg = [torch.empty(Mg, N, dtype=torch.float16, device=dev) for _ in range(2)]

    def gemm_steps(first, count):
        for i in range(first, first + count):
            w, s = sets[i % nbuf]
            ops.w8_a16_gemm_(xg, w, s, yg[i % 2], Mg, N, K)

    gemm_steps(0, 60)  # warm-up: lets the clocks settle under MFMA load
    torch.cuda.synchronize()
    ggraphs = capture_graphs(gemm_steps, args.gemm_steps, nbuf)
    gsec, greplays = timed_replays(grp, ggraphs, args.gemm_steps, min_s)
    gtimed = greplays * args.gemm_steps
    flops = 2.0 * Mg * N * K
    n_gev = max(nbuf, (args.gemm_steps // nbuf) * nbuf)
    g_mean, g_med, g_min = dispatch_kernel_time(lambda: gemm_steps(0, n_gev), n_gev)
    gemm_roofline = {"kernel": "gemm_tile_kernel", "bound": "mfma", "achieved": round(flops / g_mean / 1e12, 2),
                     "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(flops / g_mean / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                     "kernel_us_mean": round(g_mean * 1e6, 2), "kernel_us_min": round(g_min * 1e6, 2),
                     "traffic": traffic_doc.get("gemm_m1024_hbm_bytes_per_launch"),
                     "algorithmic_flops_per_launch": flops, "launches_timed": n_gev,
                     "whole_job_tflops": round(grp.world_size * gtimed * flops / gsec / 1e12, 2),
                     "ms_per_step": round(gsec * 1e3 / gtimed, 5), "timed_steps": gtimed}
    roofline["gemm_m1024"] = gemm_roofline
    gemm = {"metric": "dequant-GEMM TFLOPS @ M=1024, N=K=4096", "value": gemm_roofline["whole_job_tflops"],
            "unit": "TFLOP/s", "steps": args.gemm_steps, "timed_steps": gtimed,
            "ms_per_step": gemm_roofline["ms_per_step"], "roofline": gemm_roofline}

    # ---- CPU baselines (rank 0, N = 1 only; bounded) ----
    cpu_baseline = None
    cpu_linear = None
    if grp.rank == 0 and grp.world_size == 1 and not args.no_cpu_baseline:
        q, s = oracle.quantize(w0_cpu.numpy())
        n_it, dt = cpu_gemv_baseline(oracle, x.cpu().numpy(), q, s, args.cpu_budget)
        cpu_baseline = {"value": round(n_it * step_bytes / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                        "sample": "%d calls of oracle_w8a16_gemm_f32acc (scalar C restatement) at M=1, N=K=4096 in %.1f s"
                                  % (n_it, dt), "ms_per_call": round(dt / n_it * 1e3, 3)}
        ncpu = os.cpu_count() or 1
        counts = sorted(set(t for t in (8, 16, 32, 64, 128, 256, ncpu) if t <= ncpu))
        s1 = cpu_linear_sweep(lin_cpu, x.cpu(), 7, counts)
        s1024 = cpu_linear_sweep(lin_cpu, xg.cpu(), 3, counts)
        b1 = min(s1, key=s1.get)
        b1024 = min(s1024, key=s1024.get)
        cpu_linear = {"what": "torch.nn.Linear(4096, 4096).half() forward on host CPU (north-star baseline), best over a "
                              "sweep of torch thread counts",
                      "host_cores": ncpu, "m1_ms": round(s1[b1] * 1e3, 3), "m1_threads": b1,
                      "m1_gbps_fp16_weights": round(2.0 * K * N / s1[b1] / 1e9, 2),
                      "m1024_ms": round(s1024[b1024] * 1e3, 3), "m1024_threads": b1024,
                      "m1024_gflops": round(flops / s1024[b1024] / 1e9, 1),
                      "sweep_m1_ms": {str(t): round(v * 1e3, 2) for t, v in s1.items()},
                      "sweep_m1024_ms": {str(t): round(v * 1e3, 2) for t, v in s1024.items()}}

    # ---- BASELINE configs[4]: the whole decode path on the same box (reported beside the headline, never as `value`) ----
    config5 = None
    if not args.no_config5:
        config5 = config5_leg(grp)   # (the 640 MiB of weight sets stay resident; 13 GB more is no issue on 288 GB)

    if grp.rank == 0:
        line = {
            "metric": "w8a16 GEMV GB/s @ M=1 and dequant-GEMM TFLOPS @ M=1024, N=K=4096",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": grp.world_size, "steps": steps, "warmup": warmup,
            "ms_per_step": round(seconds * 1e3 / timed_steps, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "w8a16 GEMV M=1, N=K=4096 (BASELINE configs[1]); %d distinct weight sets rotated (%d MiB)"
                                   % (nbuf, nbuf * K * N // (1 << 20)), "M": M, "N": N, "K": K,
                       "parallelism": "replicas x%d (no data-path collective)" % grp.world_size,
                       "launch": "%d HIP graph(s) of %d dependent launches, replayed %d times (%d timed steps, %.1f ms)"
                                 % (len(graphs), steps, replays, timed_steps, seconds * 1e3),
                       "timed_steps": timed_steps, "timed_ms": round(seconds * 1e3, 3)},
            "roofline": roofline, "secondary": gemm, "cpu_baseline": cpu_baseline, "cpu_linear_fp16": cpu_linear,
            "parity": parity, "config5": config5,
        }
        print(json.dumps(line))
    grp.close()


if __name__ == "__main__":
    main()
