"""Where does the split-K kernel go wrong with two workgroups per CU?  Launches the library's kernel (MT=2, NB=1, ring 2x2) through
tools/experiments/libsk_debug.so with its own slabs, at 80 KiB of LDS (two workgroups per CU) and at 84 KiB (one), and compares
  (a) every slice's published partial tile (slab) with the partial sums of that K slice computed by torch, and
  (b) y with the sum of the reference partials and with the sum of the slabs the launch actually published.
usage: python tools/experiments/sk_debug.py [N K M S]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from eetq_amd import ops

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "libsk_debug.so"))
lib.sk_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
SMEM, SLAB = lib.sk_smem(), lib.sk_slab_floats()
dev = "cuda:0"
N, K, M, S = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (4096, 4096, 64, 2)
BN = 32
tiles = (N + BN - 1) // BN
g = torch.Generator(device=dev); g.manual_seed(3)
w = ((torch.rand(K, N, device=dev, generator=g) * 2 - 1) / K ** 0.5).half()
packed, scales = ops.quant_weights(w, torch.int8, False)
raw = ops.unprocess_weights(packed, "gfx950")                       # int8 [K, N]
wd = (raw.half() * scales.half()[None, :])                           # the kernel's fp16 dequantised weights
x = torch.rand(M, K, device=dev, generator=g).half()
KT = K // 64
steps_total = (KT + 3) // 4
bounds = [(steps_total * s // S * 256, min(steps_total * (s + 1) // S * 256, K)) for s in range(S)]
ref = torch.stack([x[:, a:b].float() @ wd[a:b].float() for a, b in bounds])       # [S, M, N]

lane = torch.arange(64, device=dev)
def decode(slabs):                                                    # [tiles, S, SLAB] -> [S, 64, N]
    t = slabs.view(tiles, S, 2, 4, 64, 4)                             # mt, wave, lane, i
    out = torch.zeros(S, 64, tiles * BN, device=dev)
    for mt in range(2):
        for wv in range(4):
            m = 32 * mt + (lane & 31)                                 # [64]
            for i in range(4):
                ncol = 8 * wv + 4 * (lane >> 5) + i                   # [64]
                vals = t[:, :, mt, wv, :, i]                          # [tiles, S, 64]
                n_idx = (torch.arange(tiles, device=dev)[:, None] * BN + ncol[None, :])   # [tiles, 64]
                for s in range(S):
                    out[s, m[None, :].expand(tiles, 64), n_idx] = vals[:, s, :]
    return out[:, :M, :N]

for lds, label in ((SMEM, "two workgroups per CU (%d B of LDS)" % SMEM), (84 * 1024, "one workgroup per CU (84 KiB)")):
    bad_runs = 0
    for rep in range(3):
        slabs = torch.full((tiles, S, SLAB), float("nan"), device=dev)
        counters = torch.zeros(tiles, dtype=torch.int32, device=dev)
        y = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        rc = lib.sk_launch(x.data_ptr(), packed.data_ptr(), scales.data_ptr(), y.data_ptr(), M, N, K, S, slabs.data_ptr(),
                           counters.data_ptr(), lds, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0, rc
        part = decode(slabs)
        perr = (part - ref).abs()
        yerr = (y.float() - ref.sum(0)).abs()
        yslab = (y.float() - part.sum(0)).abs()                       # y vs what was actually published
        pbad = (perr > 2e-2) | torch.isnan(part)
        ybad = (yerr > 2e-2) | torch.isnan(y.float())
        if pbad.any() or ybad.any():
            bad_runs += 1
            print("  rep %d: %d partial elements off (slices %s), %d outputs off, %d outputs disagree with their own slabs" % (
                rep, int(pbad.sum()), sorted(set(pbad.nonzero()[:, 0].tolist())), int(ybad.sum()), int((yslab > 2e-2).sum())))
            nz = pbad.nonzero()
            print("     bad partials by m %% 16: %s | by column %% 4: %s | by column-in-tile // 8 (wave): %s | slab values: %s" % (
                torch.bincount(nz[:, 1] % 16, minlength=16).tolist(), torch.bincount(nz[:, 2] % 4, minlength=4).tolist(),
                torch.bincount((nz[:, 2] % BN) // 8, minlength=4).tolist(), sorted(set(part[pbad].tolist()))[:4]))
            print("     bad tiles: %d distinct, lowest %d, of %d; slices %s" % (
                len(set((nz[:, 2] // BN).tolist())), int((nz[:, 2] // BN).min()), tiles, torch.bincount(nz[:, 0], minlength=S).tolist()))
            idx = nz[:2].tolist()
            for s_, m_, n_ in idx:
                print("     slice %d m %d n %d (tile %d, col-in-tile %d): slab %.5f reference %.5f" % (
                    s_, m_, n_, n_ // BN, n_ % BN, float(part[s_, m_, n_]), float(ref[s_, m_, n_])))
            yi = ybad.nonzero()[:1].tolist()
            for m_, n_ in yi:
                print("     y[%d, %d] = %.5f, sum of reference partials %.5f, sum of published slabs %.5f" % (
                    m_, n_, float(y[m_, n_]), float(ref.sum(0)[m_, n_]), float(part.sum(0)[m_, n_])))
    print("%s: N=%d K=%d M=%d S=%d: %d of 3 launches wrong" % (label, N, K, M, S, bad_runs))
