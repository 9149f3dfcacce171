"""Round 5: the split-K planner's cost model (gemm_splitk.hip::splitk_plan) against the measured time of EVERY plan.
tools/experiments/splitk_rows_scan.py forces each (column blocks, K slices, ring, row groups) plan on a (K, N) x M grid and
records its time next to AUTO's (profiles/r05_splitk_plan_regret*.jsonl); this script replays the planner's model over those
tables on the CPU -- which plan would it pick, how far is that plan's MEASURED time from the best measured plan -- and fits the
model's constants by random search on the summed regret (+ a penalty per point above 5 %).  The constants in splitk_plan are the
rounded result; `--fit` repeats the search, without it the script scores the shipped constants.
usage: python tools/experiments/splitk_plan_fit.py [--fit] [tables ...]"""
import glob, json, math, os, random, sys

NCU = 256
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHIPPED = {"a": 0.27, "b": 0.14, "shallow": 1.3, "r2": 2.2, "r4": 2.8, "slab": 0.088, "fix": 1.45, "rowpen": 0.27, "share": 1.22, "share3": 1.6}
ROUND2 = {"a": 16 * 0.014, "b": 8 * 0.014, "shallow": 1.15, "r2": 2.0, "r4": 3.6, "slab": 0.05, "fix": 2.3, "rowpen": 0.3, "idle": 1.25}


def geometry(M, N, K, nb, s, r):
    MT = -(-M // (32 * r))
    tiles, steps = -(-N // (32 * nb)), (K // 64 + 3) // 4
    wgs = tiles * s * r
    stages = 3 if (MT <= 2 and wgs <= NCU) else 2
    return MT, steps, wgs, stages


def legal(M, N, K, nb, s, ring, r):
    MT, steps, wgs, stages = geometry(M, N, K, nb, s, r)
    return MT <= 4 and -(-M // (32 * MT)) == r and not (s > 1 and steps // s < 2) and ring == 11 * stages


def model(M, N, K, nb, s, ring, r, P):
    MT, steps, wgs, stages = geometry(M, N, K, nb, s, r)
    lds = stages * (16 * MT + 8 * nb)                                   # KiB of LDS per workgroup
    if "share3" in P:   # the shipped form: up to three workgroups per CU where the LDS holds them
        cap = min(3, 160 // lds) if ((s == 1 or r > 1) and stages == 2 and MT <= 2 and lds <= 80) else 1
        rounds = -(-wgs // (NCU * cap))
        per_cu = min(cap, -(-wgs // NCU)) if rounds == 1 else cap
        share = {1: 1.0, 2: P["share"], 3: P["share3"]}[per_cu]
    else:               # the round-2 form: two per CU, each at half speed
        per_cu = 2 if ((s == 1 or r > 1) and stages == 2 and MT <= 2 and lds <= 80) else 1
        rounds = -(-wgs // (NCU * per_cu))
        share = P.get("share", 2.0) if per_cu == 2 else 1.0
    t = -(-steps // s) * (P["a"] * MT + P["b"] * nb) * share * (1.0 if stages == 3 else P["shallow"])
    t += 0.0 if s == 1 else ((P["r2"] if s == 2 else P["r4"]) + P["slab"] * MT * nb * s)
    t = P["fix"] + rounds * t
    if wgs * 2 <= NCU:
        t *= P.get("idle", 1.0)
    return t + P["rowpen"] * (r - 1)


def load(files):
    data = []
    for f in files:
        for line in open(f):
            row = json.loads(line)
            M, N, K = row["M"], row["N"], row["K"]
            if M > 128:
                continue
            plans = {tuple(int(v) for v in k.split(",")): v for k, v in row.items() if "," in k and isinstance(v, float)}
            plans = {k: v for k, v in plans.items() if legal(M, N, K, *k)}
            if len(plans) >= 2:
                data.append((M, N, K, plans, min(plans.values())))
    return data


def score(data, P, verbose=False):
    tot, bad = 0.0, 0
    for M, N, K, plans, best in data:
        pick = min(plans, key=lambda k: model(M, N, K, *k, P))
        reg = plans[pick] / best - 1
        tot += reg
        bad += reg > 0.05
        if verbose and reg > 0.05:
            print("   K=%d N=%d M=%d: pick %s %.2f us, best %s %.2f (+%.1f %%)" % (K, N, M, pick, plans[pick], min(plans, key=plans.get), best, 100 * reg))
    return tot / len(data), bad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--fit"]
    files = args or sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_splitk_plan_regret*.jsonl")))
    data = load(files)
    print("%d points from %d tables" % (len(data), len(files)))
    print("round-2 constants: mean regret %.2f %%, %d points above 5 %%" % ((lambda m, b: (100 * m, b))(*score(data, ROUND2))))
    print("shipped constants: mean regret %.2f %%, %d points above 5 %%" % ((lambda m, b: (100 * m, b))(*score(data, SHIPPED))))
    score(data, SHIPPED, verbose=True)
    if "--fit" in sys.argv:
        random.seed(1)
        obj = lambda P: (lambda m, b: m + 0.002 * b)(*score(data, P))
        best = (obj(SHIPPED), dict(SHIPPED))
        for _ in range(4000):
            P = dict(best[1])
            for k in random.sample(list(P), random.randint(1, 3)):
                P[k] *= math.exp(random.gauss(0, 0.15))
            o = obj(P)
            if o < best[0]:
                best = (o, P)
        print("fit:", {k: round(v, 4) for k, v in best[1].items()}, "mean regret %.2f %%, %d above 5 %%" % ((lambda m, b: (100 * m, b))(*score(data, best[1]))))
