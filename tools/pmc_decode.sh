#!/bin/bash
# HBM/fabric read traffic per launch of the decode-step kernels (one-launch attention, 13B GEMVs incl. the glu8 epilogue),
# from separate rocprofv3 --pmc passes (kernel-trace only), corrected as tools/profile_bench.sh does (128 B per read request).
# usage (GPU box, repo root): tools/pmc_decode.sh [tag]   ->  gpurun_out/<tag>_pmc_decode.json
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
for pass in "TCC_EA0_RDREQ_sum" "FETCH_SIZE"; do
    d=/tmp/prof_pmcd_$pass
    rm -rf $d
    ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python "$ROOT/tools/attn_bench.py" --layers 12 --splits default > /dev/null 2>> "$OUT/${TAG}_pmc_decode.err" )
    d=/tmp/prof_pmcg_$pass
    rm -rf $d
    ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python "$ROOT/examples/llama_generate.py" --accelerate --graph --layers 4 --new 20 > /dev/null 2>> "$OUT/${TAG}_pmc_decode.err" )
done
python - > "$OUT/${TAG}_pmc_decode.json" <<'PYEOF'
import csv, glob, json
def mean_counter(prefix, counter, kern, grid=None):
    v = []
    for f in glob.glob("/tmp/%s_*/**/*counter_collection.csv" % prefix, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kern in r["Kernel_Name"] and (grid is None or r.get("Grid_Size") == str(grid)):
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (None, 0)
doc = {"source": "rocprofv3 --pmc (separate passes, kernel-trace only) on tools/attn_bench.py and a 4-layer examples/llama_generate.py --accelerate --graph; mean per dispatch; read requests x 128 B, FETCH_SIZE KiB x 2 (gfx950, MI355X_MICROARCH.md)"}
for key, prefix, kern, algo in (("rope_attn_decode(13B shapes, 1025 rows)", "prof_pmcd", "rope_attn_decode_kernel", 2 * 40 * 1025 * 128 * 2),
                                ("gemv_kernel(decode, qkv + glu8 gate|up mix)", "prof_pmcg", "gemv_kernelILi1ELi16ELi2", (78643200 + 141557760) // 2),
                                ("gemv_half_kernel(down 13824x5120)", "prof_pmcg", "gemv_half_kernelILi8ELi2ELi4", 70778880),
                                ("gemv_half_kernel(o 5120x5120)", "prof_pmcg", "gemv_half_kernelILi8ELi2ELi2", 26214400)):
    rd, n = mean_counter(prefix, "TCC_EA0_RDREQ_sum", kern)
    fs, _ = mean_counter(prefix, "FETCH_SIZE", kern)
    doc[key] = {"dispatches": n, "read_requests": rd, "hbm_read_bytes_per_launch": int(rd * 128) if rd else None,
                "FETCH_SIZE_KiB_x2_bytes": int(fs * 2048) if fs else None, "algorithmic_bytes": algo,
                "ratio": round(rd * 128 / algo, 3) if rd else None}
print(json.dumps(doc, indent=1))
PYEOF
cat "$OUT/${TAG}_pmc_decode.json"
