# Round-3 experiment script (run on the GPU box from the repo root); output under profiles/ -- see profiles/README.md
export EETQ_AMD_TUNING=1   # the EETQ_AMD_QUANT_* A/B hooks answer only with this switch (csrc/common.hpp: tuning_env)
for nt in 0 1; do for fold in 0 1; do for strip in 1 4; do
echo "nt=$nt fold=$fold strip=$strip"; EETQ_AMD_QUANT_NT=$nt EETQ_AMD_QUANT_FOLD=$fold EETQ_AMD_QUANT_STRIP=$strip python tools/quant_bench.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
print('   '+'  '.join('%dx%d %s %.1f'%(d['K'],d['N'],d['dtype'][-2:],d['quant_weights_us']) for d in map(json.loads,sys.stdin)))"
done; done; done
