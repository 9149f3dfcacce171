#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp
d=/tmp/prof_bal_t; rm -rf $d
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $d -- python "$ROOT/tools/experiments/pmc_balanced.py" > /dev/null 2>&1 )
python - <<'PY' > "$ROOT/gpurun_out/trace_balanced.txt"
import csv,glob
rows=[]
for f in glob.glob('/tmp/prof_bal_t/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_splitk_kernel' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'BAL' if 'ELb1EEEv' in r['Kernel_Name'] else 'KSL'))
rows.sort()
for i in range(0,len(rows),20):
    grp=rows[i:i+20]
    d=sorted((e-s)/1e3 for s,e,_ in grp)
    gaps=sorted((grp[j+1][0]-grp[j][1])/1e3 for j in range(len(grp)-1))
    print(grp[0][2],'n',len(grp),'dur min %.2f med %.2f max %.2f'%(d[0],d[len(d)//2],d[-1]),'gap med %.2f'%gaps[len(gaps)//2])
PY
cat "$ROOT/gpurun_out/trace_balanced.txt"
