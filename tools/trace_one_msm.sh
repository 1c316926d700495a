#!/bin/bash
# per-dispatch kernel timeline of one 2^N MSM (after warm-up): gpurun_out/<tag>_msm_trace.txt
TAG=${1:-t}; N=${2:-20}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d /tmp/prof_tr -o tr -- python $REPO/tools/sweep_msm.py child $N 0 > /dev/null 2>/tmp/prof_tr.err
python - <<PY > $OUT/${TAG}_msm_trace.txt
import csv,glob
f=glob.glob('/tmp/prof_tr/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last MSM: find last k_digits
idx=[i for i,r in enumerate(rows) if 'k_digits' in r['Kernel_Name']]
rows=rows[idx[-1]:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  grid={r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','?'):>9}  {r['Kernel_Name'][:60]}")
PY
cat $OUT/${TAG}_msm_trace.txt
