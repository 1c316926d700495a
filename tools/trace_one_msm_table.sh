#!/bin/bash
# per-dispatch kernel timeline of one 2^N MSM on an SRS with its window table (after warm-up)
N=${1:-20}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_trt
rocprofv3 --kernel-trace -f csv -d /tmp/prof_trt -o tr -- python $REPO/tools/msm_time.py $N > /dev/null 2>/tmp/prof_trt.err
python - <<PY
import csv,glob
f=glob.glob('/tmp/prof_trt/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_digits' in r['Kernel_Name'] or 'k_tab_hist' in r['Kernel_Name']]
rows=rows[idx[-1]:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  grid={r.get('Grid_Size_X', r.get('Grid_Size','?')):>9}  {r['Kernel_Name'].split('(')[0][:60]}")
PY
