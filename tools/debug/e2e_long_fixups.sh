#!/bin/bash
# which MSM classes of one n-variable proof spend time in k_fixup_long?  tools/debug/e2e_long_fixups.sh <n>
N=${1:-20}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pl
ZKHIP_TUNE=msm_debug=1 rocprofv3 --kernel-trace -f csv -d /tmp/pl -o e -- python $REPO/tools/hyperplonk_bench.py --n $N --reps 0 --no-check > /tmp/pl.out 2>/tmp/pl.err
grep zk-class /tmp/pl.err | sort | uniq -c | sort -k1,1nr | head -40
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = {}
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    q = r.get('Queue_Id', '?') + '/' + r.get('Stream_Id', '?')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
    if 'k_digits' in name or 'k_accum_tiles' in name or 'k_fixup<' in name:
        last.setdefault(q, {})[name.split('<')[0]] = (g, round(d))
    if 'k_fixup_long' in name and d > 100:
        print(f"k_fixup_long {d:8.0f} us on {q}: ", last.get(q))
PY
