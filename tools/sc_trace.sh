#!/bin/bash
# per-dispatch view of one mode of the sumcheck family: kernel, grid, average duration, and the gaps between the
# dispatches of the last call (rocprofv3 kernel trace)
#   tools/sc_trace.sh <mode: product|plain|fold|open> <log2 size>
set -u
MODE=$1; LG=$2
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sct
SC_MODE=$MODE rocprofv3 --kernel-trace -f csv -d /tmp/prof_sct -o sc -- python $REPO/tools/sc_time.py $LG > /tmp/prof_sct.out 2>/tmp/prof_sct.err
python - "$(find /tmp/prof_sct -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void zk::", "") for r in rows]
# one call = the dispatches between two final stages; take the last complete call
seq = []
for r, n in zip(rows, names):
    seq.append((n, int(r.get("Grid_Size") or r["Grid_Size_X"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
# find the period: the last dispatch name repeats every call
last = seq[-1][0:2]
idx = [i for i, s in enumerate(seq) if s[0:2] == last]
per = idx[-1] - idx[-2]
calls = [seq[i - per + 1 : i + 1] for i in idx[-8:]]
for j in range(per):
    d = [c[j][3] - c[j][2] for c in calls]
    gap = [c[j][2] - c[j - 1][3] for c in calls] if j else [0]
    print(f"{calls[-1][j][0]:28s} grid {calls[-1][j][1]:9d}  {sum(d)/len(d)/1e3:8.2f} us   gap before {sum(gap)/len(gap)/1e3:6.2f} us")
tot = [c[-1][3] - c[0][2] for c in calls]
print(f"first start -> last end: {sum(tot)/len(tot)/1e3:.2f} us")
PY
