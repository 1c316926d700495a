#!/bin/bash
# GPU busy time of the LAST proof of a run (kernel trace, union of kernel intervals): tools/debug/e2e_busy.sh <n>
N=${1:-20}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb
rocprofv3 --kernel-trace -f csv -d /tmp/pb -o e -- python $REPO/tools/hyperplonk_bench.py --n $N --reps 2 --no-check > /tmp/pb.out 2>/tmp/pb.err
tail -1 /tmp/pb.out | cut -c1-400
python - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob('/tmp/pb/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in csv.DictReader(open(f))))
# segments separated by idle gaps > 3 ms
segs, cur, end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - end > 3e6:
        segs.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
segs.append(cur)
print("segments (ms):", [round((max(x[1] for x in s) - s[0][0]) / 1e6, 1) for s in segs])
for i in range(max(1, len(segs) - 4), len(segs)):
    pe = max(x[1] for x in segs[i - 1])
    print(f"   segment {i}: starts {(segs[i][0][0] - pe) / 1e6:.2f} ms after the previous one ended; previous ends with {[x[2][:24] for x in segs[i - 1][-3:]]}, this one starts with {[x[2][:24] for x in segs[i][:4]]}")
s = segs[-1]
t0, t1 = s[0][0], max(x[1] for x in s)
busy, ce = 0, t0
gaps = []
for a, b, nm in s:
    if a > ce: gaps.append((a - ce, nm))
    busy += max(0, b - max(a, ce)); ce = max(ce, b)
ce2 = t0; prev = ''
for a, b, nm in s:
    if a > ce2 + 150e3: print(f"   gap {(a - ce2) / 1e3:7.0f} us at +{(ce2 - t0) / 1e6:6.2f} ms   after {prev[:34]:34s} before {nm[:34]}")
    if b > ce2: ce2, prev = b, nm
print(f"last proof: span {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms ({100 * busy / (t1 - t0):.1f} %), {len(gaps)} gaps, largest:", [(round(g / 1e3), n[:30]) for g, n in sorted(gaps, reverse=True)[:8]])
tot = defaultdict(float)
for a, b, nm in s: tot[nm.split('<')[0]] += (b - a) / 1e6
print("kernel time sums (ms, overlapping streams add up):", sorted(((round(v, 2), k) for k, v in tot.items()), reverse=True)[:12])
PY
