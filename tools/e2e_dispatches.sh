#!/bin/bash
# the longest dispatches of one e2e proof, by kernel (rocprofv3 kernel trace): tools/e2e_dispatches.sh <n>
N=${1:-20}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd
rocprofv3 --kernel-trace -f csv -d /tmp/pd -o e -- python $REPO/tools/hyperplonk_bench.py --n $N --reps 1 --no-check > /dev/null 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob('/tmp/pd/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    by[name].append(((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size']) if 'Grid_Size' in r else int(r.get('Grid_Size_X', 0))))
t0 = min(int(r['Start_Timestamp']) for r in rows if 'k_digits' in r['Kernel_Name'])
t1 = max(int(r['End_Timestamp']) for r in rows)
print(f"span from first k_digits to last kernel: {(t1 - t0) / 1e6:.1f} ms")
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in rows if int(r['Start_Timestamp']) >= t0)
busy, cur_e, gaps, last_name = 0, iv[0][0], [], ''
for s_, e_, nm in iv:
    if s_ > cur_e:
        gaps.append((s_ - cur_e, last_name, nm))
        busy += 0
        cur_s = s_
    busy += max(0, e_ - max(s_, cur_e))
    if e_ > cur_e:
        cur_e, last_name = e_, nm
print(f"GPU busy (union of kernel intervals): {busy / 1e6:.1f} ms; idle {((t1 - t0) - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
gaps.sort(reverse=True)
for g, a, b in gaps[:14]:
    print(f"   gap {g / 1e3:8.1f} us   after {a[:40]:40s} before {b[:40]}")
import collections
cnt = collections.Counter()
for g, a, b in gaps:
    cnt[(a[:28], b[:28])] += g
print("   idle by (after, before):", [(k, round(v / 1e3)) for k, v in cnt.most_common(8)])
for name in ('zk::k_accum_tiles<zk::CvG1>', 'zk::k_fixup<zk::CvG1>', 'zk::k_halve<zk::CvG1>', 'zk::k_part_sort', 'zk::k_fixup_long<zk::CvG1>', 'zk::k_fixup_quad', 'zk::k_halve_quad'):
    d = sorted(by.get(name, []), reverse=True)
    print(name, 'calls', len(d), 'total_ms %.2f' % (sum(x[0] for x in d) / 1e3))
    print('   top:', ' '.join(f"{us:.0f}us/{g}" for us, g in d[:12]))
PY
