#!/bin/bash
# kernel totals of an arbitrary command: tools/debug/trace_cmd.sh <cmd...>
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ptc
rocprofv3 --kernel-trace --stats -f csv -d /tmp/ptc -o t -- "$@" > /tmp/ptc.out 2>/tmp/ptc.err
tail -2 /tmp/ptc.out
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ptc/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'].split('(')[0][:50]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
