#!/bin/bash
# SQ issue / wait counters of the sumcheck-family kernels, one rocprofv3 --pmc pass (8 SQ slots) per size:
#   tools/sc_valu.sh <tag> <mode> <log2 n> [...]   -> gpurun_out/<tag>_sc_valu_<mode>.csv
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md, PMC slots);
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
set -u
TAG=$1; MODE=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RES=$OUT/${TAG}_sc_valu_${MODE}.csv
echo "log2n,kernel,grid,dispatches,counter,avg_per_dispatch" > $RES
for N in "$@"; do
  D=/tmp/scv_$N; rm -rf $D
  SC_MODE=$MODE rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -f csv -d $D -o pmc -- python $REPO/tools/sc_time.py $N > $D.out 2>$D.err || tail -3 $D.err
  python - $N $(find $D -name '*counter_collection.csv' | head -1) >> $RES <<'PY'
import csv, sys
from collections import defaultdict
n, path = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "zk::" not in name: continue
    k = (name, r.get("Grid_Size", "?"), r["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (name, grid, c), (cnt, tot) in sorted(acc.items()):
    print(f"{n},{name},{grid},{cnt},{c},{tot / cnt:.1f}")
PY
done
cat $RES
