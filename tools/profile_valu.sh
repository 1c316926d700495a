#!/bin/bash
# VALU issue counters of the headline loop's kernels (run from the repo root on the GPU box):
#   tools/profile_valu.sh <tag>   -> gpurun_out/<tag>_accum_valu_counters.csv
# rocprofv3 --pmc passes only (no other trace domain), two counters per pass.
set -u
TAG=${1:-prof}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FILES=""
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVES"; do
  D=/tmp/valu_$(echo $C | tr ' ' '_')
  rm -rf $D
  rocprofv3 --pmc $C -f csv -d $D -o pmc -- python $REPO/bench.py --steps 10 --no-cpu --no-extra > /dev/null 2>$D.err || tail -3 $D.err
  F=$(find $D -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && FILES="$FILES $F"
done
python - $FILES > $OUT/${TAG}_accum_valu_counters.csv <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if "zk::" not in name:
            continue
        k = (name, r.get("Grid_Size", "?"), r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
print("kernel,grid_size,counter,dispatches,avg_value_per_dispatch   (rocprofv3 --pmc passes of `bench.py --steps 10 --no-cpu --no-extra`: the 2^20 d_msm loop)")
for (name, grid, c), (n, tot) in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][1], kv[0][2])):
    print(f"{name},{grid},{c},{n},{tot / n:.1f}")
PY
grep k_accum_tiles $OUT/${TAG}_accum_valu_counters.csv
