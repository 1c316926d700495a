#!/bin/bash
# Batched-affine go / no-go gate (tools/ubench/batch_affine.hip), run from the repo root on the GPU box:
#   tools/profile_batch_affine.sh <tag>  -> gpurun_out/<tag>_batch_affine.txt (timings) and <tag>_batch_affine_pmc.csv (counters per dispatch)
# rocprofv3 --pmc passes only (no other trace domain), FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md).
set -u
TAG=${1:-prof}
REPO=$(pwd)
OUT=$REPO/gpurun_out
EXE=$REPO/tools/ubench/batch_affine
mkdir -p $OUT
for L in 17 16; do timeout 300 $EXE $L 5; done > $OUT/${TAG}_batch_affine.txt 2>&1
cd /tmp && export TMPDIR=/tmp
FILES=""
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  D=/tmp/ba_$(echo $C | tr ' ' '_')
  rm -rf $D
  timeout 600 rocprofv3 --pmc $C -f csv -d $D -o pmc -- $EXE 17 1 > /dev/null 2>$D.err || tail -3 $D.err
  F=$(find $D -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && FILES="$FILES $F"
done
python - $FILES > $OUT/${TAG}_batch_affine_pmc.csv <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)  # (kernel, ordinal of that kernel's dispatch) -> counter -> value
for path in sys.argv[1:]:
    seen = defaultdict(int)
    last = {}
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"])):
        name = r["Kernel_Name"].split("(")[0]
        if "k_tree_level" not in name and "k_xyzz_tiles" not in name:
            continue
        did = r["Dispatch_Id"]
        if (name, did) not in last:
            last[(name, did)] = seen[name]
            seen[name] += 1
        rows[(name, last[(name, did)])][r["Counter_Name"]] = float(r["Counter_Value"])
print("kernel,dispatch_of_that_kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes(2*FETCH+WRITE),SQ_INSTS_VALU,SQ_BUSY_CYCLES   (`batch_affine 17 1`: every kernel runs twice per level -- warm-up + 1 rep; levels 1..6 in order; FETCH_SIZE doubled per the gfx950 correction)")
for (name, k), c in sorted(rows.items()):
    f, w = c.get("FETCH_SIZE", float("nan")), c.get("WRITE_SIZE", float("nan"))
    print(f"{name},{k},{f:.0f},{w:.0f},{(2 * f + w) * 1024:.0f},{c.get('SQ_INSTS_VALU', float('nan')):.0f},{c.get('SQ_BUSY_CYCLES', float('nan')):.0f}")
PY
cat $OUT/${TAG}_batch_affine.txt | head -20
cat $OUT/${TAG}_batch_affine_pmc.csv
