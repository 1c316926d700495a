#!/bin/bash
# kernel timeline of the C++ host's proofs: tools/profile_timeline.sh <tag> [n=20] [reps=4] -> gpurun_out/<tag>_timeline_n<N>.txt
set -u
TAG=${1:-prof}; N=${2:-20}; REPS=${3:-4}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -f csv -d /tmp/prof_tl -o tl -- $REPO/scalable-collaborative-zksnark_amd/host/bin/hyperplonk --l 1 --n $N --reps $REPS > $OUT/${TAG}_timeline_n${N}.txt 2>/tmp/prof_tl.err
TR=$(find /tmp/prof_tl -name '*kernel_trace.csv' | head -1)
python - $TR $OUT/${TAG}_timeline_n${N}_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:  # the three columns the analysis reads (the full trace is tens of MB)
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
    for r in rows:
        w.writerow([r["Kernel_Name"].split("(")[0], r["Start_Timestamp"], r["End_Timestamp"]])
PY
python $REPO/tools/e2e_timeline.py $OUT/${TAG}_timeline_n${N}_trace.csv 1 >> $OUT/${TAG}_timeline_n${N}.txt
tail -40 $OUT/${TAG}_timeline_n${N}.txt
