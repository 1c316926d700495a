#!/bin/bash
# kernel timeline of the C++ host's proofs: tools/profile_timeline.sh <tag> [n=20] [reps=4] -> gpurun_out/<tag>_timeline_n<N>.txt
set -u
TAG=${1:-prof}; N=${2:-20}; REPS=${3:-4}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -f csv -d /tmp/prof_tl -o tl -- $REPO/scalable-collaborative-zksnark_amd/host/bin/hyperplonk --l 1 --n $N --reps $REPS > $OUT/${TAG}_timeline_n${N}.txt 2>/tmp/prof_tl.err
python $REPO/tools/e2e_timeline.py $(find /tmp/prof_tl -name '*kernel_trace.csv' | head -1) 1 >> $OUT/${TAG}_timeline_n${N}.txt
tail -40 $OUT/${TAG}_timeline_n${N}.txt
