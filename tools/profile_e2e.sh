#!/bin/bash
# rocprofv3 kernel stats of one collaborative-HyperPlonk proof (leader mode) -> gpurun_out/<tag>_e2e_n<N>_kernel_stats.csv
set -u
TAG=${1:-prof}; N=${2:-20}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_e2e -o e2e -- python $REPO/tools/hyperplonk_bench.py --n $N --reps 3 > $OUT/${TAG}_e2e_n${N}.json 2>/tmp/prof_e2e.err
cp $(find /tmp/prof_e2e -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_e2e_n${N}_kernel_stats.csv
cat $OUT/${TAG}_e2e_n${N}.json
