#!/bin/bash
# per-kernel durations of the sumcheck family at the given log2 sizes (rocprofv3 kernel trace + stats)
#   tools/sc_prof.sh <tag> <log2 sizes...>   -> gpurun_out/<tag>_sc_kernel_stats.csv
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sc
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_sc -o sc -- python $REPO/tools/sc_time.py "$@" > $OUT/${TAG}_sc_time_under_rocprof.txt 2>/tmp/prof_sc.err
cp $(find /tmp/prof_sc -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_sc_kernel_stats.csv
python $REPO/tools/kstats.py $OUT/${TAG}_sc_kernel_stats.csv
