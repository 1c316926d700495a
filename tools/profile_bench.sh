#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root):
#   tools/profile_bench.sh <tag>      -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_hbm_traffic.csv
# Kernel timing and PMC counters are separate runs (never combined with other trace domains).
set -u
TAG=${1:-prof}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -o stats -- python $REPO/bench.py --steps 5 --no-cpu > $OUT/${TAG}_bench_under_rocprof.json 2>/tmp/prof_stats.err
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -f csv -d /tmp/prof_$C -o pmc -- python $REPO/bench.py --steps 3 --no-cpu > /dev/null 2>/tmp/prof_$C.err
done
python $REPO/tools/pmc_summary.py $(find /tmp/prof_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/prof_WRITE_SIZE -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_pmc_hbm_traffic.csv
head -12 $OUT/${TAG}_kernel_stats.csv
cat $OUT/${TAG}_pmc_hbm_traffic.csv
