#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root):
#   tools/profile_bench.sh <tag>      -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_hbm_traffic.csv
# Kernel timing and PMC counters are separate runs (never combined with other trace domains).
# The stats pass runs the headline loop only (--no-extra): its averages are the 2^20 d_msm kernels;
# the PMC passes (30 headline steps, so that the headline geometry has the most dispatches) add the large legs (2^24 MSM, sumcheck family at 2^20 / 2^24 / 2^26), keyed by grid size.
set -u
TAG=${1:-prof}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -o stats -- python $REPO/bench.py --steps 10 --no-cpu --no-extra > $OUT/${TAG}_bench_under_rocprof.json 2>/tmp/prof_stats.err
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -f csv -d /tmp/prof_$C -o pmc -- python $REPO/bench.py --steps 30 --no-cpu --no-e2e > /dev/null 2>/tmp/prof_$C.err
done
python $REPO/tools/pmc_summary.py $(find /tmp/prof_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/prof_WRITE_SIZE -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_pmc_hbm_traffic.csv
head -12 $OUT/${TAG}_kernel_stats.csv
head -60 $OUT/${TAG}_pmc_hbm_traffic.csv
