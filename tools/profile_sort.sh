#!/bin/bash
# HBM counter traffic of the MSM sort phase at 2^N points on a window table (separate rocprofv3 --pmc passes, FETCH_SIZE / WRITE_SIZE in KB per
# dispatch), round-5 kernels (ZKHIP_TUNE msm_fused_min=-1,msm_l2_tiled=-1) beside the round-6 ones:  tools/profile_sort.sh <tag> [N]
set -u
TAG=${1:-prof}; N=${2:-24}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in r06 r05; do
  T=""; [ $MODE = r05 ] && T="msm_fused_min=-1,msm_l2_tiled=-1"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ps_$C
    ZKHIP_TUNE=$T ZK_ONLY_TABLE=1 rocprofv3 --pmc $C -f csv -d /tmp/ps_$C -o pmc -- python $REPO/tools/msm_time.py $N > /dev/null 2>/tmp/ps_$C.err
  done
  echo "# $MODE: window-table MSM of 2^$N points" >> $OUT/${TAG}_sort_phase_pmc.csv
  python $REPO/tools/pmc_summary.py $(find /tmp/ps_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/ps_WRITE_SIZE -name '*counter_collection.csv' | head -1) | grep -E "counter|k_digits|k_part|k_tab|k_l2" >> $OUT/${TAG}_sort_phase_pmc.csv
done
cat $OUT/${TAG}_sort_phase_pmc.csv
