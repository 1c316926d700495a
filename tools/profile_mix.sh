#!/bin/bash
# rocprofv3 kernel stats of the wiring-identity MSM batch of an n-constraint proof (tools/proof_msm_mix.py), concurrent and
# serialised (ZKHIP_TUNE=msm_serial=1: per-kernel times = work):  tools/profile_mix.sh <tag> [n]
set -u
TAG=${1:-prof}; N=${2:-20}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/proof_msm_mix.py $N 5 > $OUT/${TAG}_mix_n${N}.txt 2>&1
for S in 0 1; do
  rm -rf /tmp/prof_mix$S
  ZKHIP_TUNE=msm_serial=$S rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_mix$S -o mix -- python $REPO/tools/proof_msm_mix.py $N 5 >> $OUT/${TAG}_mix_n${N}.txt 2>/tmp/prof_mix$S.err
  cp $(find /tmp/prof_mix$S -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_mix_n${N}_serial${S}_kernel_stats.csv
  cp $(find /tmp/prof_mix$S -name '*kernel_trace.csv' | head -1) /tmp/mix_trace$S.csv
done
python $REPO/tools/kstats.py $OUT/${TAG}_mix_n${N}_serial1_kernel_stats.csv | head -24 >> $OUT/${TAG}_mix_n${N}.txt
cat $OUT/${TAG}_mix_n${N}.txt
