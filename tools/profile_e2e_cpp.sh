#!/bin/bash
# rocprofv3 kernel stats of the collaborative-HyperPlonk proof driven by the C++ host (leader mode): setup (window tables) + REPS proofs
#   -> gpurun_out/<tag>_e2e_cpp_n<N>_kernel_stats.csv, gpurun_out/<tag>_e2e_cpp_n<N>.txt
set -u
TAG=${1:-prof}; N=${2:-20}; REPS=${3:-4}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_e2e_cpp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_e2e_cpp -o e2e -- $REPO/scalable-collaborative-zksnark_amd/host/bin/hyperplonk --l 1 --n $N --reps $REPS --digest > $OUT/${TAG}_e2e_cpp_n${N}.txt 2>/tmp/prof_e2e_cpp.err
cp $(find /tmp/prof_e2e_cpp -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_e2e_cpp_n${N}_kernel_stats.csv
tail -8 $OUT/${TAG}_e2e_cpp_n${N}.txt; head -8 $OUT/${TAG}_e2e_cpp_n${N}_kernel_stats.csv | cut -c1-160
