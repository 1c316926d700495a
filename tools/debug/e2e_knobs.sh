#!/bin/bash
# e2e sensitivity of the n = 20 proof (C++ host, leader mode) to the MSM knobs of the library: best of 5 proofs per setting
B=scalable-collaborative-zksnark_amd/host/bin/hyperplonk
N=${1:-20}
for t in "" msm_table_dc=-1 msm_table_dc=1 msm_size_classes=0 msm_split=0 msm_split=2 msm_fixq=16384 msm_fixq=262144 msm_quad=8192 msm_quad=131072 msm_qstep=1 msm_qstep=3; do
  best=$(ZKHIP_TUNE=$t $B --l 1 --n $N --reps 5 | grep "End: Distributed HyperPlonk" | awk '{print $4}' | sort -n | head -1)
  echo "n=$N ZKHIP_TUNE='$t' best $best s"
done
