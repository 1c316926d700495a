#!/bin/bash
# e2e sensitivity of the proof (C++ host, leader mode) to the window-table widths of the SRS levels: best of 5 proofs per policy
B=scalable-collaborative-zksnark_amd/host/bin/hyperplonk
N=${1:-20}
shift
for t in "$@"; do
  best=$(ZK_TABLE_POLICY=$t $B --l 1 --n $N --reps 5 | grep "End: Distributed HyperPlonk" | awk '{print $4}' | sort -n | head -1)
  echo "n=$N ZK_TABLE_POLICY='$t' best $best s"
done
