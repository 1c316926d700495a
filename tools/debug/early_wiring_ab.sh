# A/B of the early-wiring schedule (ZKHOST_EARLY_WIRING) against the runtime's hardware-queue count and the accumulation's slot share
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for Q in 4 8; do for SH in 100 75 50; do for S in 0 1; do
  echo "== GPU_MAX_HW_QUEUES=$Q msm_share=$SH ZKHOST_EARLY_WIRING=$S"
  GPU_MAX_HW_QUEUES=$Q ZKHIP_TUNE=msm_share=$SH ZKHOST_EARLY_WIRING=$S $H --l 1 --n 20 --reps 6 --check | grep -E "Distributed HyperPlonk|check:" | sort | head -6 | awk '{printf "%s ", $(NF-1)} END {print ""}'
done; done; done
GPU_MAX_HW_QUEUES=8 ZKHIP_TUNE=msm_share=50 ZKHOST_EARLY_WIRING=1 $H --l 1 --n 20 --reps 3 --marks | tail -20
