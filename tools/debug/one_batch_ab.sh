# A/B of the proof schedule of the C++ host: the sumcheck-family kernels of steps 2-4 as ONE batch (ZKHOST_ONE_BATCH, default 1) x the commit pass
# started with the two long passes (ZKHOST_LATE_COMMIT)
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
run() { echo -n "$1: "; shift; env "$@" $H --l 1 --n ${N:-20} --reps 8 --digest --check | grep -E "Distributed HyperPlonk|sha256|check:" | sort | uniq -c | sort -k3 | awk '{print $(NF-1)}' | head -7 | tr '\n' ' '; echo; }
for N in 20 16 24 12; do export N; echo "#### n = $N"
for rep in 1 2; do
run "batch per call, commit first" ZKHOST_ONE_BATCH=0
run "ONE batch,      commit first" ZKHOST_ONE_BATCH=1
run "ONE batch,      commit late " ZKHOST_ONE_BATCH=1 ZKHOST_LATE_COMMIT=1
run "batch per call, commit late " ZKHOST_ONE_BATCH=0 ZKHOST_LATE_COMMIT=1
done; done
ZKHOST_ONE_BATCH=1 $H --l 1 --n 20 --reps 3 --marks | tail -16
echo "#### digests equal across schedules?"
for e in "ZKHOST_ONE_BATCH=0" "ZKHOST_ONE_BATCH=1" "ZKHOST_ONE_BATCH=1 ZKHOST_LATE_COMMIT=1"; do for a in "--l 1 --n 14" "--l 2 --n 12" "--l 1 --n 12 --which data-parallel" "--l 1 --n 10 --mode threads"; do echo -n "$e | $a: "; env $e $H $a --reps 1 --digest --check | grep -E "sha256|check: party 0" | awk '{print $3, $7}' | tr '\n' ' '; echo; done; done
