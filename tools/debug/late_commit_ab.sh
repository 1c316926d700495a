H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
run() { echo -n "$1: "; shift; env "$@" $H --l ${L:-1} --n ${N:-20} $MODE --reps 7 --check | grep -E "End: Distributed HyperPlonk|check: party 0" | sort | awk '{print $(NF-1)}' | head -6 | tr '\n' ' '; echo; }
for N in 17 18 19 22; do export N; echo "#### n = $N"; for rep in 1 2; do run "commit first" A=1; run "commit late " ZKHOST_LATE_COMMIT=1; done; done
export MODE="--mode threads"; for N in 16 20; do export N; echo "#### threads n = $N"; run "batch per call" ZKHOST_ONE_BATCH=0; run "ONE batch     " A=1; run "ONE + late    " ZKHOST_LATE_COMMIT=1; done
export MODE=""; export L=8; for N in 20; do export N; echo "#### l = 8 n = $N"; run "batch per call" ZKHOST_ONE_BATCH=0; run "ONE batch     " A=1; run "ONE + late    " ZKHOST_LATE_COMMIT=1; done
