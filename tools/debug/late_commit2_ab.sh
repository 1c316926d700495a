# A/B: where the commit pass of a proof is started (ZKHOST_LATE_COMMIT: 0 = at step 1, 1 = with the two long passes, 2 = right after the kernel batch, before the hand-offs)
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for cfg in "--l 1 --n 20" "--l 1 --n 16" "--l 1 --n 12" "--l 1 --n 18" "--l 1 --n 22" "--l 8 --n 20" "--l 1 --n 24"; do echo "#### $cfg"
for rep in 1 2; do for v in 0 1 2; do echo -n "late_commit=$v: "; R=25; case "$cfg" in *"n 24"*) R=4;; *"n 22"*) R=9;; esac; ZKHOST_LATE_COMMIT=$v $H $cfg --reps $R | grep "proofs after"; done; done; done
for v in 0 2; do echo -n "digest late_commit=$v: "; ZKHOST_LATE_COMMIT=$v $H --l 1 --n 14 --reps 2 --digest --check | grep -E "sha256|check:" | awk '{print $3, $6}' | sort -u | tr '\n' ' '; echo; ZKHOST_LATE_COMMIT=$v $H --l 1 --n 12 --mode threads --reps 2 --digest --check | grep -E "sha256|check: party 0" | awk '{print $3, $6}' | sort -u | tr '\n' ' '; echo; done
ZKHOST_LATE_COMMIT=2 $H --l 1 --n 20 --reps 3 --marks | tail -14
