# A/B: the root opens of d_open on the host (new) against a batch of their own on the device (prev = tools/debug/bin/hyperplonk_prev,
# built from the commit before); proofs after the first, min / median / max over 24; then digests and self-checks in every mode
NEW=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
PREV=./tools/debug/bin/hyperplonk_prev
export LD_LIBRARY_PATH=$PWD/scalable-collaborative-zksnark_amd:${LD_LIBRARY_PATH:-}   # (the copy's rpath is relative to host/bin)
for N in 12 16 20; do echo "#### n = $N"
for rep in 1 2; do
for w in PREV NEW; do echo -n "$w: "; ${!w} --l 1 --n $N --reps 25 | grep "End: Distributed" | awk '{print $4}' | tail -24 | sort -n | sed -n '1p;13p;24p' | tr '\n' ' '; echo; done
done; done
echo "#### digests and checks"
make -C tests/native -s fake_rccl/librccl.so.1
for a in "--l 1 --n 14" "--l 2 --n 12" "--l 1 --n 12 --which data-parallel" "--l 1 --n 10 --mode threads" "--l 2 --n 11 --mode threads" "--l 1 --n 12 --which dpermcheck" "--l 1 --n 12 --which cpermcheck --mode threads"; do
  for w in PREV NEW; do echo -n "$w | $a: "; ${!w} $a --reps 2 --digest --check | grep -E "sha256|check: party 0|FAILED" | awk '{print $3, $6}' | sort -u | tr '\n' ' '; echo; done
done
echo -n "NEW | rccl test double --l 1 --n 12: "; LD_LIBRARY_PATH=tests/native/fake_rccl:${LD_LIBRARY_PATH:-} $NEW --l 1 --n 12 --mode rccl --share-gpus --reps 2 --digest --check 2>&1 | grep -E "sha256|check: party|FAILED" | awk '{print $3, $6}' | sort | uniq -c | tr '\n' ' '; echo
echo -n "NEW | threads --l 1 --n 12: "; $NEW --l 1 --n 12 --mode threads --reps 2 --digest --check 2>&1 | grep -E "sha256|check: party|FAILED" | awk '{print $3, $6}' | sort | uniq -c | tr '\n' ' '; echo
echo -n "NEW | --tamper exit code: "; $NEW --l 1 --n 12 --reps 1 --tamper > /dev/null 2>&1; echo $?
