# A/B: cpermcheck with ONE MSM pass + one kernel batch (default) against the reference's call-by-call order (ZKHOST_CPERM_SERIAL=1 / ZKHIP_CPERM_SERIAL=1)
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for cfg in "--l 1 --n 20" "--l 1 --n 16" "--l 2 --n 20" "--l 8 --n 20" "--l 1 --n 22"; do echo "#### C++ host $cfg --which cpermcheck (leader)"
for rep in 1 2; do for v in 1 0; do echo -n "serial=$v: "; ZKHOST_CPERM_SERIAL=$v $H $cfg --which cpermcheck --reps 6 --digest | grep -E "proofs after|sha256" | sort -u | awk '{printf "%s ", $0} END {print ""}'; done; done; done
echo "#### self-checks and digests in the party modes"
make -C tests/native -s fake_rccl/librccl.so.1
for a in "--l 1 --n 12" "--l 2 --n 12" "--l 1 --n 12 --mode threads" "--l 2 --n 11 --mode threads"; do for v in 1 0; do echo -n "serial=$v | $a: "; ZKHOST_CPERM_SERIAL=$v $H $a --which cpermcheck --reps 2 --digest --check | grep -E "sha256|check: party 0|FAILED|Comm" | awk '{print $3, $6}' | sort -u | tr '\n' ' '; echo; done; done
echo -n "rccl test double --l 1 --n 12: "; LD_LIBRARY_PATH=tests/native/fake_rccl:${LD_LIBRARY_PATH:-} $H --l 1 --n 12 --which cpermcheck --mode rccl --share-gpus --reps 2 --digest --check 2>&1 | grep -E "sha256|check: party|FAILED|Comm" | awk '{print $3, $6}' | sort | uniq -c | tr '\n' ' '; echo
echo "#### Python host, n = 20, l = 1 (tools/cpermcheck_time.py)"
for v in 1 0 1 0; do echo -n "serial=$v: "; ZKHIP_CPERM_SERIAL=$v python tools/cpermcheck_time.py 20 3; done
python -m pytest tests/test_host_cpp.py tests/test_hyperplonk.py tests/test_large_l.py tests/test_gpu_comm.py -q -m gpu -x -k "perm or cperm or l2 or large or double" 2>&1 | tail -2
