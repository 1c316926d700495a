#!/bin/bash
# End-of-round soak on the GPU box: long randomised differential runs against the C oracle and repeated multi-party runs of the compiled
# host (races between party threads show up as a crash, a failed self-check or a changing digest):  tools/soak.sh <tag> [minutes per stress leg = 12]
set -u
T=${1:-soak}; M=${2:-12}; S=$((M * 60))
OUT=gpurun_out/${T}_soak.txt
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
{
echo "== randomised differential runs against the C oracle ($M min each; new seeds)"
STRESS_SEED=${SOAK_SEED:-7071} python tools/stress_msm.py $S | tail -1
STRESS_SEED=$((${SOAK_SEED:-7071} + 1)) STRESS_TABLE=1 python tools/stress_msm.py $S | tail -1
STRESS_SEED=$((${SOAK_SEED:-7071} + 2)) python tools/stress_sumcheck.py $S | tail -1
echo "== 300 consecutive n = 20 proofs, leader mode: one digest, steady time"
$H --l 1 --n 20 --reps 300 --digest --check > /tmp/soak_leader.txt 2>&1; echo "exit code $?"
grep -c "transcript sha256" /tmp/soak_leader.txt; grep "transcript sha256" /tmp/soak_leader.txt | sort | uniq -c; grep "End: Distributed HyperPlonk" /tmp/soak_leader.txt | awk '{print $(NF-1)}' | sort -n | awk '{a[NR]=$1} END {print "proof s: min", a[1], "median", a[int(NR/2)], "p99", a[int(NR*0.99)], "max", a[NR]}'; grep "check:" /tmp/soak_leader.txt
echo "== 25 runs of the 8-party protocol, n = 16, party threads, every party self-checked"
ok=0; for i in $(seq 25); do $H --l 1 --n 16 --mode threads --reps 2 --check --digest > /tmp/soak_t.txt 2>&1; rc=$?; c=$(grep -c " ok -- anchored" /tmp/soak_t.txt); d=$(grep "transcript sha256" /tmp/soak_t.txt | sort -u | wc -l); [ $rc -eq 0 ] && [ $c -eq 8 ] && [ $d -eq 1 ] && ok=$((ok+1)) || { echo "run $i: rc=$rc checks=$c digests=$d"; tail -3 /tmp/soak_t.txt; }; done; echo "$ok of 25 runs clean"
echo "== 15 runs of the 8-party protocol over the RCCL test double (zkhost::RcclNet, zk_d_msm, zk_allgather), n = 16"
make -C tests/native -s fake_rccl/librccl.so.1
ok=0; for i in $(seq 15); do LD_LIBRARY_PATH=tests/native/fake_rccl:${LD_LIBRARY_PATH:-} $H --l 1 --n 16 --mode rccl --share-gpus --reps 2 --check --digest > /tmp/soak_r.txt 2>&1; rc=$?; c=$(grep -c " ok -- anchored" /tmp/soak_r.txt); d=$(grep "transcript sha256" /tmp/soak_r.txt | sort -u | wc -l); [ $rc -eq 0 ] && [ $c -eq 8 ] && [ $d -eq 1 ] && ok=$((ok+1)) || { echo "run $i: rc=$rc checks=$c digests=$d"; tail -3 /tmp/soak_r.txt; }; done; echo "$ok of 15 runs clean"
echo "== 10 runs, 16 parties (l = 2), party threads, n = 14"
ok=0; for i in $(seq 10); do $H --l 2 --n 14 --mode threads --reps 2 --check > /tmp/soak_t2.txt 2>&1; rc=$?; c=$(grep -c " ok -- anchored" /tmp/soak_t2.txt); [ $rc -eq 0 ] && [ $c -eq 16 ] && ok=$((ok+1)) || { echo "run $i: rc=$rc checks=$c"; tail -3 /tmp/soak_t2.txt; }; done; echo "$ok of 10 runs clean"
echo "== cpermcheck (one MSM pass + one kernel batch): 10 runs with 8 party threads, 10 over the RCCL test double, n = 14, every party self-checked (5 commits / opens recomputed by single calls)"
for m in "--mode threads" "--mode rccl --share-gpus"; do ok=0; for i in $(seq 10); do LD_LIBRARY_PATH=tests/native/fake_rccl:${LD_LIBRARY_PATH:-} $H --l 1 --n 14 --which cpermcheck $m --reps 2 --check --digest > /tmp/soak_c.txt 2>&1; rc=$?; c=$(grep -c " ok -- anchored" /tmp/soak_c.txt); d=$(grep "transcript sha256" /tmp/soak_c.txt | sort -u | wc -l); [ $rc -eq 0 ] && [ $c -eq 8 ] && [ $d -eq 1 ] && ok=$((ok+1)) || { echo "run $i ($m): rc=$rc checks=$c digests=$d"; tail -3 /tmp/soak_c.txt; }; done; echo "$m: $ok of 10 runs clean"; done
} > $OUT 2>&1
cat $OUT
