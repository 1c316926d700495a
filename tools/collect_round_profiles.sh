set -u
T=${1:-r03g}
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tools/profile_bench.sh $T > gpurun_out/${T}_profile.log 2>&1
tools/profile_valu.sh $T >> gpurun_out/${T}_profile.log 2>&1
tools/sc_prof.sh $T 20 >> gpurun_out/${T}_profile.log 2>&1
tools/sc_pmc.sh $T 24 >> gpurun_out/${T}_profile.log 2>&1
tools/sc_pmc.sh $T 26 >> gpurun_out/${T}_profile.log 2>&1
python tools/sc_time.py 12 16 18 20 22 24 26 > gpurun_out/${T}_sc_sizes.txt 2>&1
python tools/msm_time.py 12 14 16 18 20 22 24 > gpurun_out/${T}_msm_sizes.txt 2>&1
tools/trace_one_msm_table.sh 20 > gpurun_out/${T}_msm_2e20_dispatch_timeline.txt 2>&1
# the sort phase (round 6): kernel timelines at 2^22 / 2^24, counter traffic of its kernels beside the round-5 kernels, A/B on one box
for n in 22 24; do tools/trace_one_msm_table.sh $n 2>&1 | head -14 > gpurun_out/${T}_sort_trace_2p$n.txt; done
rm -f gpurun_out/${T}_sort_phase_pmc.csv; tools/profile_sort.sh $T 24 > /dev/null 2>&1
python tools/sort_ab.py 20 22 24 > gpurun_out/${T}_sort_ab.txt 2>&1
for n in 12 16 20 24; do python tools/hyperplonk_bench.py --n $n --reps 3 | tail -1; done > gpurun_out/${T}_e2e.jsonl 2>&1
for n in 12 16 20; do python tools/hyperplonk_bench.py --n $n --party-threads | tail -1; done > gpurun_out/${T}_e2e_party_threads.jsonl 2>&1
# the same proofs from the compiled C++ host (host/examples/hyperplonk.cpp): leader mode n = 12 .. 24, 8 party threads on the one GPU, cpermcheck
{ B=scalable-collaborative-zksnark_amd/host/bin/hyperplonk
  # (--check: every run verified by the compiled host itself, zkhost/verify.hpp; --serial-rep: per-step timers with every pass inside its step)
  for n in 12 16 20 24; do echo "== hyperplonk --l 1 --n $n --reps 4 --check --serial-rep (leader)"; $B --l 1 --n $n --reps 4 --check --serial-rep | tail -15; done
  # (back-to-back proofs without the digests of --check between them: the steady-state figure)
  for n in 12 16 20 24; do echo -n "== hyperplonk --l 1 --n $n --reps $((n < 24 ? 25 : 5)) (leader): "; $B --l 1 --n $n --reps $((n < 24 ? 25 : 5)) | grep "proofs after"; done
  for l in 2 4 8 16 32; do echo -n "== hyperplonk --l $l --n 20 --reps 13 (leader): "; $B --l $l --n 20 --reps 13 | grep "proofs after"; done
  # first proof of a process with and without the arena plan of an earlier one (zk_arena_plan_export / _import), and the resident footprint
  for n in 20 24; do rm -f /tmp/plan$n.bin; for pass in 1 2; do echo "== hyperplonk --l 1 --n $n --reps 4 --arena-plan /tmp/plan$n.bin (run $pass)"; $B --l 1 --n $n --reps 4 --arena-plan /tmp/plan$n.bin | grep -E "arena plan|Distributed|proofs after|HBM after"; done; done
  for n in 12 16 20; do echo "== hyperplonk --l 1 --n $n --mode threads --reps 3 --check (8 party threads, ONE GPU does the work of eight)"; $B --l 1 --n $n --mode threads --reps 3 --check | tail -15; done
  echo "== hyperplonk --l 2 --n 16 --which cpermcheck --reps 3 --check (leader)"; $B --l 2 --n 16 --which cpermcheck --reps 3 --check | tail -4
  make -C tests/native -s fake_rccl/librccl.so.1
  echo "== hyperplonk --l 1 --n 20 --mode rccl --share-gpus --reps 2 --check over the TEST DOUBLE of librccl (tests/native/fake_rccl.cpp: 8 ranks = 8 threads sharing the GPU; no wire)"
  LD_LIBRARY_PATH=tests/native/fake_rccl:${LD_LIBRARY_PATH:-} $B --l 1 --n 20 --mode rccl --share-gpus --reps 2 --check 2>&1 | grep -E "fake_rccl|Distributed HyperPlonk|Comm|check:"
  echo "== hyperplonk --l 1 --n 12 --reps 1 --tamper (must fail: exit code 3)"; $B --l 1 --n 12 --reps 1 --tamper > /tmp/tamper.out 2>&1; RC=$?; tail -2 /tmp/tamper.out; echo "exit code $RC"
} > gpurun_out/${T}_e2e_cpp_host.txt 2>&1
tools/profile_timeline.sh $T 20 5 > /dev/null 2>&1
tools/sc_valu.sh $T product 20 > /dev/null 2>&1
python tools/g2_time.py 17 0 > gpurun_out/${T}_g2.txt 2>&1
python tools/cpermcheck_time.py 20 3 > gpurun_out/${T}_cpermcheck.jsonl 2>&1
python tools/sc_batch_time.py 18 > gpurun_out/${T}_sc_batch.txt 2>&1
# N > 1 exactly as the driver launches it (`python bench.py --gpus N`, no launcher): functional runs on the one GPU of the box
export HSA_ENABLE_IPC_MODE_LEGACY=0 ZK_BENCH_DEADLINE_S=600
# (8 PROCESSES sharing one GPU: the n = 20 leg with its window tables and job-lane arenas, ~30 GB per rank, does not fit eight times: n = 16 there)
for N in 2 8; do ZK_BENCH_BACKEND=gloo python bench.py --gpus $N --steps 3 --warmup 1 --no-cpu --e2e-n $((N == 8 ? 16 : 20)) 2>gpurun_out/${T}_gloo$N.err | tail -1; done > gpurun_out/${T}_bench_gloo_ranks_sharing_one_gpu.jsonl
sleep 5  # (the eight gloo ranks above release the GPU's memory when they exit)
ZK_BENCH_BACKEND=local python bench.py --gpus 8 --party-threads --steps 3 --warmup 1 --no-cpu --e2e-n 16 2>gpurun_out/${T}_threads8.err | tail -1 > gpurun_out/${T}_bench_party_threads_sharing_one_gpu.jsonl
python bench.py --gpus 2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/${T}_bench_gpus2_on_a_one_gpu_box_error_line.json
# randomised differential runs against the C oracle, new seeds
{ STRESS_SEED=6061 python tools/stress_msm.py 120 | tail -1; STRESS_SEED=6062 STRESS_TABLE=1 python tools/stress_msm.py 120 | tail -1; STRESS_SEED=6063 python tools/stress_sumcheck.py 180 | tail -1; } > gpurun_out/${T}_stress.txt 2>&1
tail -3 gpurun_out/${T}_e2e.jsonl; cat gpurun_out/${T}_g2.txt; cat gpurun_out/${T}_stress.txt
