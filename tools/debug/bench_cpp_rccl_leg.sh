# the N = 8 bench line with its e2e_cpp_rccl leg on a ONE-GPU box: party threads for the bench's own ranks, the test double of librccl for the compiled host
make -C tests/native -s fake_rccl/librccl.so.1
export HSA_ENABLE_IPC_MODE_LEGACY=0 ZK_BENCH_DEADLINE_S=600
LD_LIBRARY_PATH=$PWD/tests/native/fake_rccl:${LD_LIBRARY_PATH:-} ZK_BENCH_BACKEND=local ZK_BENCH_CPP_RCCL=share python bench.py --gpus 8 --party-threads --steps 2 --warmup 1 --no-cpu --e2e-n ${1:-12} 2>gpurun_out/bench_cpp_rccl_leg.err | tail -1 > gpurun_out/bench_cpp_rccl_leg.json
python - <<'PY'
import json
b = json.loads(open("gpurun_out/bench_cpp_rccl_leg.json").read())
print("value", b.get("value"), "n_gpus", b.get("n_gpus"))
e = b.get("e2e") or {}
print("python e2e:", e.get("transcript_checks"), e.get("transcript_sha256"), (e.get("timers_s") or {}).get("Distributed HyperPlonk"))
print(json.dumps(b.get("e2e_cpp_rccl"), indent=1)[:3000])
PY
tail -5 gpurun_out/bench_cpp_rccl_leg.err
