python -m pytest tests/test_host_cpp.py tests/test_hyperplonk.py tests/test_gpu_e2e_fullsize.py tests/test_gpu_comm.py tests/test_gpu_bench.py tests/test_large_l.py -q -m gpu -x > gpurun_out/final_check_pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/final_check_pytest.txt | tail -3
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for n in 18 20; do echo -n "C++ n = $n: "; $H --l 1 --n $n --reps 13 | grep "proofs after"; done
for n in 18 20; do for v in 0 1; do echo -n "Python n = $n LATE_COMMIT=$v: "; ZKHIP_LATE_COMMIT=$v python tools/hyperplonk_bench.py --n $n --reps 6 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['timers_s']['Distributed HyperPlonk'], d.get('checks'))"; done; done
