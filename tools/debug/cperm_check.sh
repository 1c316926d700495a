python -m pytest tests/test_host_cpp.py -q -m gpu -x -k "checks_its_own or cpermcheck_pipelined" 2>&1 | tail -3
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
$H --l 1 --n 20 --which cpermcheck --reps 3 --check | tail -4
python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e-n24 --big 22 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('cpermcheck'))[:1800])"
