# A/B: with the padding-aware sort (struct RowReal) a class may hold items of very different lengths: one class per table width up to 2^k points
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
run() { echo -n "$1: "; shift; env "$@" $H --l 1 --n ${N:-20} --reps 8 --digest --check | grep -E "Distributed HyperPlonk|sha256|check:" | sort | uniq -c | sort -k3 | awk '{print $(NF-1)}' | head -7 | tr '\n' ' '; echo; }
for N in 20 16 24; do export N; echo "#### n = $N"
for rep in 1 2; do
run "min 14 (default)" ZKHIP_TUNE=msm_size_class_min=14
run "min 16          " ZKHIP_TUNE=msm_size_class_min=16
run "min 18          " ZKHIP_TUNE=msm_size_class_min=18
done; done
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_g2.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-extra --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['msm_phase_ms'])"
