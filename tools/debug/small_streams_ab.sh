H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for S in 0 1 0 1; do
  echo "== msm_small_streams=$S n=20"
  ZKHIP_TUNE=msm_small_streams=$S $H --l 1 --n 20 --reps 6 --digest --check | grep -E "Distributed HyperPlonk|sha256|check:" | sort | uniq -c | sort -k3 | awk '{print $(NF-1), $NF}' | head -8 | tr '\n' ' '; echo
done
for n in 12 16 24; do for S in 0 1; do echo "== msm_small_streams=$S n=$n"; ZKHIP_TUNE=msm_small_streams=$S $H --l 1 --n $n --reps 4 --check | grep -E "Distributed HyperPlonk|check:" | awk '{print $(NF-1)}' | tr '\n' ' '; echo; done; done
for S in 0 1; do echo "== threads n=20 S=$S"; ZKHIP_TUNE=msm_small_streams=$S $H --l 1 --n 20 --mode threads --reps 3 | grep -E "Distributed HyperPlonk" | awk '{print $(NF-1)}' | tr '\n' ' '; echo; done
for S in 0 1; do echo "== wiring batch (tools/proof_msm_mix.py) S=$S"; ZKHIP_TUNE=msm_small_streams=$S python tools/proof_msm_mix.py 20 5 | tail -1; done
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_g2.py -x -q -m gpu 2>&1 | tail -3
