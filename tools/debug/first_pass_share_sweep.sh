H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for S in 0 100 85 75 60 50 0; do
  echo "== ZKHOST_FIRST_PASS_SHARE=$S"
  ZKHOST_FIRST_PASS_SHARE=$S $H --l 1 --n 20 --reps 6 --digest --check | grep -E "Distributed HyperPlonk|sha256|check:" | sort | uniq -c | sort -k3 | head -12
done
echo "== n=24"
for S in 0 75; do
  echo "== ZKHOST_FIRST_PASS_SHARE=$S"; ZKHOST_FIRST_PASS_SHARE=$S $H --l 1 --n 24 --reps 3 --check | grep -E "Distributed HyperPlonk|check:"
done
echo "== marks at 75"
ZKHOST_FIRST_PASS_SHARE=75 $H --l 1 --n 20 --reps 3 --marks | tail -24
echo "== msm regression (share 100 default, headline)"
python bench.py --no-extra --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['msm_phase_ms'])"
