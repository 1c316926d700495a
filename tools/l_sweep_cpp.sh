H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for l in 2 4 8 16 32; do
  for n in 16 20; do
    echo "== hyperplonk --l $l --n $n --reps 2 --check (leader echo, $((8*l)) parties)"
    timeout 600 $H --l $l --n $n --reps 2 --check 2>&1 | grep -E "Distributed HyperPlonk|check:|hyperplonk:|Comm" | tail -4
  done
done
echo "== cpermcheck l sweep n=16"
for l in 2 8 16 32; do echo "-- l=$l"; timeout 600 $H --l $l --n 16 --which cpermcheck --reps 2 --check 2>&1 | grep -E "Collaborative Permcheck|check:|hyperplonk:" | tail -3; done
echo "== threads l=2 n=16 (16 party threads), l=4 n=14 (32 party threads)"
timeout 900 $H --l 2 --n 16 --mode threads --reps 2 --check 2>&1 | grep -E "Distributed HyperPlonk|check: party 0 |hyperplonk:" | tail -3
timeout 900 $H --l 4 --n 14 --mode threads --reps 2 --check 2>&1 | grep -E "Distributed HyperPlonk|check: party 0 |hyperplonk:" | tail -3
