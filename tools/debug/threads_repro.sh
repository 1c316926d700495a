H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for i in 1 2 3 4 5 6 7 8 9 10; do echo -n "-- run $i: "; timeout 300 $H --l 1 --n 20 --mode threads --reps 2 --check > /tmp/o.txt 2>&1; echo "rc=$? $(grep -c 'ok -- anchored' /tmp/o.txt) parties ok $(grep -E 'free|corrupt|Abort|Segm' /tmp/o.txt | head -2)"; done
for i in 1 2 3 4 5 6; do echo -n "-- l=2 n=14 run $i: "; timeout 300 $H --l 2 --n 14 --mode threads --reps 2 --check > /tmp/o.txt 2>&1; echo "rc=$? $(grep -c 'ok -- anchored' /tmp/o.txt) parties ok $(grep -E 'free|corrupt|Abort|Segm' /tmp/o.txt | head -2)"; done
