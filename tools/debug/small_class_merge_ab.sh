# A/B of the library's default for the short SRS levels: two common table widths (12 / 14 bits) + one size class up to 2^14 points
# (the default) against a width per size and a class per size (ZKHIP_TUNE=msm_small_table_widths=0,msm_size_class_min=0)
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
OLD=msm_small_table_widths=0,msm_size_class_min=0
run() { echo -n "$1: "; shift; env "$@" $H --l 1 --n ${N:-20} --reps 8 --digest --check | grep -E "Distributed HyperPlonk|sha256|check:" | sort | uniq -c | sort -k3 | awk '{print $(NF-1)}' | head -7 | tr '\n' ' '; echo; }
for N in 20 16 12 24; do export N; echo "#### n = $N"
for rep in 1 2; do
run "per-size (old)" ZKHIP_TUNE=$OLD
run "merged (new)  " A=1
done; done
echo "#### single MSMs (tools/msm_time.py), old then new"
ZKHIP_TUNE=$OLD python tools/msm_time.py 6 8 10 12 14 16
python tools/msm_time.py 6 8 10 12 14 16
echo "#### 8 party threads n = 16 / 20, old then new"
for N in 16 20; do for T in $OLD msm_debug=0; do ZKHIP_TUNE=$T $H --l 1 --n $N --mode threads --reps 3 | grep "Distributed HyperPlonk" | awk '{print $(NF-1)}' | tr '\n' ' '; echo "($T)"; done; done
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_srs.py -x -q -m gpu 2>&1 | tail -2
