# A/B of the class-sorted fix-up (ZKHIP_TUNE msm_fix_sorted = 0 / 1): MSM by size with phase timers, the bench headline, proofs; then parity suites
ph() { python - "$@" <<'PY'
import os, sys, time
sys.path.insert(0, "scalable-collaborative-zksnark_amd")
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
for lg in (18, 20, 22, 24):
    n = 1 << lg
    srs = ctx.srs_generate(123, 457, n); sc = ctx.to_device(random_fr(n, 5)); srs.precompute(0)
    outs = {}
    for v in (0, 1, 0, 1):
        ctx.dbg_tune("msm_fix_sorted", v)
        for _ in range(3): ctx.msm_g1(srs, sc, n)
        R = 20 if lg <= 20 else 5
        t0 = time.perf_counter()
        for _ in range(R): out = ctx.msm_g1(srs, sc, n)
        t = (time.perf_counter() - t0) / R
        outs[v] = out
        print(f"2^{lg} fix_sorted={v}: {t*1e3:8.3f} ms  phases {[round(float(x), 3) for x in ctx.msm_last_timing()]}", flush=True)
    assert (outs[0] == outs[1]).all()
    srs.free()
PY
}
ph
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for N in 20 24; do for rep in 1 2; do for v in 0 1; do echo -n "n = $N fix_sorted=$v: "; ZKHIP_TUNE=msm_fix_sorted=$v $H --l 1 --n $N --reps $((N == 20 ? 16 : 4)) | grep "proofs after"; done; done; done
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_g2.py -x -q -m gpu 2>&1 | tail -2
