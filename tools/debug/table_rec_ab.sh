# A/B: G1 window-table records of 128 B (one per 128-B line) against the packed 96 B (ZKHIP_TUNE / zk_dbg_tune srs_table_rec, read when a table is built)
python - <<'PY'
import sys, time
sys.path.insert(0, "scalable-collaborative-zksnark_amd")
import numpy as np, zkhip
from zkhip.field import random_fr
ctx = zkhip.Ctx(0)
for lg in (16, 20, 22, 24):
    n = 1 << lg
    srs = ctx.srs_generate(123, 457, n); sc = ctx.to_device(random_fr(n, 5))
    outs = {}
    for rec in (96, 128, 96, 128):
        ctx.dbg_tune("srs_table_rec", rec)
        srs.precompute(0)
        for _ in range(3): ctx.msm_g1(srs, sc, n)
        R = 20 if lg <= 20 else 5
        t0 = time.perf_counter()
        for _ in range(R): out = ctx.msm_g1(srs, sc, n)
        t = (time.perf_counter() - t0) / R
        outs[rec] = out
        print(f"2^{lg} rec={rec}: {t*1e3:8.3f} ms  {n/t/1e8:.3f}e8/s  phases {[round(float(x), 3) for x in ctx.msm_last_timing()]}", flush=True)
    assert (outs[96] == outs[128]).all()
    srs.free()
PY
H=./scalable-collaborative-zksnark_amd/host/bin/hyperplonk
for N in 20 24; do for v in 96 128 96 128; do echo -n "n = $N rec=$v: "; ZKHIP_TUNE=srs_table_rec=$v $H --l 1 --n $N --reps $((N == 20 ? 13 : 4)) | grep -E "proofs after|HBM after" | tr '\n' ' '; echo; done; done
ZKHIP_TUNE=srs_table_rec=128 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_srs.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
