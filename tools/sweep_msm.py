"""tuning sweep for the MSM pipeline (window bits / tile size / sort block range); run on the GPU box"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import zkhip
    from zkhip.field import random_fr
    log2n = int(sys.argv[2]); cs = [int(x) for x in sys.argv[3].split(",")]
    n = 1 << log2n
    ctx = zkhip.Ctx(0)
    srs = ctx.srs_generate(123, 457, n)
    sc = ctx.to_device(random_fr(n, 5))
    for c in cs:
        ctx.msm_set_window(c)
        ctx.msm_g1(srs, sc, n)
        ph = np.zeros(6); t0 = time.perf_counter(); R = 5
        for _ in range(R):
            ctx.msm_g1(srs, sc, n); ph += ctx.msm_last_timing()
        dt = (time.perf_counter() - t0) / R
        print(f"  log2n={log2n} c={c} tune={os.environ.get('ZKHIP_TUNE','-')}: {dt*1e3:7.3f} ms  sort={ph[0]/R:.3f} acc={ph[1]/R:.3f} fix={ph[2]/R:.3f} red={ph[3]/R:.3f} host={ph[4]/R:.3f}", flush=True)
else:
    for log2n, cs in ((20, "13,14,15,16"),):
        for T in ("32", "64"):
            subprocess.run([sys.executable, __file__, "child", str(log2n), cs], env=dict(os.environ, ZKHIP_TUNE="msm_tile=" + T))
    for log2n, cs in ((18, "12,13,14,15"), (16, "10,11,12,13"), (14, "8,9,10,11"), (12, "6,7,8,9"), (10, "5,6,7"), (22, "15,16")):
        subprocess.run([sys.executable, __file__, "child", str(log2n), cs], env=dict(os.environ, ZKHIP_TUNE="msm_tile=32"))
