import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("scalable-collaborative-zksnark_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import zkhip, coracle as co, pyoracle as po
from helpers import jac_norm_to_affine, pt_ints, rand_fr, synthetic_bases
order = sys.argv[1] if len(sys.argv) > 1 else "lam,plain"
c = zkhip.Ctx(0)
c.comm_init(0, 1, c.comm_unique_id())
lam, coeff = 0x1234567890ABCDEF1122334455667788, 0x0FEDCBA987654321
srs_l, sc_l, lens, plain = [], [], [], []
for n, seed in ((300, 3), (1024, 4)):
    bases, _ = synthetic_bases(n, seed)
    sc = rand_fr(n, seed + 10)
    srs_l.append(c.srs_register(bases)); sc_l.append(c.to_device(sc)); lens.append(n)
    plain.append(pt_ints(co.msm_g1(bases, sc)))
co_limbs = np.array([[(coeff >> (64 * i)) & (2**64 - 1) for i in range(4)]], dtype=np.uint64)
lam_m = np.array(po.fr_to_mont_limbs(lam), dtype=np.uint64)
for what in order.split(","):
    if what == "lam":
        outs = c.d_msm(srs_l, sc_l, lens, co_limbs, lam_mont=lam_m)
        print("lam", [pt_ints(jac_norm_to_affine(outs[k])) == po.g1_mul(plain[k], lam * coeff % po.R_MOD) for k in range(2)], flush=True)
    elif what == "plain":
        outs = c.d_msm(srs_l, sc_l, lens, co_limbs)
        print("plain", [pt_ints(jac_norm_to_affine(outs[k])) == po.g1_mul(plain[k], coeff) for k in range(2)], flush=True)
    elif what == "msm":
        outs = c.msm_g1_batch(srs_l, sc_l, lens)
        print("msm", [pt_ints(jac_norm_to_affine(outs[k])) == plain[k] for k in range(2)], flush=True)
c.sync()
print("done", flush=True)
c.close()
