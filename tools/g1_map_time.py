"""
VERDICT r05 item 8: what do the PSS maps ON POINTS of d_msm's leader closure (dmsm.rs:30-39: unpack2 -> sum -> pack_from_public of
[sum; l], for a batch of n + 2 = 22 commitments at n = 20) cost at l = 8, 16, 32 (64 / 128 / 256 parties)?  Three forms:
  (a) the literal maps through zk_g1_apply_matrix (dense O((8l)^2) -- what a binding that keeps the reference's closure would call):
      unpack2 of 22 vectors of 8l points, then pack_from_public of 22 vectors of l points into 8l shares each;
  (b) what this repository's hosts run (dist_primitive.d_msm): lambda_p folded into the party's scalars before its MSM, so the
      exchange is an all-gather and, per item, 8l - 1 point additions + ONE scalar multiplication by c_p on the host
      (zk_g1_lincomb_batch) -- for the party's OWN share only, every party in parallel;
  (c) the leader-mode proof of the same l (host/bin/hyperplonk --l L --n 20), for the ratio.
python tools/g1_map_time.py [l ...]
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
import numpy as np, zkhip
from zkhip.field import int_to_limbs
from zkhip.pss import PackedSharingParams

ctx = zkhip.Ctx(0)
ITEMS = 22
canon = lambda vals: np.array([int_to_limbs(int(v), 4) for v in vals], dtype=np.uint64)
for l in [int(x) for x in sys.argv[1:]] or [8, 16, 32]:
    pp = PackedSharingParams(l)
    N = pp.n
    srs = ctx.srs_generate(77 + l, 991, ITEMS * N)
    pts = srs.download()  # [ITEMS * N, 12] affine, reference form
    d_in = ctx.to_device(pts)
    m_u = canon([v for row in pp.unpack2_matrix for v in row]).reshape(l, N, 4)
    m_p = canon([pp.pack_matrix[p][j] for p in range(N) for j in range(l)]).reshape(N, l, 4)
    def literal():
        sec = ctx.g1_apply_matrix(m_u, d_in, N, 1, ITEMS, l, 1)          # 22 x (8l points -> l points)
        return ctx.g1_apply_matrix(m_p, sec, l, 1, ITEMS, 1, ITEMS)      # 22 x (l points -> 8l shares)
    literal(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): literal().download((N * ITEMS, 12))
    t_lit = (time.perf_counter() - t0) / 3
    # (b): one party's combination of the gathered results: [ITEMS, N, 18] normalised Jacobian -> [ITEMS, 18], then x c_p
    jac = np.zeros((ITEMS, N, 18), dtype=np.uint64)
    one = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d, 0x15f65ec3fa80e493], dtype=np.uint64)
    jac[:, :, :12] = pts.reshape(ITEMS, N, 12); jac[:, :, 12:] = one
    ones = np.tile(int_to_limbs(1, 4), (N, 1))
    c_p = canon([sum(pp.pack_matrix[0][j] for j in range(l)) % zkhip.field.R_MOD])
    def folded():
        sums = ctx.g1_lincomb_batch(jac, ones)
        return ctx.g1_lincomb_batch(sums.reshape(ITEMS, 1, 18), c_p)
    folded()
    t0 = time.perf_counter()
    for _ in range(3): folded()
    t_fold = (time.perf_counter() - t0) / 3
    exe = os.path.join(ROOT, "scalable-collaborative-zksnark_amd", "host", "bin", "hyperplonk")
    r = subprocess.run([exe, "--l", str(l), "--n", "20", "--reps", "7"], capture_output=True, text=True)
    med = [ln for ln in r.stdout.splitlines() if ln.startswith("proofs after")]
    proof = float(med[0].split("median")[1].split()[0]) if med else float("nan")
    print(f"l = {l:2d} ({N:3d} parties), batch of {ITEMS}: literal maps (zk_g1_apply_matrix x 2) {t_lit*1e3:8.2f} ms | hosts' folded form, one party's share {t_fold*1e3:7.2f} ms | "
          f"leader-mode proof n = 20: {proof*1e3:7.2f} ms | folded / proof = {t_fold/proof*100:5.1f} %  (one such batch per proof step: 3 d_msm exchanges per proof)", flush=True)
    srs.free()
