"""GPU parity: Fr / Fq Montgomery arithmetic and the XYZZ group law vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from helpers import jac_norm_to_affine, pt_mont, rand_fr, synthetic_bases

pytestmark = pytest.mark.gpu

R_LIMBS = np.array([0xFFFFFFFF00000001, 0x53BDA402FFFE5BFE, 0x3339D80809A1D805, 0x73EDA753299D7D48], dtype=np.uint64)


def _edge_fr():
    e = np.zeros((8, 4), dtype=np.uint64)
    e[1, 0] = 1
    e[2] = R_LIMBS
    e[2, 0] -= np.uint64(1)  # r - 1
    e[3] = [0xFFFFFFFFFFFFFFFF, 0, 0, 0]
    e[4] = [0, 0, 0, 0x73EDA753299D7D48]
    e[5] = [0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F]  # R mod r = "one"
    e[6] = [0xFFFFFFFF, 0xFFFFFFFF00000000, 0xFFFFFFFF, 0x1]
    e[7] = R_LIMBS
    e[7, 0] -= np.uint64(2)
    return e


@pytest.mark.parametrize("n", [1, 63, 1000, 70001])
def test_fr_ops(ctx, co, n):
    a = np.concatenate([_edge_fr(), rand_fr(n, 1)])
    b = np.concatenate([_edge_fr()[::-1], rand_fr(n, 2)])
    m = len(a)
    da, db = ctx.to_device(a), ctx.to_device(b)
    for name, fn, ref in (("add", ctx.fr_add, co.fr_add), ("sub", ctx.fr_sub, co.fr_sub), ("mul", ctx.fr_mul, co.fr_mul)):
        got = fn(da, db, m).download((m, 4))
        assert (got == ref(a, b)).all(), name


def test_fr_axpb_and_div(ctx, co):
    n = 5000
    a, b = rand_fr(n, 3), rand_fr(n, 4)
    al, be = rand_fr(1, 5)[0], rand_fr(1, 6)[0]
    got = ctx.fr_axpb(ctx.to_device(a), ctx.to_device(b), al, be, n).download((n, 4))
    exp = co.fr_add(co.fr_add(a, co.fr_mul(np.tile(al, (n, 1)), b)), np.tile(be, (n, 1)))
    assert (got == exp).all()
    # division: num / den == num * den^-1 per element (dhyperplonk.rs:339)
    for m in (1, 15, 16, 17, 4097):
        num, den = rand_fr(m, 7), rand_fr(m, 8)
        got = ctx.fr_batch_div(ctx.to_device(num), ctx.to_device(den), m).download((m, 4))
        assert (got == co.fr_div(num, den)).all(), m
    den = rand_fr(100, 9)
    den[37] = 0
    with pytest.raises(ZeroDivisionError):
        ctx.fr_batch_div(ctx.to_device(rand_fr(100, 1)), ctx.to_device(den), 100)


def test_fq_ops(ctx, co):
    n = 20000
    raw = rand_fr((n * 6 + 3) // 4 * 2, 11).reshape(-1)
    a = raw[: n * 6].reshape(n, 6).copy()
    b = raw[n * 6 : 2 * n * 6].reshape(n, 6).copy()
    # canonicalise: clear the top 3 bits so values are < 2^381 then reduce by mapping through the oracle
    a[:, 5] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    b[:, 5] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    a[0] = 0
    b[1] = 0
    # edge rows: q - 1, 1, and the largest value below q whose 30-bit limbs are all ones where q allows
    import pyoracle as po

    def limbs6(v):
        return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]

    for row, v in enumerate((po.Q_MOD - 1, 1, (1 << 380) - 1, po.Q_MOD - (1 << 30), (1 << 360) - 1), start=2):
        a[row] = limbs6(v)
        b[row] = limbs6(po.Q_MOD - 1 - row)
    da, db = ctx.to_device(a), ctx.to_device(b)
    for op, ref in (("add", co.fq_add), ("sub", co.fq_sub), ("mul", co.fq_mul)):
        got = ctx.dbg_fq(op, da, db, n).download((n, 6))
        assert (got == ref(a, b)).all(), op
    got = ctx.dbg_fq("mul2add", da, db, n).download((n, 6))  # a*b + b*b under one reduction
    assert (got == co.fq_add(co.fq_mul(a, b), co.fq_mul(b, b))).all()
    # the same buffer as both operands takes the dedicated squaring path
    got = ctx.dbg_fq("mul", da, da, n).download((n, 6))
    assert (got == co.fq_mul(a, a)).all()
    got = ctx.dbg_fq("mul", db, db, n).download((n, 6))
    assert (got == co.fq_mul(b, b)).all()


def test_g1_group_law(ctx, co):
    import pyoracle as po

    n = 300
    P, _ = synthetic_bases(n, 21)
    Q, _ = synthetic_bases(n, 22)
    # adversarial rows: equal points (doubling), opposite points (cancellation), infinities
    Q[0] = P[0]
    neg = pt_mont(po.g1_neg((po.fq_from_mont_limbs(P[1][:6]), po.fq_from_mont_limbs(P[1][6:]))))
    Q[1] = neg
    P[2] = 0
    Q[3] = 0
    P[4] = 0
    Q[4] = 0
    dP, dQ = ctx.to_device(P), ctx.to_device(Q)
    from helpers import pt_ints

    def ref(mode, p, q):
        p, q = pt_ints(p), pt_ints(q)
        s = po.g1_add(p, q if mode != 3 else po.g1_neg(q))
        if mode == 1:
            s = po.g1_add(s, p)
        if mode == 2:
            s = po.g1_add(s, s)
        return pt_mont(s)

    for mode in (0, 1, 2, 3):
        got = ctx.dbg_g1_op(mode, dP, dQ, n)
        for i in range(n):
            assert (jac_norm_to_affine(got[i]) == ref(mode, P[i], Q[i])).all(), (mode, i)


def test_fr_apply_matrix_pss_maps(ctx, co):
    """zk_fr_apply_matrix == pack_from_public / unpack2 on chunks, for l = 1, 2, 4 (pss.rs:93-171)"""
    import pyoracle as po
    from zkhip.dist_primitive import _mont_matrix, _pack_chunks, _unpack2_many_device
    from zkhip.pss import PackedSharingParams

    for l in (1, 2, 4):
        pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
        rng = po.SplitMix64(30 + l)
        k = 37
        vals = rng.fr_vec(k * l - (1 if l > 1 else 0))  # a short last chunk is zero-padded
        vm = np.array([po.fr_to_mont_limbs(v) for v in vals], dtype=np.uint64)
        got = _pack_chunks(ctx, vm, pp)
        exp = po.transpose([opp.pack_from_public(vals[i : i + l]) for i in range(0, len(vals), l)])
        for p in range(pp.n):
            assert [po.fr_from_mont_limbs(x) for x in got[p]] == exp[p], (l, p)
        shares = [rng.fr_vec(k) for _ in range(pp.n)]
        sm = [np.array([po.fr_to_mont_limbs(v) for v in s], dtype=np.uint64) for s in shares]
        got = _unpack2_many_device(ctx, sm, pp)
        exp = [v for j in range(k) for v in opp.unpack2([shares[i][j] for i in range(pp.n)])]
        assert [po.fr_from_mont_limbs(x) for x in got] == exp, l


def test_fr_deinterleave(ctx):
    n = 3001
    t = rand_fr(2 * n, 77)
    ev, od = ctx.fr_deinterleave(ctx.to_device(t), n)
    assert (ev.download((n, 4)) == t[0::2]).all() and (od.download((n, 4)) == t[1::2]).all()


def test_quad_lane_additions(tmp_path):
    """the four-lane XYZZ additions of the reduction / fix-up passes == the single-lane formulas"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "quad_selftest")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-pragma-unroll-threshold=1000000",
                           "-o", exe, os.path.join(root, "tools", "quad_selftest.hip")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "add_quad mismatches 0, acc_quad mismatches 0" in out.stdout, out.stdout + out.stderr


def test_share_file_to_device_and_back(ctx, tmp_path):
    """examples/delegator.rs share files <-> resident Montgomery tables, conversions on the GPU"""
    import pyoracle as po
    from zkhip import serialize as ser

    rng = po.SplitMix64(77)
    xs = rng.fr_vec(1000) + [0, 1, po.R_MOD - 1]
    p = tmp_path / "worker_0"
    p.write_bytes(ser.fr_vec_serialize(xs))
    buf, n = ser.fr_file_to_device(ctx, str(p))
    assert n == len(xs)
    mont = buf.download((n, 4))
    assert [po.fr_from_mont_limbs(r) for r in mont] == xs
    ser.fr_device_to_file(ctx, buf, n, str(tmp_path / "out"))
    assert (tmp_path / "out").read_bytes() == p.read_bytes()


@pytest.mark.parametrize("l", [1, 2, 8, 16])
def test_pss_maps_by_transforms(ctx, l):
    """zk_fr_ntt_map (ifft -> resize -> fft, pss.rs:93-171) against the oracle's maps and the dense-matrix kernel"""
    import pyoracle as po
    from zkhip.dist_primitive import _mont_matrix
    from zkhip.pss import PackedSharingParams

    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    rng = po.SplitMix64(600 + l)
    k = 37
    to_m = lambda xs: np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)
    ints = lambda a: [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]
    # pack: k chunks of l secrets -> out[p*k + j]
    sec = rng.fr_vec(k * l)
    out = ints(ctx.fr_ntt_map(pp.ntt_tables("pack"), ctx.to_device(to_m(sec)), l, 1, k, 1, k).download((pp.n * k, 4)))
    for j in (0, 1, k - 1):
        assert [out[p * k + j] for p in range(pp.n)] == opp.pack_from_public(sec[j * l : (j + 1) * l])
    dense = ints(ctx.fr_apply_matrix(_mont_matrix([row[:l] for row in pp.pack_matrix]), ctx.to_device(to_m(sec)), l, 1, k, 1, k).download((pp.n * k, 4)))
    assert out == dense
    # unpack / unpack2: shares laid out party-major [n][k] (what an all-gather delivers) -> out[j*l + r]
    sh = rng.fr_vec(pp.n * k)
    d_sh = ctx.to_device(to_m(sh))
    for kind, fn, mat in (("unpack", opp.unpack, pp.unpack_matrix), ("unpack2", opp.unpack2, pp.unpack2_matrix)):
        out = ints(ctx.fr_ntt_map(pp.ntt_tables(kind), d_sh, 1, k, k, l, 1).download((k * l, 4)))
        for j in (0, 5, k - 1):
            assert out[j * l : (j + 1) * l] == fn([sh[i * k + j] for i in range(pp.n)]), (kind, j)
        assert out == ints(ctx.fr_apply_matrix(_mont_matrix(mat), d_sh, 1, k, k, l, 1).download((k * l, 4)))
