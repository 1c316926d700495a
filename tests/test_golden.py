"""Committed golden vectors (tests/golden/vectors.json): checked by the C oracle on CPU and by the HIP path on GPU."""
import json
import os

import numpy as np
import pytest

import pyoracle as po
from helpers import jac_norm_to_affine, pt_ints, pt_mont

V = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
I = lambda xs: [int(x, 16) for x in xs]
to_m = lambda xs: np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)
ints = lambda a: [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]
P = lambda p: None if p is None else (int(p[0], 16), int(p[1], 16))


def test_golden_generator_is_reproducible():
    """vectors.json is exactly what make_golden.py produces from the oracle today"""
    g = V["sumcheck"]
    assert [[hex(a), hex(b)] for a, b in po.sumcheck(I(g["table"]), I(g["challenge"]))] == g["result"]
    k = V["acc_product_reference_kat"]
    assert po.acc_product(k["x"]) == (k["vx0"], k["vx1"], k["v1x"])


def test_c_oracle_against_golden(co):
    g = V["sumcheck"]
    assert [[hex(v) for v in ints(p)] for p in co.sumcheck(to_m(I(g["table"])), to_m(I(g["challenge"])))] == g["result"]
    g = V["sumcheck_product"]
    assert [[hex(v) for v in ints(p)] for p in co.sumcheck_product(to_m(I(g["f"])), to_m(I(g["g"])), to_m(I(g["challenge"])))] == g["result"]
    g = V["product_tree"]
    assert [hex(v) for v in ints(co.product_tree(to_m(I(g["x"]))))] == g["tree"]
    g = V["msm_g1"]
    bases = np.array([pt_mont(P(p)) for p in g["bases"]])
    assert pt_ints(co.msm_g1(bases, to_m(I(g["scalars"])))) == P(g["result"])


def test_host_pss_against_golden():
    from zkhip.pss import PackedSharingParams

    for l in (1, 2):
        g, pp = V[f"pss_l{l}"], PackedSharingParams(l)
        sec = I(g["secrets"])
        assert pp.pack_from_public(sec) == I(g["pack_from_public"])
        assert pp.pack_single(sec[0]) == I(g["pack_single_of_first"])
        assert pp.unpack2([v * v % po.R_MOD for v in pp.pack_from_public(sec)]) == I(g["unpack2_of_squares"])


@pytest.mark.gpu
def test_gpu_against_golden(ctx):
    g = V["sumcheck"]
    n = len(g["challenge"])
    pairs, last = ctx.sumcheck(ctx.to_device(to_m(I(g["table"]))), 1 << n, to_m(I(g["challenge"])))
    assert [[hex(v) for v in ints(p)] for p in pairs] == g["result"][:n] and hex(ints(last)[0]) == g["result"][n][1]
    g = V["sumcheck_product"]
    tr, lf, lg = ctx.sumcheck_product(ctx.to_device(to_m(I(g["f"]))), ctx.to_device(to_m(I(g["g"]))), 1 << n, to_m(I(g["challenge"])))
    assert [[hex(v) for v in ints(p)] for p in tr] == g["result"][:n]
    assert hex(ints(lf)[0] * ints(lg)[0] % po.R_MOD) == g["result"][n][1]
    g = V["fix_variable"]
    out = ctx.fold(ctx.to_device(to_m(I(g["table"]))), 32, to_m(I(g["points"]))).download((4, 4))
    assert [hex(v) for v in ints(out)] == g["result"]
    g = V["open_quotients"]
    q, val = ctx.open_rounds(ctx.to_device(to_m(I(g["table"]))), 32, to_m(I(g["point"])))
    assert [hex(v) for v in ints(q.download((31, 4)))] == [x for qi in g["q"] for x in qi] and hex(ints(val)[0]) == g["value"]
    g = V["product_tree"]
    assert [hex(v) for v in ints(ctx.product_tree(ctx.to_device(to_m(I(g["x"]))), 16).download((32, 4)))] == g["tree"]
    g = V["msm_g1"]
    bases = np.array([pt_mont(P(p)) for p in g["bases"]])
    got = ctx.msm_g1(ctx.srs_register(bases), ctx.to_device(to_m(I(g["scalars"]))), len(bases))
    assert pt_ints(jac_norm_to_affine(got)) == P(g["result"])


@pytest.mark.gpu
def test_gpu_d_msm_against_golden(ctx):
    """the l = 1, 8-party d_msm (dmsm.rs:9-43) with 8 party threads sharing GPU 0, vs the golden shares"""
    import zkhip
    from zkhip import dist_primitive as dp
    from zkhip.net import LocalTestNet
    from zkhip.pss import PackedSharingParams

    g = V["d_msm_l1"]
    pp = PackedSharingParams(1)

    def party(net):
        p = net.party_id
        c = zkhip.Ctx(0)
        bases = np.array([pt_mont(Q) for Q in po.g1_bases(4, g["seed_bases"][p])])
        out = dp.d_msm(c, [c.srs_register(bases)], [c.to_device(to_m(I(g["scalars"][p])))], [4], pp, net)
        c.close()
        return out

    res = LocalTestNet.simulate_network_round(8, party)
    assert [pt_ints(jac_norm_to_affine(res[p][0])) for p in range(8)] == [P(s) for s in g["shares"]]
