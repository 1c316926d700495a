"""Strong-scaling shards (zkhip/sharding.py): the sharded primitives equal the monolithic ones bit for bit."""
import numpy as np
import pytest

import pyoracle as po
from helpers import pt_ints, pt_mont
from oracle_backend import OracleBackend, OracleSrs
from zkhip import sharding as sh
from zkhip.net import LocalTestNet


def to_m(xs):
    return np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def ints(a):
    return [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]


@pytest.mark.parametrize("G", [2, 4, 8])
def test_sharded_sumchecks_equal_monolithic(G):
    rng = po.SplitMix64(900 + G)
    n = 6
    f, g, ch = rng.fr_vec(1 << n), rng.fr_vec(1 << n), rng.fr_vec(n)
    be = OracleBackend()
    F, Gm = to_m(f), to_m(g)

    def party(net):
        r = net.party_id
        lf, lg = be.to_device(sh.cyclic_shard(F, r, G)), be.to_device(sh.cyclic_shard(Gm, r, G))
        a = sh.sharded_sumcheck(be, lf, (1 << n) // G, to_m(ch), net)
        b = sh.sharded_sumcheck_product(be, lf, lg, (1 << n) // G, to_m(ch), net)
        return a, b

    res = LocalTestNet.simulate_network_round(G, party)
    for r in range(G):
        assert [tuple(ints(t)) for t in res[r][0]] == po.sumcheck(f, ch)
        assert [tuple(ints(t)) for t in res[r][1]] == po.sumcheck_product(f, g, ch)


@pytest.mark.parametrize("G", [2, 8])
def test_sharded_msm_equals_monolithic(G):
    rng = po.SplitMix64(950 + G)
    n = 48
    pts, sc = po.g1_bases(n, 17), rng.fr_vec(n)
    be = OracleBackend()
    per = n // G

    def party(net):
        r = net.party_id
        srs = OracleSrs(np.array([pt_mont(P) for P in pts[r * per : (r + 1) * per]]))
        return sh.sharded_msm(be, srs, be.to_device(to_m(sc[r * per : (r + 1) * per])), per, net)

    res = LocalTestNet.simulate_network_round(G, party)
    exp = po.g1_msm(pts, sc)
    assert all(pt_ints(r[:12]) == exp for r in res)


@pytest.mark.gpu
def test_sharded_on_gpu_threads(ctx):
    """4 shards as 4 contexts on GPU 0: sharded sumcheck_product / MSM equal the single-context results"""
    import zkhip
    from helpers import rand_fr, synthetic_bases

    G, n = 4, 14
    f, g, ch = rand_fr(1 << n, 1), rand_fr(1 << n, 2), rand_fr(n, 3)
    bases, _ = synthetic_bases(1 << 12, 5)
    sc = rand_fr(1 << 12, 6)
    tr, lf, lg = ctx.sumcheck_product(ctx.to_device(f), ctx.to_device(g), 1 << n, ch)
    mono_msm = ctx.msm_g1(ctx.srs_register(bases), ctx.to_device(sc), 1 << 12)
    per = (1 << 12) // G

    def party(net):
        r = net.party_id
        c = zkhip.Ctx(0)
        a = sh.sharded_sumcheck_product(c, c.to_device(sh.cyclic_shard(f, r, G)), c.to_device(sh.cyclic_shard(g, r, G)), (1 << n) // G, ch, net)
        b = sh.sharded_msm(c, c.srs_register(bases[r * per : (r + 1) * per]), c.to_device(sc[r * per : (r + 1) * per]), per, net)
        c.close()
        return a, b

    res = LocalTestNet.simulate_network_round(G, party)
    for r in range(G):
        assert (res[r][0][:n] == tr).all()
        assert (res[r][1] == mono_msm).all()
