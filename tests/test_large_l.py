"""
The collaborative primitives from 64 parties up (l = 8, 16), where `zkhip.dist_primitive` applies the PSS maps as
TRANSFORMS (zk_fr_ntt_map: ifft -> resize -> fft, the reference's own form, pss.rs:93-171) instead of the dense public
matrix: `c_acc_product_and_share` (dacc_product.rs:66-292: unpack2 of the received blocks :94-104, pack_from_public of
every l-chunk :155-203), `d_unpack2_many` (unpack.rs:55-70), and through them `cpermcheck`.

CPU part (this file, not gpu): the branch through the oracle-backed stand-in, whose fr_ntt_map is a dense big-int
restatement of the kernel's contract, must equal the dense-matrix branch and the all-parties restatement of the oracle --
with REAL parties (64 threads at l = 8), not the echo net.
GPU part (`-m gpu`): the same calls through zkhip.Ctx (the kernels) must reproduce the oracle-backed digests at l = 8 and
l = 16 -- this is what executes dist_primitive's `pp.n >= NTT_FROM_N` branches on the GPU box.
"""
import hashlib

import numpy as np
import pytest

import pyoracle as po
from oracle_backend import OracleBackend
from zkhip import dist_primitive as dp
from zkhip.field import random_fr
from zkhip.net import LeaderEchoNet, LocalTestNet
from zkhip.pss import PackedSharingParams


def to_m(xs):
    return np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def ints(a):
    return [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]


def _digest(x) -> str:
    h = hashlib.sha256()

    def feed(o):
        if isinstance(o, np.ndarray):
            h.update(np.ascontiguousarray(o, dtype=np.uint64).tobytes())
        elif isinstance(o, (list, tuple)):
            for e in o:
                feed(e)
        elif o is not None:
            raise TypeError(type(o))

    feed(x)
    return h.hexdigest()


class _DenseOnly(OracleBackend):
    """the stand-in WITHOUT fr_ntt_map: dist_primitive then takes the dense-matrix branch at every party count"""

    fr_ntt_map = None

    def __getattribute__(self, name):
        if name == "fr_ntt_map":
            raise AttributeError(name)
        return super().__getattribute__(name)


def test_oracle_ntt_map_equals_the_pss_maps():
    """the stand-in's fr_ntt_map against the oracle's own pack / unpack / unpack2 (dense DFT vs the oracle's FFT code)"""
    l = 8
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    be = OracleBackend()
    rng = po.SplitMix64(4242)
    k = 3
    sec = rng.fr_vec(k * l)
    out = ints(be.fr_ntt_map(pp.ntt_tables("pack"), be.to_device(to_m(sec)), l, 1, k, 1, k).download((pp.n * k, 4)))
    for j in range(k):
        assert [out[p * k + j] for p in range(pp.n)] == opp.pack_from_public(sec[j * l : (j + 1) * l])
    sh = rng.fr_vec(pp.n * k)
    for kind, fn in (("unpack", opp.unpack), ("unpack2", opp.unpack2)):
        out = ints(be.fr_ntt_map(pp.ntt_tables(kind), be.to_device(to_m(sh)), 1, k, k, l, 1).download((k * l, 4)))
        for j in range(k):
            assert out[j * l : (j + 1) * l] == fn([sh[i * k + j] for i in range(pp.n)]), (kind, j)


def test_c_acc_product_and_share_64_real_parties_transform_branch():
    """l = 8, 64 party threads: the transform branch == the dense branch == the oracle's all-parties restatement"""
    l = 8
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    assert pp.n >= dp.NTT_FROM_N
    S = 16 * pp.n  # blocks of 16 shares per party: the local tree (16 l = 128 leaves) is taller than the N_p = 64 entries sent to the leader
    rng = po.SplitMix64(900)
    tabs = [[rng.fr_vec(S) for _ in range(pp.n)] for _ in range(5)]  # shares, masks, unmask0..2 per party

    def run(be):
        def party(net):
            p = net.party_id
            d = [be.to_device(to_m(t[p])) for t in tabs]
            res = dp.c_acc_product_and_share(be, d[0], d[1], d[2], d[3], d[4], S, pp, net)
            return [ints(buf.download((cnt, 4))) for buf, cnt in res]

        return LocalTestNet.simulate_network_round(pp.n, party)

    got, dense = run(OracleBackend()), run(_DenseOnly())
    exp = po.c_acc_product_and_share_all(*tabs, opp)
    for p in range(pp.n):
        assert got[p] == dense[p], p
        assert got[p] == [list(v) for v in exp[p]], p


def test_d_unpack2_many_transform_branch_cpu():
    l = 8
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    rng = po.SplitMix64(901)
    k = 5
    sh = [rng.fr_vec(k) for _ in range(pp.n)]
    be = OracleBackend()

    def party(net):
        return dp.d_unpack2_many(to_m(sh[net.party_id]), 3, pp, net, be=be)

    res = LocalTestNet.simulate_network_round(pp.n, party)
    want = [x for j in range(k) for x in opp.unpack2([sh[i][j] for i in range(pp.n)])]
    assert ints(res[3]) == want and all(len(res[p]) == 0 for p in range(pp.n) if p != 3)


# ---------------------------------------------------------------------------------------
# GPU: the kernels behind the same calls, l = 8 and l = 16, leader-echo net (party 0 of 64 / 128)
# ---------------------------------------------------------------------------------------
def _collab_suite(be, l, n):
    """every collaborative primitive of the path once, at packing factor l, on party 0 of the echo net -> nested results"""
    from zkhip.hyperplonk import PackedProvingParameters, cpermcheck

    pp = PackedSharingParams(l)
    net = LeaderEchoNet(pp.n)
    S = 4 * (1 << n) // l
    d = lambda s: be.to_device(random_fr(S, 1000 * l + s))
    levels = [be.srs_generate(3 + 2 * i, 5 + 2 * i, max(1, (1 << i) // l)) for i in range(n + 3)]  # new_single, dpoly_comm.rs:197-219
    ch = random_fr(n + 2 + 8, 77)
    out = {}
    out["pss2ss"] = dp.pss2ss(random_fr(1, 5)[0], pp, net)
    out["c_sumcheck_product"] = dp.c_sumcheck_product(be, d(1), d(2), S, ch, pp, net)
    out["c_open"] = list(dp.c_open(be, levels, d(3), S, ch, pp, net))
    out["d_unpack2_many"] = dp.d_unpack2_many(random_fr(9, 6), 0, pp, net, be=be)
    out["degree_reduce_many"] = dp.degree_reduce_many(random_fr(S // pp.n * 2, 7), pp, net, be=be)
    dev = dp.degree_reduce_many_device(be, d(4), S // pp.n * 2, pp, net)
    out["degree_reduce_many_device"] = dev.download((S // pp.n * 2, 4))
    res = dp.c_acc_product_and_share(be, d(8), d(9), d(10), d(11), d(12), S, pp, net)
    out["c_acc_product_and_share"] = [buf.download((cnt, 4)) for buf, cnt in res]
    pk = PackedProvingParameters.new(n, pp, be, seed=40 + l, chal_seed=4711, window_tables=False)
    out["cpermcheck"] = cpermcheck(n, pk, pp, be, net, seed=50 + l)[0]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("l", [8, 16])
def test_collaborative_primitives_at_64_and_128_parties_gpu(ctx, l):
    """dist_primitive's `pp.n >= NTT_FROM_N` branches (zk_fr_ntt_map inside c_acc_product_and_share, _pack_chunks_device,
    _unpack2_many_device) executed by the HIP library; digests must equal the oracle-backed run of the same calls"""
    n = 11 if l == 8 else 12  # tables of 4 * 2^n / l = 1024 shares: 16 N_p at l = 8, 8 N_p at l = 16 (there v(1,x) is left to the leader tree)
    calls = []
    orig = ctx.fr_ntt_map

    def spy(tables, *a, **kw):
        calls.append((tables["A"], tables["B"]))
        return orig(tables, *a, **kw)

    ctx.fr_ntt_map = spy
    try:
        got = _collab_suite(ctx, l, n)
    finally:
        del ctx.fr_ntt_map
    exp = _collab_suite(OracleBackend(), l, n)
    for key in exp:
        assert _digest(got[key]) == _digest(exp[key]), key
    # the transform branch really ran: unpack2 (A = 8l -> B = 4l) and pack (A = 2l -> B = 8l)
    assert (8 * l, 4 * l) in calls and (2 * l, 8 * l) in calls and len(calls) >= 2 + 2 * (1 + 3 + 3)
