"""
Host / exchange logic of zkhip.dist_primitive, run WITHOUT a GPU: the compute backend is the
oracle-backed stand-in (tests/oracle_backend.py), the parties are threads (LocalTestNet, like the
reference's LocalTestNet::simulate_network_round) -- results must equal the all-parties-in-one
restatement in oracle/pyoracle.py (which follows the reference's star protocol literally).
"""
import numpy as np
import pytest

import pyoracle as po
from helpers import pt_ints, pt_mont
from oracle_backend import OracleBackend, OracleSrs
from zkhip import dist_primitive as dp
from zkhip.net import LeaderEchoNet, LocalTestNet
from zkhip.pss import PackedSharingParams


def to_m(xs):
    return np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def ints(a):
    return [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]


def jac_pt(j):
    return None if not j[12:].any() else pt_ints(j[:12])


@pytest.mark.parametrize("l", [1, 2])
def test_pss_matrices_match_oracle(l):
    a, b = PackedSharingParams(l), po.PackedSharingParams(l)
    assert a.pack_matrix == b.pack_matrix() and a.unpack_matrix == b.unpack_matrix() and a.unpack2_matrix == b.unpack2_matrix()
    assert a.pack_single(987654321) == b.pack_single(987654321)


@pytest.mark.parametrize("l", [1, 2])
def test_pss2ss_and_c_sumchecks(l):
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    rng = po.SplitMix64(70 + l)
    n = 3
    sf = [rng.fr_vec(1 << n) for _ in range(pp.n)]
    sg = [rng.fr_vec(1 << n) for _ in range(pp.n)]
    ch = rng.fr_vec(n + 2)
    be = OracleBackend()

    def party(net):
        p = net.party_id
        a = dp.c_sumcheck(be, be.to_device(to_m(sf[p])), 1 << n, to_m(ch), pp, net)
        b = dp.c_sumcheck_product(be, be.to_device(to_m(sf[p])), be.to_device(to_m(sg[p])), 1 << n, to_m(ch), pp, net)
        return a, b

    res = LocalTestNet.simulate_network_round(pp.n, party)
    exp_a = po.c_sumcheck_all(sf, ch, opp)
    exp_b = po.c_sumcheck_product_all(sf, sg, ch, opp)
    for p in range(pp.n):
        assert [tuple(ints(t)) for t in res[p][0]] == exp_a[p]
        assert [tuple(ints(t)) for t in res[p][1]] == exp_b[p]


def test_d_sumchecks_and_acc_product():
    rng = po.SplitMix64(81)
    np_, n = 8, 3
    pf = [rng.fr_vec(1 << n) for _ in range(np_)]
    pg = [rng.fr_vec(1 << n) for _ in range(np_)]
    ch = rng.fr_vec(n + 3)
    be = OracleBackend()

    def party(net):
        p = net.party_id
        a = dp.d_sumcheck(be, be.to_device(to_m(pf[p])), 1 << n, to_m(ch), net)
        b = dp.d_sumcheck_product(be, be.to_device(to_m(pf[p])), be.to_device(to_m(pg[p])), 1 << n, to_m(ch), net)
        tree, top = dp.d_acc_product(be, be.to_device(to_m(pf[p])), 1 << n, net)
        return a, b, tree.download((2 << n, 4)), top

    res = LocalTestNet.simulate_network_round(np_, party)
    assert [tuple(ints(t)) for t in res[0][0]] == po.d_sumcheck_all(pf, ch)
    assert [tuple(ints(t)) for t in res[0][1]] == po.d_sumcheck_product_all(pf, pg, ch)
    subtrees, leader = po.d_acc_product_all(pf)
    for p in range(np_):
        assert ints(res[p][2]) == subtrees[p]
        if p:
            assert len(res[p][0]) == 0 and len(res[p][1]) == 0 and res[p][3] is None  # workers: vec![] / None
    assert ints(res[0][3]) == leader


@pytest.mark.parametrize("l", [1, 2])
def test_d_msm_over_threads(l):
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    rng = po.SplitMix64(90 + l)
    sizes = [8, 4]  # a batch of two MSMs, like c_open's shrinking batch
    bases = [[po.g1_bases(m, 100 * p + m) for m in sizes] for p in range(pp.n)]
    scal = [[rng.fr_vec(m) for m in sizes] for p in range(pp.n)]
    be = OracleBackend()

    def party(net):
        p = net.party_id
        srs = [OracleSrs(np.array([pt_mont(P) for P in b])) for b in bases[p]]
        sc = [be.to_device(to_m(s)) for s in scal[p]]
        return dp.d_msm(be, srs, sc, sizes, pp, net)

    res = LocalTestNet.simulate_network_round(pp.n, party)
    exp = po.d_msm_all(bases, scal, opp)
    for p in range(pp.n):
        assert [jac_pt(j) for j in res[p]] == exp[p]


def test_leader_echo_mode_matches_appendix_b():
    """config 1 plumbing (no-`comm` fake): d_msm returns (4/7) * MSM for party 0 at l = 1"""
    pp = PackedSharingParams(1)
    be = OracleBackend()
    rng = po.SplitMix64(3)
    pts, sc = po.g1_bases(6, 9), rng.fr_vec(6)
    net = LeaderEchoNet(8)
    out = dp.d_msm(be, [OracleSrs(np.array([pt_mont(P) for P in pts]))], [be.to_device(to_m(sc))], [6], pp, net)
    c0 = 4 * pow(7, -1, po.R_MOD) % po.R_MOD
    assert jac_pt(out[0]) == po.g1_mul(po.g1_msm(pts, sc), c0)
    up, down = net.upload, net.download
    assert up == down == 7 * 144  # one 144-byte point "sent to 7 others"


def test_commit_open_and_d_commit():
    be = OracleBackend()
    n, np_ = 3, 8
    levels_pts = [po.g1_bases(1 << k, 300 + k) for k in range(n + 1)]
    levels = [OracleSrs(np.array([pt_mont(P) for P in lv])) for lv in levels_pts]
    rng = po.SplitMix64(5)
    chunks = [rng.fr_vec(1 << n) for _ in range(np_)]
    point = rng.fr_vec(n)
    v, proofs = dp.open_(be, levels, be.to_device(to_m(chunks[0])), 1 << n, to_m(point))
    ev, eproofs = po.open_(levels_pts, chunks[0], point)
    assert ints(v) == [ev] and [jac_pt(p) for p in proofs] == eproofs

    def party(net):
        return dp.d_commit(be, levels, be.to_device(to_m(chunks[net.party_id])), 1 << n, net)

    res = LocalTestNet.simulate_network_round(np_, party)
    exp = po.d_commit_all(levels_pts, chunks)
    assert all(jac_pt(r) == exp for r in res)


@pytest.mark.parametrize("l", [1, 2])
def test_c_open_d_open_over_threads(l):
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    be = OracleBackend()
    n = 3
    rng = po.SplitMix64(140 + l)
    levels_pts = [po.g1_bases(max(1, (1 << k) // l), 400 + k) for k in range(n + 2)]  # new_single: level k has 2^k / l points
    levels = [OracleSrs(np.array([pt_mont(P) for P in lv])) for lv in levels_pts]
    pevals = [rng.fr_vec(1 << n) for _ in range(pp.n)]
    point = rng.fr_vec(n + 1)

    def party(net):
        return dp.c_open(be, levels, be.to_device(to_m(pevals[net.party_id])), 1 << n, to_m(point), pp, net)

    res = LocalTestNet.simulate_network_round(pp.n, party)
    exp = po.c_open_all(levels_pts, pevals, point, opp)
    for p in range(pp.n):
        assert ints(res[p][0]) == [exp[p][0]]
        assert [jac_pt(j) for j in res[p][1]] == exp[p][1]


def test_d_open_over_threads():
    be = OracleBackend()
    np_, n = 8, 3
    levels_pts = [po.g1_bases(1 << k, 500 + k) for k in range(n + 1)]
    levels = [OracleSrs(np.array([pt_mont(P) for P in lv])) for lv in levels_pts]
    rng = po.SplitMix64(150)
    chunks = [rng.fr_vec(1 << n) for _ in range(np_)]
    point = rng.fr_vec(n + 3)

    def party(net):
        return dp.d_open(be, levels, be.to_device(to_m(chunks[net.party_id])), 1 << n, to_m(point), net)

    res = LocalTestNet.simulate_network_round(np_, party)
    ev, eproofs = po.d_open_all(levels_pts, chunks, point)
    assert ints(res[0][0]) == [ev] and [jac_pt(j) for j in res[0][1]] == eproofs
    for p in range(1, np_):
        assert ints(res[p][0]) == [0] and len(res[p][1]) == 0  # workers: (0, [])


@pytest.mark.parametrize("l", [1, 2])
def test_degree_reduce_unpack_and_c_acc_product(l):
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    be = OracleBackend()
    rng = po.SplitMix64(160 + l)
    k = 5
    shares = [rng.fr_vec(k) for _ in range(pp.n)]
    inputs = [rng.fr_vec(8) for _ in range(pp.n)]

    def party(net):
        p = net.party_id
        a = dp.degree_reduce_many(to_m(shares[p]), pp, net)
        b = dp.degree_reduce(to_m(shares[p][:1])[0], pp, net)
        c = dp.d_unpack2_many(to_m(shares[p]), 3, pp, net)
        d = dp.d_unpack_0(to_m(shares[p][:1])[0], pp, net)
        tree, top = dp.c_acc_product(be, be.to_device(to_m(inputs[p])), 8, pp, net)
        return a, b, c, d, tree.download((16, 4)), top

    res = LocalTestNet.simulate_network_round(pp.n, party)
    exp = po.degree_reduce_many_all(shares, opp)
    sub, leader = po.c_acc_product_all(inputs, opp)
    flat = [v for kk in range(k) for v in opp.unpack2([shares[i][kk] for i in range(pp.n)])]
    for p in range(pp.n):
        assert ints(res[p][0]) == exp[p]
        assert ints(res[p][1]) == [exp[p][0]]
        assert ints(res[p][2]) == (flat if p == 3 else [])
        assert ints(res[p][3]) == [opp.unpack([shares[i][0] for i in range(pp.n)])[0]]
        assert ints(res[p][4]) == sub[p]
    assert ints(res[0][5]) == leader and all(res[p][5] is None for p in range(1, pp.n))


def test_d_fix_variable():
    pp, opp = PackedSharingParams(2), po.PackedSharingParams(2)
    be = OracleBackend()
    rng = po.SplitMix64(170)
    n = 3
    shares = [rng.fr_vec(1 << n) for _ in range(pp.n)]
    pts = rng.fr_vec(n + 1)

    def party(net):
        short = dp.d_fix_variable(be, be.to_device(to_m(shares[net.party_id])), 1 << n, to_m(pts[:2]), pp, net)
        full = dp.d_fix_variable(be, be.to_device(to_m(shares[net.party_id])), 1 << n, to_m(pts), pp, net)
        return short.download((2, 4)), full

    res = LocalTestNet.simulate_network_round(pp.n, party)
    lasts = [po.fix_variable(shares[p], pts[:n])[0] for p in range(pp.n)]
    ss = po.pss2ss_all(lasts, opp)
    for p in range(pp.n):
        assert ints(res[p][0]) == po.fix_variable(shares[p], pts[:2])
        assert ints(res[p][1]) == po.fold(ss[p], pts[0])  # one extra round re-using points[0] (mle.rs:78)


@pytest.mark.parametrize("l", [1, 2])
def test_c_acc_product_and_share(l):
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    be = OracleBackend()
    rng = po.SplitMix64(180 + l)
    S = 16 * pp.n
    vec = lambda: [rng.fr_vec(S) for _ in range(pp.n)]
    shares, masks, u0, u1, u2 = vec(), vec(), vec(), vec(), vec()

    def party(net):
        p = net.party_id
        d = lambda v: be.to_device(to_m(v[p]))
        return dp.c_acc_product_and_share(be, d(shares), d(masks), d(u0), d(u1), d(u2), S, pp, net)

    res = LocalTestNet.simulate_network_round(pp.n, party)
    exp = po.c_acc_product_and_share_all(shares, masks, u0, u1, u2, opp)
    for p in range(pp.n):
        for k in range(3):
            buf, cnt = res[p][k]  # device-resident results: (buffer, length)
            assert ints(buf.download((cnt, 4))) == exp[p][k], (p, k)


def test_msm_queue_merges_only_items_whose_owner_it_holds():
    """
    MsmQueue's duplicate detection names an item by (SRS level, scalar address, length).  An address only identifies a table
    while its allocation is alive: the queue must hold the owner of every address it uses as a key (buffer / view objects),
    and must never merge raw integer addresses (no owner: the block may be freed and handed out again for another table).
    """
    import gc
    import weakref

    from zkhip.api import DeviceBuffer, DeviceView

    freed = []

    class FakeLib:
        def zk_free(self, h, ptr):
            freed.append(ptr)

    class FakeCtx:
        lib, h = FakeLib(), 1

    def buf(ptr, nbytes=4096):  # a DeviceBuffer without a device
        b = DeviceBuffer.__new__(DeviceBuffer)
        b.ctx, b.nbytes, b.ptr = FakeCtx(), nbytes, ptr
        return b

    class Be:
        def msm_g1_batch(self, srs, bufs, lens):
            return np.zeros((len(lens), 18), dtype=np.uint64)

    srs = OracleSrs(np.zeros((8, 12), dtype=np.uint64))
    q = dp.MsmQueue(Be(), dedup=True)
    a = buf(0x1000)
    ia = q.add([srs], [a.at(64)], [4])
    wa = weakref.ref(a)
    del a
    gc.collect()
    assert wa() is not None and freed == [], "the queued view must keep its parent allocation alive"
    assert isinstance(q.bufs[0], DeviceView) and q.bufs[0].ptr == 0x1040
    # the same address through another live view of the same parent: the same item
    ib = q.add([srs], [wa().at(64)], [4])
    assert ia[0] == ib[0] and len(q.lens) == 1
    # raw integer addresses are never merged, not even with themselves
    ic, id_ = q.add([srs], [0x1040], [4]), q.add([srs], [0x1040], [4])
    assert len({int(ia[0]), int(ic[0]), int(id_[0])}) == 3 and len(q.lens) == 3
    # scale(): the source buffer named by the key is held as well
    class Be2(Be):
        def fr_scale(self, b, lam, n):
            return buf(0x9000)
    q2 = dp.MsmQueue(Be2(), dedup=True)
    src = buf(0x2000)
    lam = np.ones(4, dtype=np.uint64)
    s1 = q2.scale(src, lam, 4)
    ws = weakref.ref(src)
    del src
    gc.collect()
    assert ws() is not None and q2.scale(ws(), lam, 4) is s1
    q2.run()
    q.run()
    gc.collect()
    assert wa() is None and ws() is None and 0x1000 in freed and 0x2000 in freed, "owners are released when the pass is over"
    assert q.index == {} and q2.scaled == {}


def test_sumcheck_product_many_forms_equal_the_single_calls():
    """c_sumcheck_product_many / d_sumcheck_product_many(_q) batch the local phases and (d_) defer the exchange to a closure: the
    transcripts must be those of one call per item, i.e. of the all-parties restatement (dsumcheck.rs:148-285, 359-512)"""
    pp, opp = PackedSharingParams(1), po.PackedSharingParams(1)
    rng = po.SplitMix64(333)
    W, s = pp.n, 3
    items = [(3, rng.fr_vec(3 + s)), (2, rng.fr_vec(2 + s)), (3, rng.fr_vec(3 + s)), (0, rng.fr_vec(0 + s))]  # (log2 length, challenges)
    pf = [[rng.fr_vec(1 << lg) for _ in range(W)] for lg, _ in items]
    pg = [[rng.fr_vec(1 << lg) for _ in range(W)] for lg, _ in items]
    cf = [rng.fr_vec(8) for _ in range(W)]
    cg = [[rng.fr_vec(8) for _ in range(W)] for _ in range(2)]
    cch = rng.fr_vec(3)
    be = OracleBackend()

    def party(net):
        p = net.party_id
        dev = lambda v: be.to_device(to_m(v))
        d_items = [(dev(pf[k][p]), dev(pg[k][p]), 1 << lg, to_m(ch)) for k, (lg, ch) in enumerate(items)]
        fin = dp.d_sumcheck_product_many_q(be, d_items, net)
        many_c = dp.c_sumcheck_product_many(be, [(dev(cf[p]), dev(cg[0][p])), (dev(cf[p]), dev(cg[1][p]))], 8, to_m(cch), pp, net)  # another exchange in between
        many_d = fin()
        single_d = [dp.d_sumcheck_product(be, f, g, length, ch, net) for f, g, length, ch in d_items]
        single_c = [dp.c_sumcheck_product(be, dev(cf[p]), dev(cg[k][p]), 8, to_m(cch), pp, net) for k in range(2)]
        return many_d, single_d, many_c, single_c

    res = LocalTestNet.simulate_network_round(W, party)
    for p in range(W):
        many_d, single_d, many_c, single_c = res[p]
        for a, b in zip(many_d, single_d):
            assert a.shape == b.shape and (a == b).all()
        for a, b in zip(many_c, single_c):
            assert (a == b).all()
    for k, (lg, ch) in enumerate(items):
        assert [tuple(ints(t)) for t in res[0][0][k]] == po.d_sumcheck_product_all(pf[k], pg[k], ch)
    for k in range(2):
        exp = po.c_sumcheck_product_all(cf, cg[k], cch, opp)
        for p in range(W):
            assert [tuple(ints(t)) for t in res[p][2][k]] == exp[p]
