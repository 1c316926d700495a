"""
dhyperplonk call sequence (hyperplonk/src/dhyperplonk.rs:159-571) end to end:
  * CPU: 8 party threads over the oracle-backed compute stand-in -- structure and leader-echo plumbing (config 1)
  * GPU: the same sequence through libzkhip.so must reproduce the oracle-backed transcript bit for bit
"""
import numpy as np
import pytest

from oracle_backend import OracleBackend
from zkhip.hyperplonk import PackedProvingParameters, dhyperplonk
from zkhip.net import LeaderEchoNet, LocalTestNet
from zkhip.pss import PackedSharingParams


def _flatten(x, out):
    if isinstance(x, np.ndarray):
        out.append(np.ascontiguousarray(x, dtype=np.uint64).reshape(-1))
    elif isinstance(x, (list, tuple)):
        for e in x:
            _flatten(e, out)
    elif x is None:
        pass
    else:
        raise TypeError(type(x))
    return out


def _digest(res):
    import hashlib

    h = hashlib.sha256()
    for a in _flatten(res, []):
        h.update(a.tobytes())
    return h.hexdigest()


def _run_threads(make_backend, n):
    pp = PackedSharingParams(1)

    def party(net):
        be = make_backend()
        pk = PackedProvingParameters.new(n, pp, be, seed=100 + net.party_id)
        res, timers = dhyperplonk(n, pk, pp, be, net, seed=200 + net.party_id)
        return res

    return LocalTestNet.simulate_network_round(8, party)


def test_dhyperplonk_structure_cpu():
    n = 5
    res = _run_threads(OracleBackend, n)
    (gate_proofs, gate_comms), (w_proofs, w_commits, w_opens) = res[0]
    assert len(gate_proofs) == 6 and all(p.shape == (n + 1, 3, 4) for p in gate_proofs)  # c_sumcheck_product: n + log2(l) + 1
    assert len(gate_comms) == 6
    s = 3
    # wiring proofs: 1 (2.c) + 3 (2.e.1) + 3 per layered round + 3 leader-tree sumchecks   dhyperplonk.rs:304,411-413,423-447,506-508
    assert len(w_proofs) == 1 + 3 + 3 * (n - s) + 3
    assert w_proofs[0].shape == (n + 2 + 1, 3, 4)
    assert w_proofs[1].shape == ((n - 1) + s, 3, 4)  # d_sumcheck_product: n' + s rounds on the leader
    assert len(w_commits) == 1 + 8 + 3 and len(w_opens) == 3 + 5 + 3 * (n - s) + 3
    assert w_opens[0][1].shape == (n + 2, 18)  # c_open(V): n+2 proofs
    assert w_opens[2][1].shape == (s + (n - 1), 18)  # d_open: root proofs first, then n' local sums
    # workers get the worker-side values: empty d_sumcheck_product vectors, (0, []) d_opens
    (_, _), (wp1, wc1, wo1) = res[1]
    assert len(wp1[1]) == 0 and len(wo1[2][1]) == 0 and len(wp1) == 1 + 3 + 3 * (n - s)


def test_dhyperplonk_leader_echo_config1():
    """config 1 plumbing: `leader` mode, l = 1 (the no-`comm` fake): runs alone, only party 0 is meaningful"""
    n = 5
    pp = PackedSharingParams(1)
    be = OracleBackend()
    pk = PackedProvingParameters.new(n, pp, be, seed=1)
    net = LeaderEchoNet(8)
    res, timers = dhyperplonk(n, pk, pp, be, net, seed=2)
    assert set(timers) >= {"Commit", "Gate identity", "Wire identity", "Open", "Distributed HyperPlonk"}
    assert net.upload > 0 and net.upload == net.download


def test_dhyperplonk_leader_echo_config0_at_its_size():
    """
    BASELINE.json configs[0] as stated: hack/run-hyperplonk in `leader` mode, l = 1, 2^12 constraints, on the CPU path (the C
    port behind the same host driver; a few seconds).  Checks: every sumcheck transcript against its verifier chain with both
    ends pinned by independently computed values (zkhip.verify), the Appendix-B closed forms of the no-comm echo net
    (d_commit = 8 x the local commitment, d_msm = (4/7) x the plain MSM), the byte counters, and the shapes at n = 12.
    """
    import pyoracle as po
    from helpers import jac_norm_to_affine, pt_ints
    from zkhip import dist_primitive as dp
    from zkhip.verify import check_dhyperplonk_transcripts, dhyperplonk_anchors, trace_anchor_values

    n = 12
    pp = PackedSharingParams(1)
    be = OracleBackend()
    pk = PackedProvingParameters.new(n, pp, be, seed=12)
    net = LeaderEchoNet(8)
    be.sc_trace = []
    res, timers = dhyperplonk(n, pk, pp, be, net, seed=13)
    trace, be.sc_trace = be.sc_trace, None
    (gate_proofs, gate_comms), (w_proofs, w_commits, w_opens) = res
    s = 3
    assert len(gate_proofs) == 6 and len(gate_comms) == 6 and len(w_proofs) == 1 + 3 + 3 * (n - s) + 3
    assert all(np.asarray(p).shape == (n + 1, 3, 4) for p in gate_proofs)  # n rounds + the closing row at l = 1
    assert w_opens[0][1].shape == (n + 2, 18)
    values = trace_anchor_values(be, trace)
    anchors = dhyperplonk_anchors([values], 0, 8)
    assert len(anchors) == len(w_proofs) + 6
    assert check_dhyperplonk_transcripts(n, res, pk, 8, True, True, anchors=anchors) == []
    # Appendix B: the echo net hands the leader 8 copies of its own message
    M = 1 << n
    local = dp.commit(be, pk.d_commitment, pk.tables["ssigma_p"], 4 * M // 8)
    assert pt_ints(jac_norm_to_affine(w_commits[1])) == po.g1_mul(pt_ints(jac_norm_to_affine(local)), 8)
    plain = dp.commit(be, pk.c_commitment, pk.tables["a_evals"], M)
    four_sevenths = 4 * pow(7, -1, po.R_MOD) % po.R_MOD
    assert pt_ints(jac_norm_to_affine(gate_comms[0][0])) == po.g1_mul(pt_ints(jac_norm_to_affine(plain)), four_sevenths)
    assert net.upload > 0 and net.upload == net.download
    assert set(timers) >= {"Commit", "Gate identity", "Wire identity", "Open", "Distributed HyperPlonk"}


def test_dhyperplonk_batched_calls_keep_reference_positions():
    """
    the driver batches independent commits / opens; every output must still sit where the reference's
    sequential calls put it (dhyperplonk.rs:296-407, 417-478): recompute selected entries one call at a time
    """
    from zkhip import dist_primitive as dp
    from zkhip.field import random_fr

    n, seed = 5, 2
    pp = PackedSharingParams(1)
    be = OracleBackend()
    pk = PackedProvingParameters.new(n, pp, be, seed=1)
    net = LeaderEchoNet(8)
    (gate, (w_proofs, w_commits, w_opens)), _ = dhyperplonk(n, pk, pp, be, net, seed=seed)
    T, M, npar = pk.tables, 1 << n, 8
    hlen = 4 * M // npar
    local_s_p = be.to_device(random_fr(hlen, seed * 31 + 1))  # as drawn inside dhyperplonk
    dc, cc = pk.d_commitment, pk.c_commitment

    def same_open(a, b):
        return (np.asarray(a[0]) == np.asarray(b[0])).all() and np.asarray(a[1]).shape == np.asarray(b[1]).shape and (np.asarray(a[1]) == np.asarray(b[1])).all()

    assert (w_commits[0] == dp.d_commit(be, dc, local_s_p, hlen, net)).all()                   # 2.b
    assert (w_commits[1] == dp.d_commit(be, dc, T["ssigma_p"], hlen, net)).all()               # first of :363-380
    assert (w_commits[2] == dp.d_commit(be, dc, T["sid_p"], hlen, net)).all()
    assert same_open(w_opens[0], dp.c_open(be, cc, T["V"], 4 * M // pp.l, pk.challenge_r1, pp, net))  # 2.d
    assert same_open(w_opens[1], dp.c_open(be, cc, T["V"], 4 * M // pp.l, pk.challenge_r2, pp, net))
    assert same_open(w_opens[2], dp.d_open(be, dc, local_s_p, hlen, pk.challenge_r2, net))
    assert same_open(w_opens[3], dp.d_open(be, dc, T["ssigma_p"], hlen, pk.challenge_r2, net))  # first of :383-407
    assert same_open(w_opens[4], dp.d_open(be, dc, T["sid_p"], hlen, pk.challenge_r2, net))
    # gate commitments: (commitment, open) pairs in the order a, b, c, I, S1, S2 (:517-553)
    assert (gate[1][0][0] == dp.c_commit(be, cc, [T["a_evals"]], [pk.lens["a_evals"]], pp, net)[0]).all()
    assert same_open(gate[1][1][1], dp.c_open(be, cc, T["b_evals"], pk.lens["b_evals"], pk.challenge, pp, net))
    assert (gate[1][3][0] == dp.d_commit(be, dc, T["I_p"], pk.lens["I_p"], net)).all()
    assert same_open(gate[1][5][1], dp.d_open(be, dc, T["S2_p"], pk.lens["S2_p"], pk.challenge, net))


@pytest.mark.gpu
def test_dhyperplonk_gpu_matches_oracle_backed_run():
    import zkhip

    n = 6
    exp = _run_threads(OracleBackend, n)
    got = _run_threads(lambda: zkhip.Ctx(0), n)
    for p in range(8):
        assert _digest(got[p]) == _digest(exp[p]), f"party {p}"


@pytest.mark.gpu
def test_dhyperplonk_data_parallel_gpu():
    import zkhip

    n = 5
    pp = PackedSharingParams(1)

    def party_with(make):
        def party(net):
            be = make()
            pk = PackedProvingParameters.new(n, pp, be, seed=300 + net.party_id)
            return dhyperplonk(n, pk, pp, be, net, seed=400 + net.party_id, data_parallel=True)[0]
        return party

    exp = LocalTestNet.simulate_network_round(8, party_with(OracleBackend))
    got = LocalTestNet.simulate_network_round(8, party_with(lambda: zkhip.Ctx(0)))
    assert _digest(got[0]) == _digest(exp[0])


def _run_perm(make_backend, n, which):
    from zkhip.hyperplonk import cpermcheck, dpermcheck

    pp = PackedSharingParams(1)
    fn = {"d": dpermcheck, "c": cpermcheck}[which]

    def party(net):
        be = make_backend()
        pk = PackedProvingParameters.new(n, pp, be, seed=500 + net.party_id)
        return fn(n, pk, pp, be, net, seed=600 + net.party_id)[0]

    return LocalTestNet.simulate_network_round(8, party)


def test_permchecks_structure_cpu():
    n = 5
    dres = _run_perm(OracleBackend, n, "d")
    full = _run_threads(OracleBackend, n)
    # dpermcheck is exactly the wiring-identity step of dhyperplonk: same shapes
    assert [len(x) for x in dres[0]] == [len(x) for x in full[0][1]]
    cres = _run_perm(OracleBackend, n, "c")
    proofs, commits, opens = cres[0]
    assert len(proofs) == 6 and len(commits) == 2 + 2 * 4 and len(opens) == 2 + 2 * 5  # dhyperplonk.rs:1289-1375
    assert proofs[0].shape == (n + 2 + 1, 3, 4) and opens[0][1].shape == (n + 2, 18)


@pytest.mark.parametrize("l,n,fn_name", [(1, 5, "dhyperplonk"), (2, 6, "dhyperplonk"), (1, 5, "dpermcheck")])
def test_one_batch_schedule_equals_the_batch_per_call_form(l, n, fn_name, monkeypatch):
    """ONE_BATCH (default): the sumcheck-family kernels of steps 2-4 are added to one dp.ScQueue, the exchanges that need a kernel
    result follow in phase B; ONE_BATCH = False runs every primitive's kernels inside its own call.  Same transcript on every
    party, same bytes on the wire"""
    from zkhip import hyperplonk as hp

    pp = PackedSharingParams(l)

    def party(net):
        be = OracleBackend()
        pk = PackedProvingParameters.new(n, pp, be, seed=1200 + net.party_id, chal_seed=78)
        res = getattr(hp, fn_name)(n, pk, pp, be, net, seed=1250 + net.party_id)[0]
        return res, (net.upload, net.download)

    got = LocalTestNet.simulate_network_round(pp.n, party)
    monkeypatch.setattr(hp, "ONE_BATCH", False)
    exp = LocalTestNet.simulate_network_round(pp.n, party)
    for p in range(pp.n):
        assert _digest(got[p][0]) == _digest(exp[p][0]), f"party {p}"
        assert got[p][1] == exp[p][1], f"party {p}: bytes on the wire"


@pytest.mark.parametrize("l,n", [(1, 5), (2, 6)])
def test_cpermcheck_pipelined_equals_the_call_by_call_form(l, n, monkeypatch):
    """cpermcheck queues all commitments / quotient commitments into ONE MSM pass, batches the opens' fold rounds and the product
    sumchecks, and runs the repeated open of num / den (dhyperplonk.rs:1324, :1371: same table, same point) once; CPERM_SERIAL is
    the reference's call-by-call order.  Same transcript for every party, and the repeated opens are equal"""
    from zkhip import hyperplonk as hp

    pp = PackedSharingParams(l)

    def party(net):
        be = OracleBackend()
        pk = PackedProvingParameters.new(n, pp, be, seed=900 + net.party_id, chal_seed=77)
        return hp.cpermcheck(n, pk, pp, be, net, seed=950 + net.party_id)[0]

    got = LocalTestNet.simulate_network_round(pp.n, party)
    monkeypatch.setattr(hp, "CPERM_SERIAL", True)
    exp = LocalTestNet.simulate_network_round(pp.n, party)
    for p in range(pp.n):
        assert _digest(got[p]) == _digest(exp[p]), f"party {p}"
    opens = got[0][2]
    assert len(opens) == 12
    for a, b in ((2, 6), (7, 11)):  # num, den: opened at :1324 and again at :1371
        assert (opens[a][0] == opens[b][0]).all() and (opens[a][1] == opens[b][1]).all()


@pytest.mark.gpu
def test_permchecks_gpu_match_oracle_backed_run():
    import zkhip

    n = 5
    for which in ("d", "c"):
        exp = _run_perm(OracleBackend, n, which)
        got = _run_perm(lambda: zkhip.Ctx(0), n, which)
        assert _digest(got[0]) == _digest(exp[0]) and _digest(got[3]) == _digest(exp[3]), which


def _run_l2(make_backend, n, fn_name):
    from zkhip import hyperplonk as hp

    pp = PackedSharingParams(2)  # l = 2: 16 parties, the packed maps are no longer the l = 1 closed forms

    def party(net):
        be = make_backend()
        pk = PackedProvingParameters.new(n, pp, be, seed=700 + net.party_id, chal_seed=4711)
        return getattr(hp, fn_name)(n, pk, pp, be, net, seed=800 + net.party_id)[0]

    return LocalTestNet.simulate_network_round(pp.n, party)


def test_dhyperplonk_l2_structure_cpu():
    """packing factor 2 (16 parties): c_sumcheck_product gains log2(l) rounds, d_* leader rounds grow to log2(16)"""
    n = 6  # the 16-leaf top tree needs d_commitment level 4 = n + 2 - log2(16) (dhyperplonk.rs:101)
    res = _run_l2(OracleBackend, n, "dhyperplonk")
    (gate_proofs, gate_comms), (w_proofs, w_commits, w_opens) = res[0]
    assert all(p.shape == ((n - 1) + 1 + 1, 3, 4) for p in gate_proofs)  # log2(M / l) + log2(l) + 1
    assert w_proofs[1].shape == ((n - 2) + 4, 3, 4)                      # log2(4M / 16) local + log2(16) leader rounds
    assert w_opens[0][1].shape == ((n + 1) + 1, 18)                      # c_open(V): log2(4M / l) + log2(l) proofs


@pytest.mark.gpu
@pytest.mark.parametrize("fn_name", ["dhyperplonk", "cpermcheck"])
def test_l2_sixteen_parties_gpu_matches_oracle_backed_run(fn_name):
    import zkhip

    n = 6
    exp = _run_l2(OracleBackend, n, fn_name)
    got = _run_l2(lambda: zkhip.Ctx(0), n, fn_name)
    for p in (0, 1, 15):
        assert _digest(got[p]) == _digest(exp[p]), f"{fn_name}: party {p}"


def test_sc_queue_two_phase_forms_cpu():
    """dp.ScQueue (the mirror of zkhost/pipeline.hpp): a result cannot be read before the batch ran; the two-phase open equals the
    one-call form; the same table at the same point twice is ONE request whose MSM items the queue computes once"""
    from zkhip import dist_primitive as dp
    from zkhip.field import random_fr
    from zkhip.net import LeaderEchoNet

    be = OracleBackend()
    pp = PackedSharingParams(1)
    net = LeaderEchoNet(pp.n)
    n = 4
    length = 1 << n
    pc = dp.PolynomialCommitmentCub.new_single(be, n + 1, pp, seed=5)
    tab, other = be.to_device(random_fr(length, 31)), be.to_device(random_fr(length, 32))
    pt = random_fr(n + 3, 33)
    sq, q = dp.ScQueue(be), dp.MsmQueue(be)
    phase_b = dp.c_open_many_sq(be, sq, q, pc.powers_of_g, [tab, other, tab], [length] * 3, [pt, pt, pt], pp, net)
    assert len(sq.reqs) == 2  # (tab, pt) twice -> one request
    with pytest.raises(RuntimeError):
        sq.at(0)
    sq.run()
    fin = phase_b()
    assert len(q.lens) == 2 * n  # the repeated open's items are the first one's
    q.run()
    got = fin()
    exp = dp.c_open_many(be, pc.powers_of_g, [tab, other], [length] * 2, [pt, pt], pp, net)
    for g, e in zip(got, [exp[0], exp[1], exp[0]]):
        assert (g[0] == e[0]).all() and (g[1] == e[1]).all()
