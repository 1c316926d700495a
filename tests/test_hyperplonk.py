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


@pytest.mark.gpu
def test_permchecks_gpu_match_oracle_backed_run():
    import zkhip

    n = 5
    for which in ("d", "c"):
        exp = _run_perm(OracleBackend, n, which)
        got = _run_perm(lambda: zkhip.Ctx(0), n, which)
        assert _digest(got[0]) == _digest(exp[0]) and _digest(got[3]) == _digest(exp[3]), which
