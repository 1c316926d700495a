"""
The collaborative primitives against the reference's OWN sample run.

hack/run-hyperplonk/output.txt is the leader's log of a real 128-party run of the reference (l = 16, 2^12 constraints); every
`Comm: from 0 to X, <bytes>B` line is the size of party 0's ark-serialize (compressed) message, i.e. the SHAPE of what a
primitive hands to the network.  tests/golden/ref_log_n12_l16.json holds those lines (tests/golden/make_ref_log_fixture.py);
here every primitive of the path that appears in the log runs at the same parameters (party 0, 128 parties, tables of
2^14 / 16 = 1024 shares) over a recording net, and the sizes of its exchange payloads -- re-expressed as the reference would
serialise them: Fr 32 B, G1 48 B, Vec<T> = 8 + items -- must reproduce the log:

    c_commit (one polynomial)             56 B            Vec<G1> of 1                     dpoly_comm.rs:244-267, dmsm.rs:29
    pss2ss                                32 B            one Fr                           unpack.rs:72-97
    c_sumcheck_product                    32, 32 B        two pss2ss hand-offs             dsumcheck.rs:224-225
    c_acc_product_and_share               128 x 264 B     blocks of S / N_p = 8 masked Fr  dacc_product.rs:94-104
                                          4104 B          last min(N_p, 2N) = 128 entries  dacc_product.rs:321-329
    c_open                                488, 32 B       10 = log2(1024) commitments in ONE d_msm, then pss2ss   dpoly_comm.rs:436,439
    degree_reduce_many(reduce_target)     1032 B          M / (8 l) = 32 Fr                degree_reduce.rs:10-26

This pins the exchange structure of the restatement (batch shapes, the l-dependence, which values travel) to an artefact the
reference itself produced; the VALUES inside the messages stay pinned by the oracle and the KATs only.
"""
import json
import os

import numpy as np
import pytest

from oracle_backend import OracleBackend
from zkhip import dist_primitive as dp
from zkhip.field import random_fr
from zkhip.net import LeaderEchoNet
from zkhip.pss import PackedSharingParams

HERE = os.path.dirname(os.path.abspath(__file__))
LOG = json.load(open(os.path.join(HERE, "golden", "ref_log_n12_l16.json")))
N, L_PACK, PARTIES = LOG["n"], LOG["l"], LOG["parties"]
SHARES = 4 * (1 << N) // L_PACK  # the 4M-element tables (V, sid, ssigma, ...) are 4M / l = 1024 packed shares per party


def _ser(a) -> int:
    """ark-serialize (compressed) size of what this numpy payload is in the reference"""
    a = np.asarray(a)
    if a.ndim == 1 and a.size == 4:
        return 32  # one Fr
    if a.shape[-1] == 4:
        return 8 + 32 * (a.size // 4)  # Vec<Fr>
    if a.shape[-1] == 18:
        return 8 + 48 * (a.size // 18)  # Vec<G1>, compressed points
    raise AssertionError(a.shape)


class RecordingNet(LeaderEchoNet):
    """the no-comm echo net of config 0, logging the reference-equivalent size of party 0's message of every exchange"""

    def __init__(self, n):
        super().__init__(n)
        self.log = []

    def all_gather(self, a):
        self.log.append(("to_leader", _ser(a)))
        return super().all_gather(a)

    def all_to_all(self, chunks, echo="slot0"):
        self.log.append(("to_each", [8 + np.asarray(c).nbytes for c in chunks]))
        return super().all_to_all(chunks, echo)

    def all_gather_device(self, d_send, nbytes, d_recv=None, be=None):
        self.log.append(("to_leader", 8 + nbytes))
        return super().all_gather_device(d_send, nbytes, d_recv, be)

    def all_to_all_device(self, d_send, nbytes_per_peer, d_recv=None, be=None, echo="slot0"):
        self.log.append(("to_each", [8 + nbytes_per_peer] * self.n_parties))
        return super().all_to_all_device(d_send, nbytes_per_peer, d_recv, be, echo)


def _ref(section: str, to=None):
    """byte sizes of the log's exchanges whose enclosing timers end with `section`, in log order"""
    return [(e[3], e[4]) for e in LOG["exchanges"] if e[1].endswith(section) and e[3] != "all" and (to is None or e[3] == to)]


@pytest.fixture(scope="module")
def setup():
    assert (N, L_PACK, PARTIES) == (12, 16, 128) and SHARES == 1024
    pp = PackedSharingParams(L_PACK)
    assert pp.n == PARTIES
    be = OracleBackend()
    levels = [be.srs_generate(3 + 2 * i, 5 + 2 * i, max(1, (1 << i) // L_PACK)) for i in range(N + 3)]  # new_single, dpoly_comm.rs:197-219
    return pp, be, levels


def test_log_fixture_is_what_the_script_extracts():
    assert len(LOG["exchanges"]) == 400 and LOG["comm_totals_up_down"] == [14411071, 2425319]
    assert _ref("Commit > Send to leader for MSM")[0] == ("leader", 56)


def test_c_commit_and_pss2ss_message_shapes(setup):
    pp, be, levels = setup
    net = RecordingNet(PARTIES)
    dp.c_commit(be, levels, [be.to_device(random_fr(SHARES, 1))], [SHARES], pp, net)
    assert net.log == [("to_leader", 56)] and ("leader", 56) in _ref("Commit > Send to leader for MSM")
    net.log.clear()
    out = dp.pss2ss(random_fr(1, 2)[0], pp, net)
    assert out.shape == (L_PACK, 4) and net.log == [("to_leader", 32)]
    assert set(_ref("PSStoSS")) == {("leader", 32)}


def test_c_sumcheck_product_hands_off_twice(setup):
    pp, be, _ = setup
    net = RecordingNet(PARTIES)
    ch = random_fr(N + 8, 3)
    proof = dp.c_sumcheck_product(be, be.to_device(random_fr(SHARES, 4)), be.to_device(random_fr(SHARES, 5)), SHARES, ch, pp, net)
    assert proof.shape == (10 + 4 + 1, 3, 4)  # log2(1024) rounds on the shares, log2(l) on the l-vector, the closing row
    first = [b for _, b in _ref("Distributed sumcheck product > PSStoSS")][:2]
    assert [b for _, b in net.log] == first == [32, 32]


def test_c_open_batches_its_commitments_like_the_log(setup):
    pp, be, levels = setup
    net = RecordingNet(PARTIES)
    val, proofs = dp.c_open(be, levels, be.to_device(random_fr(SHARES, 6)), SHARES, random_fr(N + 2, 7), pp, net)
    assert proofs.shape == (10 + 4, 18)
    want = [_ref("Distributed opening > Send to leader for MSM")[0][1], _ref("Distributed opening > PSStoSS")[0][1]]
    assert want == [488, 32]
    # same two messages; the library hands the last value off BEFORE its MSM pass (the phase-2 commitments of :441-462 then ride in
    # the same batched pass as the 10 phase-1 commitments), the reference after it (:436, :439): a re-ordering, on every party alike
    assert [b for _, b in net.log] == [32, 488]


def test_c_acc_product_and_share_blocks_and_tail(setup):
    pp, be, _ = setup
    net = RecordingNet(PARTIES)
    d = lambda s: be.to_device(random_fr(SHARES, s))
    res = dp.c_acc_product_and_share(be, d(8), d(9), d(10), d(11), d(12), SHARES, pp, net)
    assert all(cnt == SHARES for _, cnt in res)
    masked = _ref("Leader distributes masked elements")[:PARTIES]
    assert [t for t, _ in masked] == [str(i) for i in range(PARTIES)]  # one dynamic scatter per receiving party, in order
    assert net.log[0] == ("to_each", [b for _, b in masked]) and masked[0][1] == 264
    assert net.log[1] == ("to_leader", _ref("Send elements to leader")[0][1]) and net.log[1][1] == 4104


def test_degree_reduce_many_of_the_reduce_target(setup):
    pp, be, _ = setup
    net = RecordingNet(PARTIES)
    k = (1 << N) // (8 * L_PACK)  # reduce_target: M / (8 l) shares (dhyperplonk.rs:117-120)
    out = dp.degree_reduce_many(random_fr(k, 13), pp, net)
    assert out.shape == (k, 4)
    assert [b for _, b in net.log] == [b for _, b in _ref("Degree reduce")] == [1032]
