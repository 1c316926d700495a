"""
BASELINE configs 3 / 4 at FULL size on one GPU, self-checking (collaborative HyperPlonk l = 1, n = 20 and
n = 24; hyperplonk/src/dhyperplonk.rs:159-571).  No oracle run is affordable at these sizes, so the checks
are the size-independent properties the domain offers:

  P1  every sumcheck transcript passes its verifier chain (dsumcheck.rs:541-588, zkhip.verify) WITH BOTH ENDS PINNED by
      values computed through kernels the product sumcheck does not use: the claim sum_j f_j g_j (element-wise product +
      the plain sumcheck's first round) and the final evaluation f(r) g(r) (two folds) -- the chain alone is vacuous on
      tables of 2^18 elements and more, where the library derives t1 from it (see zkhip.verify.sumcheck_product_chain);
      the closing row of every c_sumcheck_product is checked against pss2ss of the independently folded last values;
  P2  d_commit == sum over parties of the local commitments (dpoly_comm.rs:276-297), added with the
      ORACLE's group law;
  P3  sampled opens / commits equal a one-call-at-a-time recomputation (other batch shapes, other window
      classes: the batched driver must put every output at the reference's position);
  P4  opened values equal the CPU oracle's fold of the downloaded table (coracle.fold);
  P5  a second run reproduces the transcript bit for bit (stream / batch races would show here).

Modes: `leader` (party 0 alone over the no-comm echo net, config 1 style) and 8 party threads sharing GPU 0
(LocalTestNet).  The 8-GPU RCCL run of the same driver is tests/test_gpu_comm.py + tools/hyperplonk_bench.py.
"""
import hashlib
from types import SimpleNamespace

import numpy as np
import pytest

import pyoracle as po
from helpers import jac_norm_to_affine, pt_ints

pytestmark = pytest.mark.gpu


def _flatten(x, out):
    if isinstance(x, np.ndarray):
        out.append(np.ascontiguousarray(x, dtype=np.uint64).reshape(-1))
    elif isinstance(x, (list, tuple)):
        for e in x:
            _flatten(e, out)
    elif x is not None:
        raise TypeError(type(x))
    return out


def _digest(res):
    h = hashlib.sha256()
    for a in _flatten(res, []):
        h.update(a.tobytes())
    return h.hexdigest()


def _same_open(a, b):
    return (np.asarray(a[0]) == np.asarray(b[0])).all() and np.asarray(a[1]).shape == np.asarray(b[1]).shape and (np.asarray(a[1]) == np.asarray(b[1])).all()


def _oracle_eval(co, table: np.ndarray, point: np.ndarray) -> np.ndarray:
    """fold the table with every coordinate of `point` on the CPU oracle -> the single remaining element"""
    cur = table
    for r in point:
        cur = co.fold(cur, r)
    assert len(cur) == 1
    return cur[0]


def _leader_run(n, seed=3):
    import zkhip
    from zkhip.hyperplonk import PackedProvingParameters, dhyperplonk
    from zkhip.net import LeaderEchoNet
    from zkhip.pss import PackedSharingParams

    pp = PackedSharingParams(1)
    ctx = zkhip.Ctx(0)
    pk = PackedProvingParameters.new(n, pp, ctx, seed=seed)
    net = LeaderEchoNet(8)
    ctx.sc_trace = []  # dist_primitive records the operands of every product sumcheck of the run (test hook)
    res, timers = dhyperplonk(n, pk, pp, ctx, net, seed=seed + 1)
    return ctx, pk, pp, net, res, timers, seed + 1


def _closing_rows_ok(dp, values, proofs, pp, net):
    """c_sumcheck_product's last row is (0, pss2ss(f_last)[0] * pss2ss(g_last)[0], 0) at l = 1 (dsumcheck.rs:224-225,282);
    f_last / g_last here come from zk_fold, not from the sumcheck under test.  Every party calls this in lock step."""
    from zkhip.verify import check_closing_rows

    return check_closing_rows(values, proofs, pp, net)


def _check_leader(n, co, ctx, pk, pp, net, res, run_seed, oracle_open=True):
    from zkhip import dist_primitive as dp
    from zkhip.field import random_fr
    from zkhip.hyperplonk import dhyperplonk
    from zkhip.verify import check_dhyperplonk_transcripts, dhyperplonk_anchors, trace_anchor_values

    (gate_proofs, gate_comms), (w_proofs, w_commits, w_opens) = res
    T, M, npar = pk.tables, 1 << n, 8
    hlen = 4 * M // npar
    dc, cc = pk.d_commitment, pk.c_commitment
    # P1: 6 gate + 1 + 3 + 3 (n - 3) + 3 transcripts, every one with an independent claim and final evaluation
    trace, ctx.sc_trace = ctx.sc_trace, None
    values = trace_anchor_values(ctx, trace)
    assert len(values) == 6 + 1 + 3 + 3 * (n - 3) + 3
    anchors = dhyperplonk_anchors([values], 0, npar)
    assert len(anchors) == len(values)
    assert check_dhyperplonk_transcripts(n, res, pk, npar, True, True, anchors=anchors) == []
    assert _closing_rows_ok(dp, values[:7], list(gate_proofs) + [w_proofs[0]], pp, net)
    # (the check has teeth: a transcript with one t2 off by one is rejected)
    broken = np.array(gate_proofs[3], copy=True)
    broken[n // 2, 2, 0] ^= np.uint64(1)
    res_bad = ((list(gate_proofs[:3]) + [broken] + list(gate_proofs[4:]), gate_comms), (w_proofs, w_commits, w_opens))
    assert check_dhyperplonk_transcripts(n, res_bad, pk, npar, True, True, anchors=anchors) == ["gate[3]"]
    del trace
    # P2: the echo net hands the leader N_p copies of its own commitment: d_commit = 8 * local (Appendix B)
    local = dp.commit(ctx, dc, T["ssigma_p"], hlen)
    assert pt_ints(jac_norm_to_affine(w_commits[1])) == po.g1_mul(pt_ints(jac_norm_to_affine(local)), 8)
    # P3: one call at a time
    local_s_p = ctx.to_device(random_fr(hlen, run_seed * 31 + 1))  # as drawn inside dhyperplonk
    assert (w_commits[0] == dp.d_commit(ctx, dc, local_s_p, hlen, net)).all()
    assert _same_open(w_opens[1], dp.c_open(ctx, cc, T["V"], 4 * M, pk.challenge_r2, pp, net))
    assert _same_open(w_opens[4], dp.d_open(ctx, dc, T["sid_p"], hlen, pk.challenge_r2, net))
    assert (gate_comms[0][0] == dp.c_commit(ctx, cc, [T["a_evals"]], [pk.lens["a_evals"]], pp, net)[0]).all()
    assert _same_open(gate_comms[5][1], dp.d_open(ctx, dc, T["S2_p"], pk.lens["S2_p"], pk.challenge, net))
    # P4: d_open's root value = the local table folded with point[s..] (eight equal leaves fold to themselves)
    if oracle_open:
        tab = T["sid_p"].download((hlen, 4))
        assert (w_opens[4][0] == _oracle_eval(co, tab, pk.challenge_r2[3 : 3 + (hlen.bit_length() - 1)])).all()
    # P5
    res2, _ = dhyperplonk(n, pk, pp, ctx, net, seed=run_seed)
    assert _digest(res2) == _digest(res)


def test_dhyperplonk_n20_leader_mode(co):
    n = 20
    ctx, pk, pp, net, res, timers, run_seed = _leader_run(n)
    try:
        _check_leader(n, co, ctx, pk, pp, net, res, run_seed)
    finally:
        ctx.close()


def test_dhyperplonk_n20_eight_party_threads(co):
    """8 parties = 8 threads with one ctx each on GPU 0, different tables per party, shared challenges"""
    import zkhip
    from zkhip import dist_primitive as dp
    from zkhip.hyperplonk import PackedProvingParameters, dhyperplonk
    from zkhip.net import LocalTestNet
    from zkhip.pss import PackedSharingParams
    from zkhip.verify import check_dhyperplonk_transcripts, dhyperplonk_anchors, trace_anchor_values

    n = 20
    pp = PackedSharingParams(1)
    M, npar = 1 << n, 8
    hlen = 4 * M // npar

    def party(net):
        ctx = zkhip.Ctx(0)
        try:
            pk = PackedProvingParameters.new(n, pp, ctx, seed=40 + net.party_id, chal_seed=999)
            ctx.sc_trace = []
            res, _ = dhyperplonk(n, pk, pp, ctx, net, seed=50 + net.party_id)
            trace, ctx.sc_trace = ctx.sc_trace, None
            values = trace_anchor_values(ctx, trace)  # this party's claims / final evaluations, independent kernels
            closing_ok = _closing_rows_ok(dp, values[:7], list(res[0][0]) + [res[1][0][0]], pp, net)
            del trace
            bad = check_dhyperplonk_transcripts(n, res, pk, npar, net.is_leader, False)
            local_commit = dp.commit(ctx, pk.d_commitment, pk.tables["ssigma_p"], hlen)
            local_open = dp.open_(ctx, pk.d_commitment, pk.tables["sid_p"], hlen, pk.challenge_r2[3:])
            res2, _ = dhyperplonk(n, pk, pp, ctx, net, seed=50 + net.party_id)
            sid_tab = pk.tables["sid_p"].download((hlen, 4)) if net.is_leader else None
            return dict(bad=bad, digest=_digest(res), digest2=_digest(res2), local_commit=local_commit, local_open=local_open,
                        values=values, closing_ok=closing_ok, sc=(res[0][0], res[1][0]),
                        chal=SimpleNamespace(challenge=pk.challenge, challenge_r1=pk.challenge_r1, challenge_r2=pk.challenge_r2),
                        res=res if net.is_leader else None, w_commit_ssigma=res[1][1][1], chal_r2=pk.challenge_r2, sid_tab=sid_tab)
        finally:
            ctx.close()

    out = LocalTestNet.simulate_network_round(8, party)
    vals = [out[p]["values"] for p in range(8)]
    for p in range(8):
        assert out[p]["bad"] == [], (p, out[p]["bad"])  # P1 (consistency) ...
        assert out[p]["closing_ok"], p
        # ... and with both ends of every chain pinned: c_ transcripts by the party's own tables, the leader's d_ transcripts
        # by the sums over all parties' claims and the fold of all parties' final values
        gate_proofs, w_proofs = out[p]["sc"]
        fake = ((gate_proofs, None), (w_proofs, None, None))
        anchors = dhyperplonk_anchors(vals, p, npar)
        assert len(anchors) == (len(vals[0]) if p == 0 else 7)
        assert check_dhyperplonk_transcripts(n, fake, out[0]["chal"], npar, p == 0, False, anchors=anchors) == [], p
        assert out[p]["digest"] == out[p]["digest2"], p  # P5
    # P2: every party holds the same d_commit, equal to the sum of the local commitments (oracle group law)
    total = None
    for p in range(8):
        total = po.g1_add(total, pt_ints(jac_norm_to_affine(out[p]["local_commit"])))
    for p in range(8):
        assert pt_ints(jac_norm_to_affine(out[p]["w_commit_ssigma"])) == total
    # P3 / P4 on the leader's d_open of sid_p: root value = the 8 local values folded with point[..3];
    # proofs = 3 root proofs, then the party-sums of the local proofs (dpoly_comm.rs:372-391)
    (_, _), (_, _, w_opens) = out[0]["res"]
    val, proofs = w_opens[4]
    ch = out[0]["chal_r2"]
    local_vals = np.stack([out[p]["local_open"][0] for p in range(8)])
    assert (val == _oracle_eval(co, local_vals, ch[:3])).all()
    assert (out[0]["local_open"][0] == _oracle_eval(co, out[0]["sid_tab"], ch[3 : 3 + (hlen.bit_length() - 1)])).all()
    nl = hlen.bit_length() - 1
    assert proofs.shape == (3 + nl, 18)
    for i in (0, nl // 2, nl - 1):
        s = None
        for p in range(8):
            s = po.g1_add(s, pt_ints(jac_norm_to_affine(out[p]["local_open"][1][i])))
        assert pt_ints(jac_norm_to_affine(proofs[3 + i])) == s


def test_dhyperplonk_n24_leader_mode(co):
    """config 4's per-party work (16.7 M constraints, 398 M scalar-muls) -- behind a memory guard"""
    import zkhip

    probe = zkhip.Ctx(0)
    free, total = probe.mem_info()
    probe.close()
    if free < 96 << 30:
        pytest.skip(f"only {free >> 30} GiB of HBM free: the n = 24 run wants ~70 GiB")
    n = 24
    ctx, pk, pp, net, res, timers, run_seed = _leader_run(n, seed=5)
    try:
        _check_leader(n, co, ctx, pk, pp, net, res, run_seed)
    finally:
        ctx.close()
