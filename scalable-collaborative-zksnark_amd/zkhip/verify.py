"""
Transcript verifiers -- host big-int restatement of the reference's checkers
(dist-primitive/src/dsumcheck.rs:541-588: `check_sumcheck`, `check_sumcheck_product`) plus the
self-checks the end-to-end drivers run on their own output.  They only read transcripts (a few
hundred field elements), so they are size-independent: the same checks run at n = 5 and at n = 24.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from .field import R_MOD, fr_from_mont

_INV2 = pow(2, -1, R_MOD)


def _ints(a) -> list:
    return [fr_from_mont(x) for x in np.asarray(a, dtype=np.uint64).reshape(-1, 4)]


def sumcheck_chain(proof, challenge, claimed: int | None = None) -> bool:
    """
    dsumcheck.rs:541-555: rows (lo, hi); row i+1 must sum to the line through row i evaluated at
    challenge[i].  `proof`: [k, 2, 4] Montgomery limbs (a trailing (0, last) row, if present, is the
    final evaluation and is checked against the last target).
    """
    rows = [tuple(_ints(r)) for r in np.asarray(proof, dtype=np.uint64).reshape(-1, 2, 4)]
    ch = _ints(challenge)
    if not rows:
        return True
    if claimed is not None and (rows[0][0] + rows[0][1]) % R_MOD != claimed % R_MOD:
        return False
    for i in range(1, len(rows)):
        lo, hi = rows[i - 1]
        target = ((hi - lo) * ch[i - 1] + lo) % R_MOD
        if rows[i][0] == 0 and i == len(rows) - 1 and rows[i][1] == target:
            continue  # the closing (0, last) row (dsumcheck.rs:24)
        if (rows[i][0] + rows[i][1]) % R_MOD != target:
            return False
    return True


def product_round_target(t0: int, t1: int, t2: int, x: int) -> int:
    """dsumcheck.rs:562-575: the degree-2 round polynomial through (0,t0), (1,t1), (2,t2) at x"""
    c = t0
    b = (-t2 + 4 * t1 - 3 * t0) * _INV2 % R_MOD
    a = (t2 - 2 * t1 + t0) * _INV2 % R_MOD
    return (a * x * x + b * x + c) % R_MOD


def sumcheck_product_chain(proof, challenge, claimed: int | None = None, closing: str = "auto", final: int | None = None) -> bool:
    """
    dsumcheck.rs:558-588 on rows (t0, t1, t2).  closing = "product": the last row is (0, f*g, 0)
    (`sumcheck_product`, :88) and must equal the last target; "none": every row is a round
    (`d_sumcheck_product` on the leader, Appendix A of SURVEY.md); "skip": a closing row is present but
    not comparable (c_sumcheck_product: pss2ss of the last values, :224-225,282); "auto": "product" if
    the last row has the (0, x, 0) shape.

    WHAT THE CHAIN ALONE PROVES.  The library derives t1 of every round after the first from this very identity
    (t1_k := p_{k-1}(r_{k-1}) - t0_k, csrc/zk_fr.hip derive_t1) on tables of 2^18 elements and more, so for such
    transcripts `t0 + t1 == previous target` holds by construction, whatever the t0 / t2 kernels produced.  The
    chain becomes a real check only with its two ends pinned by independently computed values:
      claimed  sum_j f_j g_j, compared with t0 + t1 of the first round;
      final    f(r) g(r) at the challenge point, compared with the LAST round polynomial at its challenge.
    With both given, a wrong t0 or t2 anywhere changes some round polynomial and the last target misses `final`
    (except with probability ~ rounds * 2 / r over the challenges): that is the sumcheck verifier's own soundness
    argument.  `product_anchor` computes the two values through kernels the product sumcheck does not use.
    """
    rows = [tuple(_ints(r)) for r in np.asarray(proof, dtype=np.uint64).reshape(-1, 3, 4)]
    ch = _ints(challenge)
    if not rows:
        return True
    if closing == "auto":
        closing = "product" if (rows[-1][0] == 0 and rows[-1][2] == 0 and len(rows) > 1) else "none"
    rounds = rows if closing == "none" else rows[:-1]
    if claimed is not None and rounds and (rounds[0][0] + rounds[0][1]) % R_MOD != claimed % R_MOD:
        return False
    cur = None
    for i, (t0, t1, t2) in enumerate(rounds):
        if cur is not None and (t0 + t1) % R_MOD != cur:
            return False
        cur = product_round_target(t0, t1, t2, ch[i])
    if final is not None and rounds and cur != final % R_MOD:
        return False
    if closing == "product" and rounds:
        return rows[-1][1] == cur
    return True


def product_anchor(be, f, g, length: int, challenge) -> tuple:
    """
    (sum_j f_j g_j,  f(r),  g(r)) as python ints, through kernels the product sumcheck does not share: the element-wise
    product + the PLAIN sumcheck's first round (zk_fr_mul, zk_sumcheck) and two folds (zk_fold).  r = the first
    log2(length) challenges.
    """
    n = length.bit_length() - 1
    ch = np.ascontiguousarray(challenge, dtype=np.uint64).reshape(-1, 4)[:n]
    prod = be.fr_mul(f, g, length)
    if n == 0:
        claimed = fr_from_mont(prod.download((1, 4))[0])
    else:
        pairs, _ = be.sumcheck(prod, length, ch)
        claimed = (fr_from_mont(pairs[0][0]) + fr_from_mont(pairs[0][1])) % R_MOD
    fr_ = fr_from_mont(be.fold(f, length, ch).download((1, 4))[0])
    gr_ = fr_from_mont(be.fold(g, length, ch).download((1, 4))[0])
    return claimed, fr_, gr_


def fold_ints(vals, challenges) -> int:
    """mle.rs:95-103 on python ints: the leader rounds of d_sumcheck_product fold the parties' last values"""
    v = [x % R_MOD for x in vals]
    for r in challenges:
        h = len(v) // 2
        v = [(v[j] + r * (v[j + h] - v[j])) % R_MOD for j in range(h)]
    assert len(v) == 1
    return v[0]


def trace_anchor_values(be, trace) -> list:
    """per traced product sumcheck (dist_primitive._trace): (kind, claimed, f(r), g(r), leader-round challenges as ints)"""
    out = []
    for kind, f, g, length, ch in trace:
        if kind == "keepalive":  # (a buffer the traced slices point into, no transcript)
            continue
        n = length.bit_length() - 1
        cl, fr_, gr_ = product_anchor(be, f, g, length, ch)
        out.append((kind, cl, fr_, gr_, _ints(np.asarray(ch, dtype=np.uint64).reshape(-1, 4)[n:])))
    return out


def check_closing_rows(values, proofs, pp, net) -> bool:
    """
    c_sumcheck_product's last row is (0, vf[0] * vg[0], 0) with vf / vg = pss2ss of the last values followed by the log2(l)
    extra rounds (dsumcheck.rs:224-225,282); at l = 1 that is pss2ss(f_last)[0] * pss2ss(g_last)[0].  f_last / g_last come
    from `values` (trace_anchor_values: two zk_fold per transcript), not from the sumcheck under test.  COLLECTIVE: every
    party calls it in lock step with its own values (two pss2ss exchanges per transcript).  l = 1 only.
    """
    from . import dist_primitive as dp
    from .field import fr_mont

    assert pp.l == 1
    ok = True
    for (kind, _cl, fr_, gr_, _), pr in zip(values, proofs):
        assert kind == "c"
        vf = fr_from_mont(dp.pss2ss(fr_mont(fr_), pp, net)[0])
        vg = fr_from_mont(dp.pss2ss(fr_mont(gr_), pp, net)[0])
        last = np.asarray(pr, dtype=np.uint64).reshape(-1, 3, 4)[-1]
        ok &= fr_from_mont(last[0]) == 0 and fr_from_mont(last[2]) == 0 and fr_from_mont(last[1]) == vf * vg % R_MOD
    return bool(ok)


def dhyperplonk_anchors(values_by_party: list, me: int, n_parties: int) -> dict:
    """
    label -> (claimed, final) for party `me` from the traced values of a dhyperplonk run.  values_by_party: one
    trace_anchor_values list per party, or a single list for the leader-echo net (the leader then stands for all
    N_p parties, as the echo net hands it N_p copies of its own message).  Trace order = transcript order: the six
    gate sumchecks, wiring[0], then wiring[1..].  d_sumcheck_product rows are sums over the parties and end with
    log2(N_p) leader rounds over the parties' last values (dsumcheck.rs:440-507).
    """
    echo = len(values_by_party) == 1
    mine = values_by_party[0 if echo else me]
    anchors = {}
    for e, (kind, cl, fr_, gr_, lead_ch) in enumerate(mine):
        label = f"gate[{e}]" if e < 6 else f"wiring[{e - 6}]"
        if kind == "d":
            if me != 0:
                continue
            per = [mine] * n_parties if echo else values_by_party
            claimed = sum(v[e][1] for v in per) % R_MOD
            final = fold_ints([v[e][2] for v in per], lead_ch) * fold_ints([v[e][3] for v in per], lead_ch) % R_MOD
            anchors[label] = (claimed, final)
        else:
            anchors[label] = (cl, fr_ * gr_ % R_MOD)
    return anchors


def check_dhyperplonk_transcripts(n: int, res, pk, n_parties: int, leader: bool, echo: bool, anchors: dict | None = None) -> list:
    """
    every sumcheck transcript of a dhyperplonk / dpermcheck result against its verifier chain; returns
    the list of failing labels (empty = all good).  c_sumcheck_product rows are share-level sums of this
    party's tables, d_sumcheck_product rows (leader only) the party-summed rounds plus log2(N_p) leader rounds.
    anchors: label -> (claimed, final) pinning both ends of a chain (see sumcheck_product_chain); a transcript without
    an anchor is only checked for internal consistency.  A missing proof is a failure, never skipped.
    """
    (gate_proofs, _gate_comms), (w_proofs, _w_commits, _w_opens) = res
    anchors = anchors or {}
    bad = []

    def chk(label, proof, ch, closing):
        cl, fi = anchors.get(label, (None, None))
        if not sumcheck_product_chain(proof, ch, claimed=cl, closing=closing, final=fi):
            bad.append(label)

    if len(gate_proofs) != 6:
        bad.append(f"gate: {len(gate_proofs)} proofs instead of 6")
    for i, pr in enumerate(gate_proofs):  # c_sumcheck_product(.., challenge)  dhyperplonk.rs:223-260
        chk(f"gate[{i}]", pr, pk.challenge, "skip")
    if not w_proofs:
        return bad + ["wiring: no proofs"]
    chk("wiring[0]", w_proofs[0], pk.challenge_r1, "skip")  # 2.c
    if leader:
        s = n_parties.bit_length() - 1
        want = 1 + 3 + 3 * (n - s) + 3
        if len(w_proofs) != want:
            bad.append(f"wiring: {len(w_proofs)} proofs instead of {want}")
            return bad
        k = 1
        for j in range(3):  # 2.e.1 :411-413
            chk(f"wiring[{k}]", w_proofs[k], pk.challenge_r2, "none")
            k += 1
        for i in range(1, n - s + 1):  # layered sumchecks :417-478, challenge_r2[i..]
            for j in range(3):
                chk(f"wiring[{k}]", w_proofs[k], pk.challenge_r2[i:], "none")
                k += 1
        for j in range(3):  # leader-tree sumchecks :506-508: plain sumcheck_product with the closing row
            chk(f"wiring[{k}]", w_proofs[k], pk.challenge_r2[:s], "product")
            k += 1
    return bad


def open_equation_terms(value, point, s) -> list:
    """
    The verifier equation of the multilinear commitment (PolynomialCommitment::verify, dpoly_comm.rs:466-484):
        e(C - value * g1, g2) == sum_i e(proof_i, s_i * g2 - point_i * g2).
    With the trapdoor s known (it is in every test of the reference, which builds the SRS from s, :502-531), both
    sides pull back to G1 through the non-degenerate pairing:  C - value * g1 == sum_i (s_i - point_i) * proof_i.
    Returns the canonical coefficients [(s_i - point_i) mod r]; the caller combines the points (zk_g1_lincomb) and
    compares group elements.  A pairing-free statement of the SAME equation, not a replacement for `verify`.
    """
    sv, pt = _ints(s), _ints(point)
    return [(a - b) % R_MOD for a, b in zip(sv, pt)]
