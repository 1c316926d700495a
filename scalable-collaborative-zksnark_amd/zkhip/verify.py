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


def sumcheck_product_chain(proof, challenge, claimed: int | None = None, closing: str = "auto") -> bool:
    """
    dsumcheck.rs:558-588 on rows (t0, t1, t2).  closing = "product": the last row is (0, f*g, 0)
    (`sumcheck_product`, :88) and must equal the last target; "none": every row is a round
    (`d_sumcheck_product` on the leader, Appendix A of SURVEY.md); "skip": a closing row is present but
    not comparable (c_sumcheck_product: pss2ss of the last values, :224-225,282); "auto": "product" if
    the last row has the (0, x, 0) shape.
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
    if closing == "product" and rounds:
        return rows[-1][1] == cur
    return True


def check_dhyperplonk_transcripts(n: int, res, pk, n_parties: int, leader: bool, echo: bool) -> list:
    """
    every sumcheck transcript of a dhyperplonk / dpermcheck result against its verifier chain; returns
    the list of failing labels (empty = all good).  c_sumcheck_product rows are share-level sums of this
    party's tables, d_sumcheck_product rows (leader only) the party-summed rounds plus log2(N_p) leader rounds.
    """
    (gate_proofs, _gate_comms), (w_proofs, _w_commits, _w_opens) = res
    bad = []
    for i, pr in enumerate(gate_proofs):  # c_sumcheck_product(.., challenge)  dhyperplonk.rs:223-260
        if not sumcheck_product_chain(pr, pk.challenge, closing="skip"):
            bad.append(f"gate[{i}]")
    if not sumcheck_product_chain(w_proofs[0], pk.challenge_r1, closing="skip"):  # 2.c
        bad.append("wiring[0]")
    if leader:
        s = n_parties.bit_length() - 1
        k = 1
        for j in range(3):  # 2.e.1 :411-413
            if not sumcheck_product_chain(w_proofs[k], pk.challenge_r2, closing="none"):
                bad.append(f"wiring[{k}]")
            k += 1
        for i in range(1, n - s + 1):  # layered sumchecks :417-478, challenge_r2[i..]
            for j in range(3):
                if not sumcheck_product_chain(w_proofs[k], pk.challenge_r2[i:], closing="none"):
                    bad.append(f"wiring[{k}] (layer {i})")
                k += 1
        for j in range(3):  # leader-tree sumchecks :506-508: plain sumcheck_product with the closing row
            if k < len(w_proofs) and not sumcheck_product_chain(w_proofs[k], pk.challenge_r2[:s], closing="product"):
                bad.append(f"wiring[{k}] (top tree)")
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
