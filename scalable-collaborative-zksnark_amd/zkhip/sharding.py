"""
Strong-scaling ONE primitive across G GPUs.  One process per GPU; every
function here has exactly one exchange step (an all-gather of a few hundred bytes) because the
partial results combine linearly:

  sharded_msm               contiguous chunks of (base, scalar) pairs per rank (the SRS chunk stays
                            resident); G partial points are all-gathered and added.
  sharded_sumcheck(_product) CYCLIC layout: global index i lives on rank i mod G at local slot
                            i div G.  Pairs (j, j + m/2) stay on one rank while m/2 >= G, so each
                            rank runs the unmodified kernel for log2(N/G) rounds with challenge[0..];
                            the per-round sums add up across ranks, and the G leftovers (one per rank,
                            rank r holding global index r) finish the last log2(G) rounds.  The
                            transcript is bit-identical to the monolithic sumcheck(_product) on the
                            full table (dsumcheck.rs:6-26, 28-90) -- unlike d_sumcheck, whose
                            contiguous chunks consume the variables in a different order.
"""
from __future__ import annotations

import numpy as np

from .dist_primitive import _fr_vec_to_ints, _ints_to_fr, _round_plain, _round_product
from .field import R_MOD, fr_from_mont, int_to_limbs
from .net import Net


def sharded_msm(be, srs_chunk, scalars_chunk, n_local: int, net: Net) -> np.ndarray:
    """sum over ranks of MSM(bases_chunk, scalars_chunk): [18] normalised Jacobian on every rank"""
    local = be.msm_g1(srs_chunk, scalars_chunk, n_local)
    pts = np.stack(net.all_gather(local))
    ones = np.tile(int_to_limbs(1, 4), (net.n_parties, 1))
    return be.g1_lincomb(pts, ones)


def cyclic_shard(table: np.ndarray, rank: int, world: int) -> np.ndarray:
    """the slice of a full table [N,4] owned by `rank` under the cyclic layout"""
    return np.ascontiguousarray(np.asarray(table).reshape(-1, 4)[rank::world])


def sharded_sumcheck(be, local_tab, local_len: int, challenge: np.ndarray, net: Net) -> np.ndarray:
    """== sumcheck(full_table, challenge): [n+1, 2, 4] on every rank"""
    G = net.n_parties
    nl = local_len.bit_length() - 1
    g = G.bit_length() - 1
    challenge = np.asarray(challenge, dtype=np.uint64).reshape(-1, 4)
    pairs, last = be.sumcheck(local_tab, local_len, challenge[:nl])
    allp = net.all_gather(np.concatenate([pairs.reshape(-1, 4), np.asarray(last).reshape(1, 4)]))
    res = []
    for i in range(nl):
        res.append(tuple(sum(fr_from_mont(allp[r][2 * i + k]) for r in range(G)) % R_MOD for k in range(2)))
    v = [fr_from_mont(allp[r][2 * nl]) for r in range(G)]
    ch = _fr_vec_to_ints(challenge)
    for i in range(nl, nl + g):
        s, v = _round_plain(v, ch[i])
        res.append(s)
    res.append((0, v[0]))
    return _ints_to_fr([x for p in res for x in p]).reshape(-1, 2, 4)


def sharded_sumcheck_product(be, local_f, local_g, local_len: int, challenge: np.ndarray, net: Net) -> np.ndarray:
    """== sumcheck_product(full_f, full_g, challenge): [n+1, 3, 4] on every rank"""
    G = net.n_parties
    nl = local_len.bit_length() - 1
    g = G.bit_length() - 1
    challenge = np.asarray(challenge, dtype=np.uint64).reshape(-1, 4)
    tr, lf, lg = be.sumcheck_product(local_f, local_g, local_len, challenge[:nl])
    allp = net.all_gather(np.concatenate([tr.reshape(-1, 4), np.asarray(lf).reshape(1, 4), np.asarray(lg).reshape(1, 4)]))
    res = []
    for i in range(nl):
        res.append(tuple(sum(fr_from_mont(allp[r][3 * i + k]) for r in range(G)) % R_MOD for k in range(3)))
    f = [fr_from_mont(allp[r][3 * nl]) for r in range(G)]
    gg = [fr_from_mont(allp[r][3 * nl + 1]) for r in range(G)]
    ch = _fr_vec_to_ints(challenge)
    for i in range(nl, nl + g):
        t, f, gg = _round_product(f, gg, ch[i])
        res.append(t)
    res.append((0, f[0] * gg[0] % R_MOD, 0))
    return _ints_to_fr([x for t in res for x in t]).reshape(-1, 3, 4)
