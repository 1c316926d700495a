"""
Glue between the product's protocol drivers and the oracle's straight-line drivers (oracle/pyoracle.py `dhyperplonk_all`,
`dpermcheck_all`, `cpermcheck_all`): turn a host parameter set into the plain-int dictionaries the oracle takes, run the oracle with
the plain-C port doing the group operations, and compare two transcripts POSITION BY POSITION (the error names the first position
that differs: which list, which entry, which round / proof, which party).

Test infrastructure: imports oracle/, lives under tests/.
"""
import numpy as np

import coracle as co
import pyoracle as po
from oracle_backend import OracleBackend
from zkhip.field import random_fr, splitmix_fr

TABLE_NAMES = ("V", "a_evals", "b_evals", "c_evals", "I", "S1", "S2", "I_p", "S1_p", "S2_p", "ssigma", "ssigma_p", "sid", "sid_p", "eq", "eq_r1",
               "eq_r1_p", "eq_r2", "eq_r2_p")


# ---------------------------------------------------------------- limbs <-> ints
def fr_ints(a) -> list:
    """[k, 4] Montgomery limbs -> python ints (canonical values)"""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    if not len(a):
        return []
    raw = co.fr_from_mont(a).astype("<u8").tobytes()
    return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(len(a))]


def fr_limbs(xs) -> np.ndarray:
    """python ints -> [k, 4] Montgomery limbs"""
    xs = list(xs)
    if not xs:
        return np.zeros((0, 4), dtype=np.uint64)
    raw = b"".join((int(x) % po.R_MOD).to_bytes(32, "little") for x in xs)
    return co.fr_to_mont(np.frombuffer(raw, dtype="<u8").astype(np.uint64).reshape(-1, 4))


def points_of(bases12) -> list:
    """[k, 12] affine Montgomery limbs -> list of (x, y) ints / None"""
    b = np.ascontiguousarray(bases12, dtype=np.uint64).reshape(-1, 12)
    raw = co.fq_from_mont(b.reshape(-1, 6)).astype("<u8").tobytes()
    out = []
    for i in range(len(b)):
        if not b[i].any():
            out.append(None)
        else:
            out.append((int.from_bytes(raw[96 * i : 96 * i + 48], "little"), int.from_bytes(raw[96 * i + 48 : 96 * i + 96], "little")))
    return out


def point_limbs(P) -> np.ndarray:
    if P is None:
        return np.zeros(12, dtype=np.uint64)
    raw = int(P[0]).to_bytes(48, "little") + int(P[1]).to_bytes(48, "little")
    return co.fq_to_mont(np.frombuffer(raw, dtype="<u8").astype(np.uint64).reshape(2, 6)).reshape(12)


# ---------------------------------------------------------------- the plain-C port behind the oracle's three group operations
class CGroup:
    """msm / add / mul with pyoracle's signatures on the C port (zk_oracle.c); SRS levels are converted to limbs once"""

    def __init__(self):
        self.levels = {}

    def register(self, points: list, limbs12: np.ndarray):
        self.levels[id(points)] = (points, np.ascontiguousarray(limbs12, dtype=np.uint64).reshape(-1, 12))

    def msm(self, bases, scalars):
        if len(bases) != len(scalars):
            raise ValueError(min(len(bases), len(scalars)))
        if not len(bases):
            return None
        ent = self.levels.get(id(bases))
        b = ent[1] if ent is not None and ent[0] is bases else np.stack([point_limbs(P) for P in bases])
        return points_of(co.msm_g1(b, fr_limbs(scalars)))[0]

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        return points_of(co.g1_add_affine(point_limbs(P), point_limbs(Q)))[0]

    def mul(self, P, k):
        k %= po.R_MOD
        if P is None or k == 0:
            return None
        kk = np.frombuffer(int(k).to_bytes(32, "little"), dtype="<u8").astype(np.uint64)
        return points_of(co.g1_mul_affine(point_limbs(P), kk))[0]

    def backend(self):
        return po.g1_backend(msm=self.msm, add=self.add, mul=self.mul)


# ---------------------------------------------------------------- a host parameter set as the oracle's plain ints
def pk_ints(pk, grp: CGroup) -> dict:
    """a zkhip PackedProvingParameters built on OracleBackend -> the dict pyoracle's drivers take (dhyperplonk.rs:22-62 field names)"""
    d = {}
    for name in TABLE_NAMES + ("mask", "unmask0", "unmask1", "unmask2"):
        if name in pk.tables:
            d[name] = fr_ints(pk.tables[name].a[: pk.lens[name]])
    for name in ("challenge", "challenge_r1", "challenge_r2"):
        d[name] = fr_ints(getattr(pk, name))
    for name in ("alpha", "beta", "gamma"):
        d[name] = fr_ints(getattr(pk, name))[0]
    for name in ("c_commitment", "d_commitment"):
        levels = []
        for srs in getattr(pk, name):
            pts = points_of(srs.bases)
            grp.register(pts, srs.bases)
            levels.append(pts)
        d[name] = levels
    return d


def per_run_ints(pk, pp, npar: int, n: int, seed: int, data_parallel: bool = False) -> dict:
    """the "Jump from sky" tables as zkhip.hyperplonk._per_run_data draws (or carries) them -> ints"""
    M, l = 1 << n, pp.l
    T = pk.tables
    if "local_s_p" in T:  # the SplitMix64 parameter set carries them
        run = {"local_s_p": fr_ints(T["local_s_p"].a), "local_s": fr_ints(T["local_s_l"].a), "eq": fr_ints(T["eq_top"].a)}
        if data_parallel:
            run["s"] = fr_ints(T["s_data_parallel"].a)
        return run
    run = {"local_s_p": fr_ints(random_fr(4 * M // npar, seed * 31 + 1)), "local_s": fr_ints(random_fr(4 * M // npar // l, seed * 31 + 2)),
           "eq": fr_ints(random_fr(pp.n, seed * 31 + 3))}
    if data_parallel:
        run["s"] = fr_ints(random_fr(4 * M // l, seed * 31 + 4))
    return run


def cperm_masks(pk, be, n: int, l: int, seed: int, splitmix: bool = False):
    """the four mask tables of cpermcheck as the hosts draw them (zkhip.hyperplonk.cpermcheck / host/examples/hyperplonk.cpp)"""
    G4 = 4 * ((1 << n) // l)
    for i, name in enumerate(("mask", "unmask0", "unmask1", "unmask2")):
        if name not in pk.tables:
            pk.tables[name] = be.to_device(splitmix_fr(G4, seed * 977 + 50 + i) if splitmix else random_fr(G4, seed * 977 + 50 + i))
            pk.lens[name] = G4


# ---------------------------------------------------------------- comparison, position by position
def _norm_point(j):
    """a host point (normalised Jacobian [18], or affine [12]) -> affine Montgomery [12]"""
    j = np.asarray(j, dtype=np.uint64).reshape(-1)
    if j.size == 12:
        return j
    if not j[12:].any():
        return np.zeros(12, dtype=np.uint64)
    one = np.array(po.fq_to_mont_limbs(1), dtype=np.uint64)
    assert (j[12:] == one).all(), "host points must be normalised"
    return j[:12]


def _cmp_proof(host, want, where):
    host = np.asarray(host, dtype=np.uint64)
    rows = len(want)
    assert host.reshape(-1, 4).shape[0] == sum(len(r) for r in want), f"{where}: {host.shape} against {rows} rounds"
    if rows == 0:
        return
    width = len(want[0])
    exp = fr_limbs([v for row in want for v in row]).reshape(rows, width, 4)
    got = host.reshape(rows, width, 4)
    bad = np.argwhere((got != exp).any(axis=2))
    assert not len(bad), f"{where}: round {bad[0][0]}, component {bad[0][1]} differs ({len(bad)} of {rows * width} entries)"


def _cmp_point(host, want, where):
    assert (_norm_point(host) == point_limbs(want)).all(), f"{where}: point differs"


def _cmp_open(host, want, where):
    val, proofs = host
    assert (np.asarray(val, dtype=np.uint64).reshape(-1)[:4] == fr_limbs([want[0]])[0]).all(), f"{where}: value differs"
    proofs = np.asarray(proofs, dtype=np.uint64)
    got = proofs.reshape(-1, proofs.shape[-1]) if proofs.size else np.zeros((0, 18), dtype=np.uint64)
    assert len(got) == len(want[1]), f"{where}: {len(got)} proofs against {len(want[1])}"
    for i, P in enumerate(want[1]):
        _cmp_point(got[i], P, f"{where}: proof {i}")


def compare_wiring(host, want, who):
    hp, hc, ho = host
    wp, wc, wo = want
    assert len(hp) == len(wp), f"{who}: {len(hp)} wiring proofs against {len(wp)}"
    assert len(hc) == len(wc), f"{who}: {len(hc)} wiring commits against {len(wc)}"
    assert len(ho) == len(wo), f"{who}: {len(ho)} wiring opens against {len(wo)}"
    for i, (a, b) in enumerate(zip(hp, wp)):
        _cmp_proof(a, b, f"{who}: wiring_proofs[{i}]")
    for i, (a, b) in enumerate(zip(hc, wc)):
        _cmp_point(a, b, f"{who}: wiring_commits[{i}]")
    for i, (a, b) in enumerate(zip(ho, wo)):
        _cmp_open(a, b, f"{who}: wiring_opens[{i}]")


def compare_dhyperplonk(host, want, who):
    (hg, hgc), hw = host
    (wg, wgc), ww = want
    assert len(hg) == len(wg) == 6 and len(hgc) == len(wgc) == 6, f"{who}: gate lists"
    for i, (a, b) in enumerate(zip(hg, wg)):
        _cmp_proof(a, b, f"{who}: gate_identity_proofs[{i}]")
    for i, ((hcom, hop), (wcom, wop)) in enumerate(zip(hgc, wgc)):
        _cmp_point(hcom, wcom, f"{who}: gate_identity_commitments[{i}].0")
        _cmp_open(hop, wop, f"{who}: gate_identity_commitments[{i}].1")
    compare_wiring(hw, ww, who)


# ---------------------------------------------------------------- the oracle side of a run
def oracle_run(which: str, n: int, pp_l: int, seeds, chal_seed, run_seeds, comm: bool = True, splitmix: bool = False, mask_seeds=None):
    """
    build the parties' parameter sets exactly as the hosts do (same seeds -> same tables, challenges and synthetic SRS), as plain
    ints, and run the oracle's straight-line driver.  seeds[p] / run_seeds[p]: party p's table seed / per-run seed.
    -> the per-party list the *_all driver returns (one entry in echo mode).
    """
    from zkhip.hyperplonk import PackedProvingParameters
    from zkhip.pss import PackedSharingParams

    pp_host = PackedSharingParams(pp_l)
    pp = po.PackedSharingParams(pp_l)
    be = OracleBackend()
    grp = CGroup()
    parties = range(pp.n if comm else 1)
    pks, runs = [], []
    for p in parties:
        make = PackedProvingParameters.new_splitmix if splitmix else PackedProvingParameters.new
        pk = make(n, pp_host, be, seed=seeds[p], chal_seed=chal_seed, window_tables=False)
        if which == "cpermcheck":
            cperm_masks(pk, be, n, pp_l, (mask_seeds or run_seeds)[p], splitmix)
        pks.append(pk_ints(pk, grp))
        runs.append(per_run_ints(pk, pp_host, pp.n, n, run_seeds[p], which == "data-parallel"))
    with grp.backend():
        if which == "cpermcheck":
            return po.cpermcheck_all(n, pks, pp, comm=comm)
        if which == "dpermcheck":
            return po.dpermcheck_all(n, pks, pp, runs, comm=comm)
        return po.dhyperplonk_all(n, pks, pp, runs, data_parallel=which == "data-parallel", comm=comm)
