"""Structured SRS, packed SRS and the PSS maps on G1 points, all on the device, against the oracle's literal
restatement of PolynomialCommitmentCub::new / to_packed (dpoly_comm.rs:37-67,164-194) and pss.rs:93-171."""
import numpy as np
import pytest

import pyoracle as po
from helpers import jac_norm_to_affine, pt_ints, pt_mont, rand_fr  # noqa: F401

pytestmark = pytest.mark.gpu


def _canon(xs):
    return np.array([[(int(x) >> (64 * i)) & (2**64 - 1) for i in range(4)] for x in xs], dtype=np.uint64)


def _mont(xs):
    return np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def test_srs_powers_matches_reference_construction(ctx):
    rng = po.SplitMix64(2024)
    n = 5
    s = rng.fr_vec(n)
    exp = po.srs_powers(po.G1_GEN, s)
    levels = ctx.srs_powers(_mont(s))
    assert len(levels) == n + 1
    for k in range(n + 1):
        got = levels[k].download()
        assert [pt_ints(r) for r in got] == exp[k], f"level {k}"
    # another base, and a commitment against the structured level equals the oracle MSM
    g = po.g1_mul(po.G1_GEN, 0xDEADBEEF)
    lv = ctx.srs_powers(_mont(s[:3]), g=pt_mont(g))
    assert [pt_ints(r) for r in lv[3].download()] == po.srs_powers(g, s[:3])[3]
    sc = rng.fr_vec(1 << n)
    got = ctx.msm_g1(levels[n], ctx.to_device(_mont(sc)), 1 << n)
    assert pt_ints(jac_norm_to_affine(got)) == po.g1_msm(exp[n], sc)


def test_srs_powers_large_is_a_valid_eq_basis(ctx, co):
    """2^16 points: sum of the level = g (the eq weights sum to 1) and a random linear check against the exponents"""
    rng = po.SplitMix64(5)
    n = 16
    s = rng.fr_vec(n)
    levels = ctx.srs_powers(_mont(s))
    ones = ctx.to_device(_mont([1] * (1 << n)))
    for k in (n - 1, n):
        got = ctx.msm_g1(levels[k], ones, 1 << k)
        assert pt_ints(jac_norm_to_affine(got)) == po.G1_GEN
    # <level_n, t> == g^{sum_j E_n[j] t_j}: E_n[j] = prod_i (s_i if bit else 1 - s_i); index bit (n-1-i) <-> s_i ... checked on 16 sampled j
    E = [1]
    for i in range(n):
        sv = s[n - i - 1]
        E = [e * (1 - sv) % po.R_MOD for e in E] + [e * sv % po.R_MOD for e in E]
    pts = levels[n].download()
    for j in (0, 1, 2, 12345, (1 << n) - 1):
        assert pt_ints(pts[j]) == po.g1_mul(po.G1_GEN, E[j])


@pytest.mark.parametrize("l", [1, 2, 4])
def test_srs_to_packed_matches_reference(ctx, l):
    from zkhip.pss import PackedSharingParams

    rng = po.SplitMix64(77 + l)
    n = 4
    s = rng.fr_vec(n)
    exp_levels = po.srs_powers(po.G1_GEN, s)
    opp, pp = po.PackedSharingParams(l), PackedSharingParams(l)
    exp = po.srs_to_packed(exp_levels, opp)
    levels = ctx.srs_powers(_mont(s))
    for party in (0, 3, pp.n - 1):
        row = _canon([pp.pack_matrix[party][j] for j in range(l)])
        for k in range(n + 1):
            got = ctx.srs_to_packed(levels[k], row, l).download()
            assert [pt_ints(r) for r in got] == exp[party][k], (party, k)


@pytest.mark.parametrize("l", [1, 2])
def test_g1_pss_maps_on_device(ctx, l):
    """pack_from_public / unpack2 on vectors of points through zk_g1_apply_matrix == the oracle's group FFTs"""
    from zkhip.pss import PackedSharingParams

    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    k = 5
    pts = po.g1_bases(k * pp.n, 9) + []
    pts[3] = None  # an infinity among the inputs
    d_in = ctx.to_device(np.array([pt_mont(P) for P in pts]))
    # unpack2: vectors of n shares (contiguous) -> l secrets each
    m = _canon([v for row in pp.unpack2_matrix for v in row]).reshape(l, pp.n, 4)
    out = ctx.g1_apply_matrix(m, d_in, pp.n, 1, k, l, 1).download((k * l, 12))
    for j in range(k):
        exp = opp.unpack2_g1(pts[j * pp.n : (j + 1) * pp.n])
        assert [pt_ints(r) for r in out[j * l : (j + 1) * l]] == exp
    # pack_from_public: vectors of l secrets -> n shares, written party-major (out[p*k + j])
    secrets = pts[: k * l]
    d_s = ctx.to_device(np.array([pt_mont(P) for P in secrets]))
    mp = _canon([pp.pack_matrix[p][j] for p in range(pp.n) for j in range(l)]).reshape(pp.n, l, 4)
    out = ctx.g1_apply_matrix(mp, d_s, l, 1, k, 1, k).download((pp.n * k, 12))
    for j in range(k):
        exp = opp.pack_from_public_g1(secrets[j * l : (j + 1) * l])
        assert [pt_ints(out[p * k + j]) for p in range(pp.n)] == exp


def _oracle_map(co, matrix_rows, vec):
    """sum_c M[r][c] * P_c per row through the C oracle's group arithmetic (matrix: python ints, vec: [cols, 12] affine Montgomery)"""
    return [pt_ints(co.msm_g1(vec, _mont(row))) for row in matrix_rows]


@pytest.mark.parametrize("l", [8, 16])
def test_g1_pss_maps_on_device_at_64_and_128_parties(ctx, co, l):
    """
    pack_from_public / unpack / unpack2 on vectors of POINTS (pss.rs:93-171 with G: DomainCoeff) at the party counts the reference
    sweeps (handle_server.sh:26-31: l up to 32): zk_g1_apply_matrix with the host mirror's matrices against the ORACLE's
    matrices (derived from its own domain code) applied with the C oracle's group law; at l = 8 one vector also against the
    oracle's literal group FFTs.  Both kernel forms (one lane per output / one lane per term) must give the same bits; the time of
    the leader's map of d_msm (dmsm.rs:30-39: unpack2 of 8l points into l) is printed for both.
    """
    import time

    from helpers import synthetic_bases
    from zkhip.pss import PackedSharingParams

    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    k = 3
    pts = synthetic_bases(k * pp.n, 40 + l)[0].copy()
    pts[5] = 0  # an infinity among the inputs
    d_in = ctx.to_device(pts)
    o_unpack2, o_unpack, o_pack = opp.unpack2_matrix(), opp.unpack_matrix(), opp.pack_matrix()
    times = {}
    for name, mine, theirs in (("unpack2", pp.unpack2_matrix, o_unpack2), ("unpack", pp.unpack_matrix, o_unpack)):
        m = _canon([v for row in mine for v in row]).reshape(l, pp.n, 4)
        outs = {}
        for form in (1, 0):
            ctx.dbg_tune("g1_map_by_column", form)
            try:
                ctx.g1_apply_matrix(m, d_in, pp.n, 1, k, l, 1)  # (first call of a form: allocations)
                ctx.sync()
                t0 = time.perf_counter()
                outs[form] = ctx.g1_apply_matrix(m, d_in, pp.n, 1, k, l, 1).download((k * l, 12))
                times[(name, form)] = (time.perf_counter() - t0) * 1e3
            finally:
                ctx.dbg_tune("g1_map_by_column", 1)
        assert (outs[0] == outs[1]).all(), name
        for j in range(k):
            exp = _oracle_map(co, theirs, pts[j * pp.n : (j + 1) * pp.n])
            assert [pt_ints(r) for r in outs[1][j * l : (j + 1) * l]] == exp, (name, j)
    if l == 8:  # the definition itself: ifft on the share domain, fft on the secret2 coset, every second slot
        got = ctx.g1_apply_matrix(_canon([v for row in pp.unpack2_matrix for v in row]).reshape(l, pp.n, 4), d_in, pp.n, 1, 1, l, 1).download((l, 12))
        assert [pt_ints(r) for r in got] == opp.unpack2_g1([pt_ints(r) for r in pts[: pp.n]])
    # pack_from_public: vectors of l secrets -> n shares, written party-major (out[p*k + j])
    secrets = pts[: k * l]
    d_s = ctx.to_device(secrets)
    mp = _canon([pp.pack_matrix[p][j] for p in range(pp.n) for j in range(l)]).reshape(pp.n, l, 4)
    out = ctx.g1_apply_matrix(mp, d_s, l, 1, k, 1, k).download((pp.n * k, 12))
    for j in range(k):
        exp = _oracle_map(co, [row[:l] for row in o_pack], secrets[j * l : (j + 1) * l])
        assert [pt_ints(out[p * k + j]) for p in range(pp.n)] == exp
    print(f"\nzk_g1_apply_matrix at l = {l} ({pp.n} parties), {k} vectors, ms per call: " + ", ".join(f"{n} {'per-term' if f else 'per-output'} {t:.2f}" for (n, f), t in sorted(times.items())))
    assert times[("unpack2", 1)] < 10.0 * k, "the leader's point map must stay below 10 ms per item"


@pytest.mark.parametrize("l", [8, 16])
def test_srs_to_packed_at_64_and_128_parties(ctx, co, l):
    """PolynomialCommitmentCub::to_packed (dpoly_comm.rs:164-194) for single parties at l = 8, 16: every l-chunk of a level -> pack_from_public"""
    from zkhip.pss import PackedSharingParams

    rng = po.SplitMix64(300 + l)
    n = 6
    s = rng.fr_vec(n)
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    o_pack = opp.pack_matrix()
    levels = ctx.srs_powers(_mont(s))
    for party in (0, 5, pp.n - 1):
        row = _canon([pp.pack_matrix[party][j] for j in range(l)])
        for k in (n - 1, n):  # levels of 2^k points: 2^k / l chunks (the short levels pad with infinity, exercised at l = 1, 2, 4)
            pts = levels[k].download()
            got = ctx.srs_to_packed(levels[k], row, l).download()
            assert len(got) == max(1, (1 << k) // l)
            for c in range(len(got)):
                chunk = pts[c * l : (c + 1) * l]
                if len(chunk) < l:
                    chunk = np.concatenate([chunk, np.zeros((l - len(chunk), 12), dtype=np.uint64)])
                assert pt_ints(got[c]) == _oracle_map(co, [o_pack[party][:l]], chunk)[0], (party, k, c)


def test_d_msm_with_64_real_party_threads(ctx, co):
    """
    d_msm (dmsm.rs:9-43) at l = 8 with all 64 parties as threads of this process, a ctx each on GPU 0 -- no echo shortcut: every
    party's local MSM, the all-gather of the 64 results, and the public map on the gathered points.  The reference's identity
    (dmsm.rs:92-138): packed bases x packed scalars -> the shares of [MSM; l], i.e. unpack over the 64 outputs == the plain MSM.
    """
    import zkhip
    from helpers import synthetic_bases
    from zkhip import dist_primitive as dp
    from zkhip.net import LocalTestNet
    from zkhip.pss import PackedSharingParams

    l, m = 8, 64  # m secrets -> m / l packed chunks per party
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    o_pack = opp.pack_matrix()
    bases = synthetic_bases(m, 71)[0]
    rng = po.SplitMix64(72)
    scal = rng.fr_vec(m)
    plain = co.msm_g1(bases, _mont(scal))
    # party p's share of chunk c: pack_from_public of the chunk's l bases (points) and of its l scalars
    sh_b = [np.array([pt_mont(_oracle_map(co, [o_pack[p][:l]], bases[c * l : (c + 1) * l])[0]) for c in range(m // l)]) for p in range(pp.n)]
    sh_s = [[] for _ in range(pp.n)]
    for c in range(m // l):
        for p, v in enumerate(opp.pack_from_public(list(scal[c * l : (c + 1) * l]))):
            sh_s[p].append(v)

    def party(net):
        be = zkhip.Ctx(0)
        try:
            srs = be.srs_register(sh_b[net.party_id])
            return dp.d_msm(be, [srs], [be.to_device(_mont(sh_s[net.party_id]))], [m // l], pp, net)[0]
        finally:
            be.close()

    outs = LocalTestNet.simulate_network_round(pp.n, party)
    shares = np.array([jac_norm_to_affine(o) for o in outs])
    got = _oracle_map(co, opp.unpack_matrix(), shares)  # unpack over the parties' outputs: [MSM; l]
    assert got == [pt_ints(plain)] * l


def test_packed_commitment_shares_recombine(ctx):
    """
    the reference's own property (dpoly_comm.rs:502-531 / dmsm.rs:92-138): c_commit against the PACKED structured
    SRS with packed scalar shares, unpack2 over the parties' results == commit of the plain polynomial
    """
    from zkhip import dist_primitive as dp
    from zkhip.net import LocalTestNet
    from zkhip.pss import PackedSharingParams

    l, n = 2, 5
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    rng = po.SplitMix64(99)
    s, poly = rng.fr_vec(n), rng.fr_vec(1 << n)
    cub = dp.PolynomialCommitmentCub.new(ctx, _mont(s))
    plain = dp.commit(ctx, cub.mature(), ctx.to_device(_mont(poly)), 1 << n)
    # scalar shares: pack every l-chunk, party p collects share p
    sh = [[] for _ in range(pp.n)]
    for k in range(0, 1 << n, l):
        for p, v in enumerate(pp.pack_from_public(poly[k : k + l])):
            sh[p].append(v)

    import zkhip

    def party(net):
        be = zkhip.Ctx(0)  # one ctx per party thread (calls on one ctx are not concurrent)
        try:
            packed = cub.to_packed(be, pp, net.party_id).mature()
            return dp.c_commit(be, packed, [be.to_device(_mont(sh[net.party_id]))], [(1 << n) // l], pp, net)[0]
        finally:
            be.close()

    outs = LocalTestNet.simulate_network_round(pp.n, party)
    # every party's output is its share of the packed result [C; l]: unpack recovers C in every slot
    got = opp.unpack_g1([pt_ints(jac_norm_to_affine(o)) for o in outs])
    assert got == [pt_ints(jac_norm_to_affine(plain))] * l


@pytest.mark.parametrize("n", [4, 10, 20])
def test_commit_open_pairs_satisfy_the_verifier_equation(ctx, n):
    """
    should_commit_and_open (dpoly_comm.rs:502-531): commit + open against the structured SRS must satisfy
    verify's pairing equation (:466-484); checked pulled back to G1 through the known trapdoor, with the
    ORACLE's group law: C - v g == sum_i (s_i - u_i) pi_i.
    """
    from zkhip import dist_primitive as dp
    from zkhip.verify import open_equation_terms

    rng = po.SplitMix64(4242 + n)
    s, u = rng.fr_vec(n), rng.fr_vec(n)
    cub = dp.PolynomialCommitmentCub.new(ctx, _mont(s))
    if n <= 12:
        poly = rng.fr_vec(1 << n)
        d_poly = ctx.to_device(_mont(poly))
    else:  # full size (2^20 evaluations, 2^21 structured SRS points): the equation alone, no big-int evaluation of the polynomial
        from helpers import rand_fr

        poly = None
        d_poly = ctx.to_device(rand_fr(1 << n, 4343))
    C = pt_ints(jac_norm_to_affine(dp.commit(ctx, cub.mature(), d_poly, 1 << n)))
    value, proofs = dp.open_(ctx, cub.mature(), d_poly, 1 << n, _mont(u))
    v = po.fr_from_mont_limbs(value)
    # the opened value is the multilinear extension at u (variable 0 = the top index bit, folded first)
    if poly is not None:
        tab = list(poly)
        for r in u:
            tab = po.fold(tab, r)
        assert tab == [v]
    coeffs = open_equation_terms(value, _mont(u), _mont(s))
    rhs = None
    for c, pi in zip(coeffs, proofs):
        rhs = po.g1_add(rhs, po.g1_mul(pt_ints(jac_norm_to_affine(pi)), c))
    lhs = po.g1_add(C, po.g1_neg(po.g1_mul(po.G1_GEN, v)))
    assert lhs == rhs
    # a wrong value must break it
    assert po.g1_add(C, po.g1_neg(po.g1_mul(po.G1_GEN, (v + 1) % po.R_MOD))) != rhs
