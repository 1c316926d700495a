"""
The host pairing behind PolynomialCommitment::verify (zkhip/pairing.py; dist-primitive/src/dpoly_comm.rs:466-484)
against the reference's own tests: `should_pair` (:495-500) and `should_commit_and_open` (:502-531), the latter on
the ORACLE's commit / open over the oracle's structured SRS (CPU), and on the library's (GPU).
"""
import numpy as np
import pytest

import pyoracle as po
from zkhip import pairing as pr


def test_should_pair_bilinearity():
    rng = po.SplitMix64(11)
    a, b, s = rng.fr(), rng.fr(), rng.fr()
    g1, g2 = pr.g1_mul(pr.G1_GEN, a), pr.g2_mul(pr.G2_GEN, b)
    e = pr.pairing(g2, g1)
    assert not (e == pr.Fq12.one()) and (e ** pr.R_MOD) == pr.Fq12.one()
    assert pr.pairing(pr.g2_mul(g2, s), g1) == pr.pairing(g2, pr.g1_mul(g1, s)) == e ** s  # dpoly_comm.rs:499
    assert pr.pairing(None, g1) == pr.Fq12.one() and pr.pairing(g2, None) == pr.Fq12.one()
    # the small curve arithmetic of the module agrees with the oracle's
    assert pr.g1_mul(pr.G1_GEN, a) == po.g1_mul(po.G1_GEN, a) and pr.g2_mul(pr.G2_GEN, b) == po.g2_mul(po.G2_GEN, b)


def test_should_commit_and_open_on_the_oracle():
    rng = po.SplitMix64(12)
    n = 4
    s, u, poly = rng.fr_vec(n), rng.fr_vec(n), rng.fr_vec(1 << n)
    g1, g2 = po.g1_mul(po.G1_GEN, rng.fr()), po.g2_mul(po.G2_GEN, rng.fr())
    levels = po.srs_powers(g1, s)
    C = po.commit(levels, poly)
    value, proof = po.open_(levels, poly, u)
    pg2 = pr.powers_of_g2(s, g2)
    assert pr.verify(g1, pg2, C, value, proof, u)
    assert not pr.verify(g1, pg2, C, (value + 1) % po.R_MOD, proof, u)
    bad = list(proof)
    bad[1] = po.g1_add(bad[1], g1)
    assert not pr.verify(g1, pg2, C, value, bad, u)
    assert not pr.verify(g1, pg2, C, value, proof, u[::-1])


@pytest.mark.gpu
def test_library_commit_open_verifies_with_the_pairing(ctx):
    """commit + open on the GPU against zk_srs_powers, checked by the verifier's pairing equation"""
    from helpers import jac_norm_to_affine, pt_ints
    from zkhip import dist_primitive as dp

    rng = po.SplitMix64(13)
    n = 8
    s, u, poly = rng.fr_vec(n), rng.fr_vec(n), rng.fr_vec(1 << n)
    mont = lambda xs: np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)
    cub = dp.PolynomialCommitmentCub.new(ctx, mont(s))
    d = ctx.to_device(mont(poly))
    C = pt_ints(jac_norm_to_affine(dp.commit(ctx, cub.mature(), d, 1 << n)))
    value, proofs = dp.open_(ctx, cub.mature(), d, 1 << n, mont(u))
    v = po.fr_from_mont_limbs(value)
    pis = [pt_ints(jac_norm_to_affine(p)) for p in proofs]
    pg2 = pr.powers_of_g2(s)
    assert pr.verify(pr.G1_GEN, pg2, C, v, pis, u)
    assert not pr.verify(pr.G1_GEN, pg2, C, (v + 1) % po.R_MOD, pis, u)
    # the reference-shaped entry point on the library's own array formats
    assert dp.verify(pg2, dp.commit(ctx, cub.mature(), d, 1 << n), value, proofs, mont(u))
