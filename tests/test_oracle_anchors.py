"""
Pins the oracle: (1) every constant re-derived from first principles, (2) the reference's own
known-answer tests that touch the hot path.  (MSM / sumcheck / PSS have no golden vectors in
the reference -- see the "parity unpinned" note in oracle/pyoracle.py.)
"""
import pyoracle as po


def _is_probable_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_moduli_and_montgomery_constants():
    assert _is_probable_prime(po.R_MOD) and po.R_MOD.bit_length() == 255
    assert _is_probable_prime(po.Q_MOD) and po.Q_MOD.bit_length() == 381
    assert (-pow(po.R_MOD, -1, 1 << 64)) % (1 << 64) == 0xFFFFFFFEFFFFFFFF
    assert (-pow(po.Q_MOD, -1, 1 << 64)) % (1 << 64) == 0x89F3FFFCFFFCFFFD
    assert (1 << 256) % po.R_MOD == 0x1824B159ACC5056F998C4FEFECBC4FF55884B7FA0003480200000001FFFFFFFE
    # 2-adicity 32, 7 generates the multiplicative group's 2-part
    assert (po.R_MOD - 1) % (1 << 32) == 0 and (po.R_MOD - 1) % (1 << 33) != 0
    assert pow(7, (po.R_MOD - 1) // 2, po.R_MOD) == po.R_MOD - 1  # 7 is a non-residue
    w = po.FR_ROOT_OF_UNITY
    assert w == 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B
    assert pow(w, 1 << 32, po.R_MOD) == 1 and pow(w, 1 << 31, po.R_MOD) != 1


def test_g1_generator_and_subgroup():
    assert po.g1_is_on_curve(po.G1_GEN)
    assert po.g1_mul(po.G1_GEN, po.R_MOD - 1) == po.g1_neg(po.G1_GEN)
    assert po.g1_add(po.g1_mul(po.G1_GEN, po.R_MOD - 1), po.G1_GEN) is None  # r*G = O
    two_g = po.g1_add(po.G1_GEN, po.G1_GEN)
    assert po.g1_is_on_curve(two_g) and po.g1_mul(po.G1_GEN, 2) == two_g


def test_reference_kat_sub_index():
    """dacc_product.rs:442-448"""
    assert po.sub_index(26) == (20, 21)


def test_reference_kat_acc_product():
    """
    dacc_product.rs:450-466 asserts ([1,3,2,24],[2,4,12,0],[2,12,24,0]); those are the outputs
    for input [1,2,3,4] (the test feeds 1..=8, for which the code at :30-57 returns the
    8-element vectors below -- the in-tree assertion is stale).
    """
    assert po.acc_product([1, 2, 3, 4]) == ([1, 3, 2, 24], [2, 4, 12, 0], [2, 12, 24, 0])
    assert po.acc_product(list(range(1, 9))) == (
        [1, 3, 5, 7, 2, 30, 24, 40320],
        [2, 4, 6, 8, 12, 56, 1680, 0],
        [2, 12, 30, 56, 24, 1680, 40320, 0],
    )


def test_reference_kat_transpose():
    """utils/operator.rs:42-49"""
    assert po.transpose([[1, 2, 3], [4, 5, 6], [7, 8, 9]]) == [[1, 4, 7], [2, 5, 8], [3, 6, 9]]


def test_compressed_encoding_shape():
    """48-byte zcash-style G1 encoding; 8+48 = the 56-byte Vec<G1> messages of hack/run-hyperplonk/output.txt:25"""
    b = po.g1_compress(po.G1_GEN)
    assert len(b) == 48 and b[0] & 0x80 and not b[0] & 0x40
    assert b.hex().startswith("97f1d3a73197d7942695638c4fa9ac0f")  # generator, well-known encoding
    assert po.g1_compress(None)[0] == 0xC0
    assert len(po.fr_serialize(5)) == 32 and po.fr_serialize(5)[0] == 5


def test_g2_generator_anchor():
    """the G2 generator of the oracle: on the twist y^2 = x^3 + 4(1 + u) and of order r"""
    import pyoracle as po

    assert po.g2_is_on_curve(po.G2_GEN)
    assert po.g2_mul(po.G2_GEN, po.R_MOD - 1) == po.g2_neg(po.G2_GEN)
    assert po.g2_add(po.g2_mul(po.G2_GEN, po.R_MOD - 1), po.G2_GEN) is None
    P = po.g2_mul(po.G2_GEN, 123456789)
    assert po.g2_is_on_curve(P) and po.g2_add(P, po.g2_neg(P)) is None
    assert po.g2_add(po.g2_mul(po.G2_GEN, 5), po.g2_mul(po.G2_GEN, 7)) == po.g2_mul(po.G2_GEN, 12)
