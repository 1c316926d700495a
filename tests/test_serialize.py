"""arkworks encodings (zkhip/serialize.py) against the oracle's independent encoder and round trips"""
import pyoracle as po
from zkhip import serialize as ser


def test_fr_and_vec_roundtrip():
    rng = po.SplitMix64(1)
    xs = rng.fr_vec(9) + [0, 1, po.R_MOD - 1]
    assert all(ser.fr_serialize(x) == po.fr_serialize(x) for x in xs)
    blob = ser.fr_vec_serialize(xs)
    assert len(blob) == 8 + 32 * len(xs) and ser.fr_vec_deserialize(blob) == xs


def test_g1_compressed_matches_oracle_and_roundtrips():
    pts = po.g1_bases(20, 5) + [None, po.G1_GEN, po.g1_neg(po.G1_GEN)]
    for P in pts:
        b = ser.g1_serialize_compressed(P)
        assert b == po.g1_compress(P) and len(b) == 48
        assert ser.g1_deserialize_compressed(b) == P
        u = ser.g1_serialize_uncompressed(P)
        assert len(u) == 96 and ser.g1_deserialize_uncompressed(u) == P
    # the 56-byte one-point Vec<G1> message of the reference's log (hack/run-hyperplonk/output.txt:25)
    assert len(ser.g1_vec_serialize_compressed([po.G1_GEN])) == 56
    assert ser.g1_serialize_compressed(po.G1_GEN).hex().startswith("97f1d3a73197d794")
