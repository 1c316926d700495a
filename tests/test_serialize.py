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


def test_delegator_share_files_roundtrip(tmp_path):
    """examples/delegator.rs: `delegator` + `worker_i` files; the shares unpack back to the witness"""
    import numpy as np
    import pytest

    from zkhip.pss import PackedSharingParams

    rng = po.SplitMix64(33)
    l, n = 2, 10
    x = rng.fr_vec(n)
    pp, opp = PackedSharingParams(l), po.PackedSharingParams(l)
    ser.delegator_write(str(tmp_path), x, pp)
    assert sorted(p.name for p in tmp_path.iterdir()) == sorted(["delegator"] + [f"worker_{i}" for i in range(8 * l)])
    assert ser.fr_vec_deserialize((tmp_path / "delegator").read_bytes()) == x
    shares = [ser.fr_vec_deserialize((tmp_path / f"worker_{i}").read_bytes()) for i in range(8 * l)]
    assert all(len(s) == n // l for s in shares)
    for k in range(n // l):  # the ORACLE's unpack recovers chunk k from the k-th share of every worker
        assert opp.unpack([shares[i][k] for i in range(8 * l)]) == x[k * l : (k + 1) * l]
    limbs = ser.fr_file_to_limbs(str(tmp_path / "worker_3"))
    assert limbs.shape == (n // l, 4) and [int.from_bytes(r.tobytes(), "little") for r in limbs] == shares[3]
    bad = tmp_path / "bad"
    bad.write_bytes((1).to_bytes(8, "little") + (po.R_MOD).to_bytes(32, "little"))
    with pytest.raises(ValueError):
        ser.fr_file_to_limbs(str(bad))
    with pytest.raises(FileNotFoundError):
        ser.delegator_write(str(tmp_path / "missing"), x, pp)
