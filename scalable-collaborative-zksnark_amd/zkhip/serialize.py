"""
arkworks wire / on-disk encodings for the types that cross the reference's network and file
boundaries (dist-primitive/src/utils/serializing_net.rs:17,50,88,111 `serialize_compressed`;
examples/delegator.rs:35-39,64-68 `serialize_uncompressed`).  On one MI355X node the exchanges move
raw limbs (no compression on xGMI); these encoders exist so shares and proofs can be handed to /
taken from the Rust prover unchanged.

  Fr            canonical little-endian, 32 bytes (ark-serialize 0.4.2 for Fp)
  Vec<T>        u64 LE length, then the items; tuples = concatenation
  G1 compressed ark-bls12-381 0.4.0 (zcash style): 48-byte BIG-endian x; top three bits of byte 0 =
                (compressed = 1, infinity, y is the lexicographically larger root)
  G1 uncompressed  96 bytes: x || y big-endian, same flag bits with compressed = 0
ASSUMPTION: restated from the public specification; the only in-tree
evidence is message sizes (56 B = 8 + 48 for a one-point Vec<G1>, hack/run-hyperplonk/output.txt:25).
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

from .field import Q_MOD, R_MOD

Point = Optional[Tuple[int, int]]


def fr_serialize(x: int) -> bytes:
    return int(x % R_MOD).to_bytes(32, "little")


def fr_deserialize(b: bytes) -> int:
    v = int.from_bytes(b[:32], "little")
    if v >= R_MOD:
        raise ValueError("non-canonical Fr encoding")
    return v


def vec_serialize(items: Sequence[bytes]) -> bytes:
    return struct.pack("<Q", len(items)) + b"".join(items)


def fr_vec_serialize(xs: Sequence[int]) -> bytes:
    return vec_serialize([fr_serialize(x) for x in xs])


def fr_vec_deserialize(b: bytes) -> List[int]:
    (n,) = struct.unpack_from("<Q", b, 0)
    return [fr_deserialize(b[8 + 32 * i : 40 + 32 * i]) for i in range(n)]


def _sqrt_fq(a: int) -> Optional[int]:
    # q = 3 mod 4
    r = pow(a, (Q_MOD + 1) // 4, Q_MOD)
    return r if r * r % Q_MOD == a % Q_MOD else None


def g1_serialize_compressed(P: Point) -> bytes:
    if P is None:
        b = bytearray(48)
        b[0] = 0xC0
        return bytes(b)
    x, y = P
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (Q_MOD - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def g1_deserialize_compressed(b: bytes) -> Point:
    flags = b[0] & 0xE0
    if not flags & 0x80:
        raise ValueError("not a compressed encoding")
    if flags & 0x40:
        return None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    y = _sqrt_fq((x * x * x + 4) % Q_MOD)
    if y is None:
        raise ValueError("x is not on the curve")
    if (y > (Q_MOD - 1) // 2) != bool(flags & 0x20):
        y = Q_MOD - y
    return (x, y)


def g1_serialize_uncompressed(P: Point) -> bytes:
    if P is None:
        b = bytearray(96)
        b[0] = 0x40
        return bytes(b)
    return P[0].to_bytes(48, "big") + P[1].to_bytes(48, "big")


def g1_deserialize_uncompressed(b: bytes) -> Point:
    if b[0] & 0x40:
        return None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    y = int.from_bytes(b[48:96], "big")
    if (y * y - x * x * x - 4) % Q_MOD:
        raise ValueError("point not on the curve")
    return (x, y)


def g1_vec_serialize_compressed(ps: Sequence[Point]) -> bytes:
    return vec_serialize([g1_serialize_compressed(P) for P in ps])


# ---------------------------------------------------------------------------------------
# share files of dist-primitive/examples/delegator.rs: `<dir>/delegator` holds the witness Vec<Fr>,
# `<dir>/worker_<i>` party i's Vec<Fr> of packed shares, both written with `serialize_uncompressed`
# (:35-39, :64-68, :82-95).  For a prime field the uncompressed encoding IS the canonical 32-byte
# little-endian one, so a file is `u64 LE length || 32-byte elements`.  (The example is instantiated on
# ark_bls12_377::Fr; the layout does not depend on the field, the modulus check below does: BLS12-381.)
# ---------------------------------------------------------------------------------------
def delegator_share(x: Sequence[int], pp) -> List[List[int]]:
    """Delegator::delegate (:48-62): chunks of l secrets -> pack_from_public -> worker j collects share j"""
    workers: List[List[int]] = [[] for _ in range(pp.n)]
    for k in range(0, len(x), pp.l):
        for j, s in enumerate(pp.pack_from_public(list(x[k : k + pp.l]))):
            workers[j].append(s)
    return workers


def delegator_write(directory: str, x: Sequence[int], pp) -> None:
    """main (:71-95): the directory must exist; writes `delegator` and `worker_0 .. worker_{8l-1}`"""
    import os

    if not os.path.isdir(directory):
        raise FileNotFoundError(f"{directory} does not exist")  # the example panics
    with open(os.path.join(directory, "delegator"), "wb") as f:
        f.write(fr_vec_serialize(x))
    for i, w in enumerate(delegator_share(x, pp)):
        with open(os.path.join(directory, f"worker_{i}"), "wb") as f:
            f.write(fr_vec_serialize(w))


def fr_file_to_limbs(path: str):
    """a share file -> [n, 4] uint64 CANONICAL limbs (numpy view of the payload, no per-element python work)"""
    import numpy as np

    raw = open(path, "rb").read()
    (n,) = struct.unpack_from("<Q", raw, 0)
    if len(raw) != 8 + 32 * n:
        raise ValueError(f"{path}: length prefix {n} does not match {len(raw)} bytes")
    a = np.frombuffer(raw, dtype="<u8", offset=8).reshape(n, 4).astype(np.uint64)
    r = np.array([(R_MOD >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    lt, eq = np.zeros(n, dtype=bool), np.ones(n, dtype=bool)
    for k in (3, 2, 1, 0):
        lt |= eq & (a[:, k] < r[k])
        eq &= a[:, k] == r[k]
    if not lt.all():
        raise ValueError(f"{path}: non-canonical Fr encoding")
    return a


_R2 = (1 << 512) % R_MOD


def fr_file_to_device(ctx, path: str):
    """
    a share file -> (device buffer of n Fr in the library's Montgomery form, n).  The conversion runs on the
    GPU: canonical limbs a are the Montgomery form of a/R, one Montgomery multiplication by R^2 gives a*R.
    """
    from .field import int_to_limbs

    a = fr_file_to_limbs(path)
    buf = ctx.to_device(a)
    return ctx.fr_scale(buf, int_to_limbs(_R2, 4), len(a), out=buf), len(a)


def fr_device_to_file(ctx, buf, n: int, path: str) -> None:
    """device Fr vector (Montgomery) -> share file; out of Montgomery form on the GPU (multiplication by 1)"""
    from .field import int_to_limbs

    tmp = ctx.fr_scale(buf, int_to_limbs(1, 4), n)
    a = tmp.download((n, 4))
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", n) + a.astype("<u8").tobytes())
