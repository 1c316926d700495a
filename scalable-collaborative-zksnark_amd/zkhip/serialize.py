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
