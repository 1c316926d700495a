// arkworks wire / on-disk encodings for the types that cross the reference's network and file boundaries
// (dist-primitive/src/utils/serializing_net.rs:17,50,88,111 `serialize_compressed`; examples/delegator.rs:35-39,64-68
// `serialize_uncompressed`).  On one MI355X node the exchanges move raw limbs (no compression on xGMI); these encoders exist so
// that shares and proofs can be handed to / taken from the Rust prover unchanged.
//
//   Fr               canonical little-endian, 32 bytes (ark-serialize 0.4.2 for Fp)
//   Vec<T>           u64 LE length, then the items; tuples = concatenation
//   G1 compressed    ark-bls12-381 0.4.0 (zcash style): 48-byte BIG-endian x; top three bits of byte 0 =
//                    (compressed = 1, infinity, y is the lexicographically larger root)
//   G1 uncompressed  96 bytes: x || y big-endian, same flag bits with compressed = 0
// ASSUMPTION: restated from the public specification; the only in-tree evidence is message sizes
// (56 B = 8 + 48 for a one-point Vec<G1>, hack/run-hyperplonk/output.txt:25).
#pragma once
#include <cstdio>
#include <string>

#include "device.hpp"
#include "pss.hpp"

namespace zkhost {

struct FqParams {
    static constexpr size_t N = 6;
    // q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    static constexpr uint64_t MOD[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    static constexpr uint64_t INV = 0x89f3fffcfffcfffdull;  // -q^{-1} mod 2^64
};
using Fq = Mont<FqParams>;
using Bytes = std::vector<uint8_t>;

struct G1Affine {  // ark-ec Affine { x, y, infinity }, coordinates in Montgomery form
    Fq x, y;
    bool infinity;
};
// the library returns points as NORMALISED Jacobian (z = R mod q, or z = 0 for infinity): x, y are the affine coordinates
inline G1Affine g1_affine(const G1 &p) {
    G1Affine a;
    std::memcpy(a.x.v, &p[0], 48), std::memcpy(a.y.v, &p[6], 48);
    a.infinity = !(p[12] | p[13] | p[14] | p[15] | p[16] | p[17]);
    return a;
}

// ---- Fr ----
inline void fr_serialize(const Fr &x, Bytes &out) {
    Fr c = x.to_canonical();
    const uint8_t *p = (const uint8_t *)c.v;  // little-endian host, little-endian encoding
    out.insert(out.end(), p, p + 32);
}
inline Fr fr_deserialize(const uint8_t *b) {
    Fr c;
    std::memcpy(c.v, b, 32);
    if (Fr::geq_mod(c.v)) throw std::invalid_argument("non-canonical Fr encoding");
    return Fr::from_canonical(c);
}
inline Bytes fr_vec_serialize(const FrVec &xs) {
    Bytes out(8);
    uint64_t n = xs.size();
    std::memcpy(out.data(), &n, 8);
    for (const Fr &x : xs) fr_serialize(x, out);
    return out;
}
inline FrVec fr_vec_deserialize(const Bytes &b) {
    uint64_t n = 0;
    if (b.size() < 8) throw std::invalid_argument("Vec<Fr>: truncated length prefix");
    std::memcpy(&n, b.data(), 8);
    if (b.size() != 8 + 32 * n) throw std::invalid_argument("Vec<Fr>: the length prefix does not match the payload");
    FrVec out(n);
    for (uint64_t i = 0; i < n; ++i) out[i] = fr_deserialize(&b[8 + 32 * i]);
    return out;
}

// ---- G1 ----
namespace detail {
inline void fq_be(const Fq &x, uint8_t *out48) {  // canonical big-endian
    Fq c = x.to_canonical();
    for (int i = 0; i < 48; ++i) out48[i] = (uint8_t)(c.v[5 - i / 8] >> (56 - 8 * (i % 8)));
}
inline Fq fq_from_be(const uint8_t *b48, uint8_t first) {  // `first`: byte 0 with the flag bits cleared
    Fq c = Fq::zero();
    for (int i = 0; i < 48; ++i) c.v[5 - i / 8] |= (uint64_t)(i ? b48[i] : first) << (56 - 8 * (i % 8));
    if (Fq::geq_mod(c.v)) throw std::invalid_argument("non-canonical Fq encoding");
    return Fq::from_canonical(c);
}
inline bool lexicographically_largest(const Fq &y) {  // y > (q - 1) / 2  <=>  2y > q - 1  <=>  y > -y as integers
    Fq a = y.to_canonical(), b = (-y).to_canonical();
    for (int i = 5; i >= 0; --i)
        if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
    return false;
}
inline Fq curve_rhs(const Fq &x) { return x * x * x + Fq::from_u64(4); }  // y^2 = x^3 + 4
}  // namespace detail

inline void g1_serialize_compressed(const G1Affine &p, Bytes &out) {
    uint8_t b[48] = {0};
    if (p.infinity) {
        b[0] = 0xC0;
    } else {
        detail::fq_be(p.x, b);
        b[0] |= 0x80;
        if (detail::lexicographically_largest(p.y)) b[0] |= 0x20;
    }
    out.insert(out.end(), b, b + 48);
}
inline G1Affine g1_deserialize_compressed(const uint8_t *b) {
    uint8_t flags = b[0] & 0xE0;
    if (!(flags & 0x80)) throw std::invalid_argument("not a compressed encoding");
    if (flags & 0x40) return {Fq::zero(), Fq::zero(), true};
    Fq x = detail::fq_from_be(b, b[0] & 0x1F), rhs = detail::curve_rhs(x);
    // q = 3 mod 4: sqrt = rhs^((q + 1) / 4)
    uint64_t e[6];
    unsigned __int128 c = 1;
    for (int i = 0; i < 6; ++i) {
        c += FqParams::MOD[i];
        e[i] = (uint64_t)c;
        c >>= 64;
    }
    for (int i = 0; i < 6; ++i) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
    Fq y = rhs.pow(e);
    if (y * y != rhs) throw std::invalid_argument("x is not on the curve");
    if (detail::lexicographically_largest(y) != bool(flags & 0x20)) y = -y;
    return {x, y, false};
}
inline void g1_serialize_uncompressed(const G1Affine &p, Bytes &out) {
    uint8_t b[96] = {0};
    if (p.infinity) {
        b[0] = 0x40;
    } else {
        detail::fq_be(p.x, b), detail::fq_be(p.y, b + 48);
    }
    out.insert(out.end(), b, b + 96);
}
inline G1Affine g1_deserialize_uncompressed(const uint8_t *b) {
    if (b[0] & 0x40) return {Fq::zero(), Fq::zero(), true};
    Fq x = detail::fq_from_be(b, b[0] & 0x1F), y = detail::fq_from_be(b + 48, b[48]);
    if (y * y != detail::curve_rhs(x)) throw std::invalid_argument("point not on the curve");
    return {x, y, false};
}
// Vec<G1> as the reference's nets send it (`serialize_compressed` of a Vec: 8 + 48 k bytes)
inline Bytes g1_vec_serialize_compressed(const G1Vec &ps) {
    Bytes out(8);
    uint64_t n = ps.size();
    std::memcpy(out.data(), &n, 8);
    for (const G1 &p : ps) g1_serialize_compressed(g1_affine(p), out);
    return out;
}
// the 96-byte record zk_srs_register takes (x || y Montgomery limbs, zeros = infinity)
inline void g1_affine_record(const G1Affine &p, uint8_t *out96) {
    std::memset(out96, 0, 96);
    if (!p.infinity) std::memcpy(out96, p.x.v, 48), std::memcpy(out96 + 48, p.y.v, 48);
}

// ---- share files of dist-primitive/examples/delegator.rs: `<dir>/delegator` holds the witness Vec<Fr>, `<dir>/worker_<i>` party
// i's Vec<Fr> of packed shares, both written with `serialize_uncompressed` (:35-39, :64-68, :82-95).  For a prime field the
// uncompressed encoding IS the canonical 32-byte little-endian one: a file is `u64 LE length || 32-byte elements`. ----
namespace detail {
inline Bytes read_file(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error(path + ": cannot open");
    Bytes b;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + k);
    std::fclose(f);
    return b;
}
inline void write_file(const std::string &path, const Bytes &b) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error(path + ": cannot create (the delegator example panics when the directory does not exist)");
    if (!b.empty() && std::fwrite(b.data(), 1, b.size(), f) != b.size()) {
        std::fclose(f);
        throw std::runtime_error(path + ": short write");
    }
    std::fclose(f);
}
}  // namespace detail

// Delegator::delegate (:48-62): chunks of l secrets -> pack_from_public -> worker j collects share j
inline std::vector<FrVec> delegator_share(const FrVec &x, const PackedSharingParams &pp) {
    std::vector<FrVec> workers(pp.n);
    for (size_t k = 0; k < x.size(); k += pp.l) {
        FrVec sh = pp.pack_from_public(FrVec(x.begin() + k, x.begin() + std::min(x.size(), k + pp.l)));
        for (size_t j = 0; j < pp.n; ++j) workers[j].push_back(sh[j]);
    }
    return workers;
}
// main (:71-95): writes `delegator` and `worker_0 .. worker_{8l-1}` into an EXISTING directory
inline void delegator_write(const std::string &dir, const FrVec &x, const PackedSharingParams &pp) {
    detail::write_file(dir + "/delegator", fr_vec_serialize(x));
    std::vector<FrVec> w = delegator_share(x, pp);
    for (size_t i = 0; i < w.size(); ++i) detail::write_file(dir + "/worker_" + std::to_string(i), fr_vec_serialize(w[i]));
}
// a share file -> (device buffer of n Fr in the library's Montgomery form, n).  The conversion runs on the GPU: canonical limbs
// a are the Montgomery form of a / R, one Montgomery multiplication by R^2 gives a R.
inline std::pair<DevPtr, size_t> fr_file_to_device(Ctx &be, const std::string &path) {
    Bytes raw = detail::read_file(path);
    uint64_t n = 0;
    if (raw.size() < 8) throw std::invalid_argument(path + ": truncated length prefix");
    std::memcpy(&n, raw.data(), 8);
    if (raw.size() != 8 + 32 * n) throw std::invalid_argument(path + ": the length prefix does not match the file size");
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t limbs[4];
        std::memcpy(limbs, &raw[8 + 32 * i], 32);
        if (Fr::geq_mod(limbs)) throw std::invalid_argument(path + ": non-canonical Fr encoding");
    }
    DevPtr d = be.alloc_fr(n);
    be.upload(d, raw.data() + 8, 32 * n);
    return {be.fr_scale(d, Fr::r2(), n), (size_t)n};
}
// device Fr vector (Montgomery) -> share file; out of Montgomery form on the GPU (multiplication by the integer 1)
inline void fr_device_to_file(Ctx &be, const DevPtr &buf, size_t n, const std::string &path) {
    DevPtr canon = be.fr_scale(buf, Fr::raw_u64(1), n);
    Bytes out(8 + 32 * n);
    uint64_t n64 = n;
    std::memcpy(out.data(), &n64, 8);
    be.download(out.data() + 8, canon, 32 * n);
    detail::write_file(path, out);
}

}  // namespace zkhost
