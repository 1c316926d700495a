// Host-side BLS12-381 scalar field for the C++ mirror of `dist-primitive` (zkhost).
//
// The reference keeps every field element as ark-ff `Fp<MontBackend<FrConfig, 4>, 4>`: four little-endian u64 limbs in
// Montgomery form, R = 2^256 (ark-bls12-381 0.4.0, fields/fr.rs).  `Fr` below has that layout, so a
// `std::vector<Fr>` is byte-for-byte what the C ABI (include/zkhip.h) takes and returns.
//
// Only the SMALL host-side pieces of the path use this arithmetic -- the leader rounds of the collaborative sumchecks
// (dsumcheck.rs:226-283, :440-507), pss2ss (unpack.rs:72-97), the top of the product tree (dacc_product.rs:339-361),
// the PSS matrices (pss.rs:38-172): a few hundred elements per call.  Tables never pass through it.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

namespace zkhost {

struct Fr {
    uint64_t v[4];

    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static constexpr uint64_t MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    static constexpr uint64_t INV = 0xfffffffeffffffffull;  // -r^{-1} mod 2^64

    static Fr zero() { return Fr{{0, 0, 0, 0}}; }
    bool is_zero() const { return !(v[0] | v[1] | v[2] | v[3]); }
    bool operator==(const Fr &o) const { return !std::memcmp(v, o.v, 32); }
    bool operator!=(const Fr &o) const { return !(*this == o); }

    static bool geq_mod(const uint64_t a[4]) {
        for (int i = 3; i >= 0; --i) {
            if (a[i] != MOD[i]) return a[i] > MOD[i];
        }
        return true;
    }
    static void sub_mod(uint64_t a[4]) {
        unsigned __int128 b = 0;
        for (int i = 0; i < 4; ++i) {
            unsigned __int128 d = (unsigned __int128)a[i] - MOD[i] - (uint64_t)b;
            a[i] = (uint64_t)d;
            b = (d >> 64) & 1;
        }
    }

    friend Fr operator+(const Fr &a, const Fr &b) {
        Fr r;
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (unsigned __int128)a.v[i] + b.v[i];
            r.v[i] = (uint64_t)c;
            c >>= 64;
        }
        if (c || geq_mod(r.v)) sub_mod(r.v);  // (r < 2^255: the carry never occurs for reduced inputs)
        return r;
    }
    friend Fr operator-(const Fr &a, const Fr &b) {
        Fr r;
        unsigned __int128 bw = 0;
        for (int i = 0; i < 4; ++i) {
            unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - (uint64_t)bw;
            r.v[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
        if (bw) {
            unsigned __int128 c = 0;
            for (int i = 0; i < 4; ++i) {
                c += (unsigned __int128)r.v[i] + MOD[i];
                r.v[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    Fr operator-() const { return zero() - *this; }

    // Montgomery product a b R^-1 mod r (CIOS, 4 x 64-bit limbs)
    friend Fr operator*(const Fr &a, const Fr &b) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (unsigned __int128)a.v[j] * b.v[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * INV;
            c = (unsigned __int128)m * MOD[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; ++j) {
                c += (unsigned __int128)m * MOD[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fr r{{t[0], t[1], t[2], t[3]}};
        if (t[4] || geq_mod(r.v)) sub_mod(r.v);
        return r;
    }
    Fr &operator+=(const Fr &o) { return *this = *this + o; }
    Fr &operator-=(const Fr &o) { return *this = *this - o; }
    Fr &operator*=(const Fr &o) { return *this = *this * o; }

    // R^2 mod r by 512 modular doublings of 1 (no constant to mistype)
    static const Fr &r2() {
        static const Fr k = [] {
            Fr x{{1, 0, 0, 0}};
            for (int i = 0; i < 512; ++i) x = x + x;
            return x;
        }();
        return k;
    }
    static Fr one() {
        static const Fr k = from_canonical(Fr{{1, 0, 0, 0}});
        return k;
    }
    // canonical integer limbs (< r) -> Montgomery form and back (`from_bigint` / `into_bigint`)
    static Fr from_canonical(const Fr &c) { return c * r2(); }
    Fr to_canonical() const { return *this * Fr{{1, 0, 0, 0}}; }
    static Fr from_u64(uint64_t x) { return from_canonical(Fr{{x, 0, 0, 0}}); }

    Fr pow(const uint64_t e[4]) const {
        Fr acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc * acc;
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
        }
        return acc;
    }
    Fr pow_u64(uint64_t e) const {
        uint64_t ee[4] = {e, 0, 0, 0};
        return pow(ee);
    }
    // x^(r-2); the reference's `inverse().unwrap()` panics on zero: callers check
    Fr inverse() const {
        uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]};
        return pow(e);
    }
};

using FrVec = std::vector<Fr>;

// F::GENERATOR = 7 and TWO_ADIC_ROOT_OF_UNITY = 7^((r-1) / 2^32) (ark-bls12-381 fr.rs; SURVEY.md 8 "PSS exact semantics")
inline Fr fr_generator() { return Fr::from_u64(7); }
inline Fr fr_two_adic_root() {
    static const Fr k = [] {
        // (r - 1) >> 32
        uint64_t e[4];
        uint64_t m[4] = {Fr::MOD[0] - 1, Fr::MOD[1], Fr::MOD[2], Fr::MOD[3]};
        for (int i = 0; i < 4; ++i) e[i] = (m[i] >> 32) | (i < 3 ? m[i + 1] << 32 : 0);
        return fr_generator().pow(e);
    }();
    return k;
}

}  // namespace zkhost
