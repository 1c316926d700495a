// Prime fields in Montgomery form, N little-endian u64 limbs, R = 2^(64 N): the memory layout of ark-ff's
// `Fp<MontBackend<_, N>, N>`, so a std::vector of these is byte-for-byte what the C ABI takes and returns.
// Host-side only, for the handful of elements the reference's LEADER handles and for the wire formats; tables never pass here.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

namespace zkhost {

// P provides: static constexpr size_t N; static constexpr uint64_t MOD[N]; static constexpr uint64_t INV (= -MOD^-1 mod 2^64)
template <class P>
struct Mont {
    static constexpr size_t N = P::N;
    uint64_t v[N];

    static constexpr const uint64_t *MOD = P::MOD;
    static constexpr uint64_t INV = P::INV;

    static Mont zero() {
        Mont z;
        std::memset(z.v, 0, sizeof z.v);
        return z;
    }
    static Mont raw_u64(uint64_t x) {  // the integer x as limbs (NOT in Montgomery form)
        Mont z = zero();
        z.v[0] = x;
        return z;
    }
    bool is_zero() const {
        uint64_t o = 0;
        for (size_t i = 0; i < N; ++i) o |= v[i];
        return !o;
    }
    bool operator==(const Mont &o) const { return !std::memcmp(v, o.v, sizeof v); }
    bool operator!=(const Mont &o) const { return !(*this == o); }

    static bool geq_mod(const uint64_t *a) {
        for (size_t i = N; i-- > 0;) {
            if (a[i] != P::MOD[i]) return a[i] > P::MOD[i];
        }
        return true;
    }
    static void sub_mod(uint64_t *a) {
        unsigned __int128 b = 0;
        for (size_t i = 0; i < N; ++i) {
            unsigned __int128 d = (unsigned __int128)a[i] - P::MOD[i] - (uint64_t)b;
            a[i] = (uint64_t)d;
            b = (d >> 64) & 1;
        }
    }

    friend Mont operator+(const Mont &a, const Mont &b) {
        Mont r;
        unsigned __int128 c = 0;
        for (size_t i = 0; i < N; ++i) {
            c += (unsigned __int128)a.v[i] + b.v[i];
            r.v[i] = (uint64_t)c;
            c >>= 64;
        }
        if (c || geq_mod(r.v)) sub_mod(r.v);
        return r;
    }
    friend Mont operator-(const Mont &a, const Mont &b) {
        Mont r;
        unsigned __int128 bw = 0;
        for (size_t i = 0; i < N; ++i) {
            unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - (uint64_t)bw;
            r.v[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
        if (bw) {
            unsigned __int128 c = 0;
            for (size_t i = 0; i < N; ++i) {
                c += (unsigned __int128)r.v[i] + P::MOD[i];
                r.v[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    Mont operator-() const { return zero() - *this; }

    // Montgomery product a b R^-1 mod p (CIOS)
    friend Mont operator*(const Mont &a, const Mont &b) {
        uint64_t t[N + 2];
        std::memset(t, 0, sizeof t);
        for (size_t i = 0; i < N; ++i) {
            unsigned __int128 c = 0;
            for (size_t j = 0; j < N; ++j) {
                c += (unsigned __int128)a.v[j] * b.v[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N] = (uint64_t)c;
            t[N + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * P::INV;
            c = (unsigned __int128)m * P::MOD[0] + t[0];
            c >>= 64;
            for (size_t j = 1; j < N; ++j) {
                c += (unsigned __int128)m * P::MOD[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N - 1] = (uint64_t)c;
            t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        Mont r;
        std::memcpy(r.v, t, sizeof r.v);
        if (t[N] || geq_mod(r.v)) sub_mod(r.v);
        return r;
    }
    Mont &operator+=(const Mont &o) { return *this = *this + o; }
    Mont &operator-=(const Mont &o) { return *this = *this - o; }
    Mont &operator*=(const Mont &o) { return *this = *this * o; }

    // R^2 mod p by 2 * 64 N modular doublings of 1 (no constant to mistype)
    static const Mont &r2() {
        static const Mont k = [] {
            Mont x = raw_u64(1);
            for (size_t i = 0; i < 128 * N; ++i) x = x + x;
            return x;
        }();
        return k;
    }
    static Mont one() {
        static const Mont k = from_canonical(raw_u64(1));
        return k;
    }
    // canonical integer limbs (< p) -> Montgomery form and back (`from_bigint` / `into_bigint`)
    static Mont from_canonical(const Mont &c) { return c * r2(); }
    Mont to_canonical() const { return *this * raw_u64(1); }
    static Mont from_u64(uint64_t x) { return from_canonical(raw_u64(x)); }

    Mont pow(const uint64_t *e) const {  // e: N limbs
        Mont acc = one();
        for (size_t i = 64 * N; i-- > 0;) {
            acc = acc * acc;
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
        }
        return acc;
    }
    Mont pow_u64(uint64_t e) const {
        uint64_t ee[N] = {e};
        return pow(ee);
    }
    // x^(p-2); the reference's `inverse().unwrap()` panics on zero: callers check
    Mont inverse() const {
        uint64_t e[N];
        std::memcpy(e, P::MOD, sizeof e);
        e[0] -= 2;  // (both moduli end in ..01 / ..ab: no borrow)
        return pow(e);
    }
};

}  // namespace zkhost
