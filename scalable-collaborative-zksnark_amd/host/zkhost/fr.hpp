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
#include "mont.hpp"

namespace zkhost {

struct FrParams {
    static constexpr size_t N = 4;
    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static constexpr uint64_t MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    static constexpr uint64_t INV = 0xfffffffeffffffffull;  // -r^{-1} mod 2^64
};
using Fr = Mont<FrParams>;
using FrVec = std::vector<Fr>;

// F::GENERATOR = 7 and TWO_ADIC_ROOT_OF_UNITY = 7^((r-1) / 2^32) (ark-bls12-381 fr.rs; SURVEY.md 8 "PSS exact semantics")
inline Fr fr_generator() { return Fr::from_u64(7); }
inline Fr fr_two_adic_root() {
    static const Fr k = [] {
        // (r - 1) >> 32
        uint64_t e[4];
        uint64_t m[4] = {FrParams::MOD[0] - 1, FrParams::MOD[1], FrParams::MOD[2], FrParams::MOD[3]};
        for (int i = 0; i < 4; ++i) e[i] = (m[i] >> 32) | (i < 3 ? m[i + 1] << 32 : 0);
        return fr_generator().pow(e);
    }();
    return k;
}

}  // namespace zkhost
