// PackedSharingParams -- C++ host mirror of secret-sharing/src/pss.rs:17-172.
//
// Every PSS map is a fixed PUBLIC linear map over Fr on vectors of N_p = 8l entries.  The maps are evaluated here as the
// reference defines them (ifft on one domain, resize, fft on the other: pss.rs:93-171) with dense DFTs -- the domains have
// 2l, 4l and 8l points -- and kept as small matrices, which is the form the C ABI applies to whole tables
// (zk_fr_apply_matrix) and to points (zk_g1_lincomb_batch).  The ark-poly behaviour the reference relies on is isolated in
// Domain::fft / Domain::ifft: `fft_in_place` / `ifft_in_place` first RESIZE the vector to the domain size (SURVEY.md
// Appendix C: restated from the public definition, not checkable against arkworks in this environment).
#pragma once
#include <cassert>
#include <stdexcept>
#include <utility>
#include <vector>

#include "fr.hpp"

namespace zkhost {

// Radix2EvaluationDomain::new(size) and its `get_coset(offset)` (pss.rs:43-51)
struct Domain {
    size_t size;
    Fr offset, omega;
    FrVec w, winv;  // powers of omega / omega^-1

    Domain() : size(0) {}
    Domain(size_t size_, const Fr &offset_) : size(size_), offset(offset_) {
        assert(size && !(size & (size - 1)));
        omega = fr_two_adic_root().pow_u64((1ull << 32) / size);
        Fr wi = omega.inverse();
        w.assign(size, Fr::one());
        winv.assign(size, Fr::one());
        for (size_t i = 1; i < size; ++i) {
            w[i] = w[i - 1] * omega;
            winv[i] = winv[i - 1] * wi;
        }
    }
    FrVec resized(const FrVec &v) const {
        FrVec r(v.begin(), v.begin() + std::min(v.size(), size));
        r.resize(size, Fr::zero());
        return r;
    }
    // in-place radix-2 transform of `size` values with the given powers of the root (w or winv): out[j] = sum_k v[k] root^(jk).
    // Exact field arithmetic: the same canonical values as the O(size^2) sums of the definition, 60 x fewer multiplications at
    // 256 parties (the constructor below evaluates 3 size maps on every unit vector)
    void transform(FrVec &v, const FrVec &roots) const {
        for (size_t i = 1, j = 0; i < size; ++i) {  // bit reversal
            size_t bit = size >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(v[i], v[j]);
        }
        for (size_t len = 2; len <= size; len <<= 1) {
            size_t half = len >> 1, step = size / len;
            for (size_t i = 0; i < size; i += len)
                for (size_t k = 0; k < half; ++k) {
                    Fr u = v[i + k], t = v[i + k + half] * roots[k * step];
                    v[i + k] = u + t;
                    v[i + k + half] = u - t;
                }
        }
    }
    // evaluate sum_k c_k x^k at x_j = offset omega^j
    FrVec fft(const FrVec &coeffs) const {
        FrVec c = resized(coeffs);
        Fr op = Fr::one();
        for (size_t k = 0; k < size; ++k) {
            c[k] *= op;
            op *= offset;
        }
        transform(c, w);
        return c;
    }
    FrVec ifft(const FrVec &evals) const {
        FrVec e = resized(evals);
        transform(e, winv);
        Fr op = Fr::from_u64(size).inverse(), oinv = offset.inverse();
        for (size_t k = 0; k < size; ++k) {
            e[k] *= op;
            op *= oinv;
        }
        return e;
    }
};

// twiddles and scales of one map for zk_fr_ntt_map (Montgomery limbs): see include/zkhip.h
struct NttTables {
    size_t A, B, n_in, take, step;
    FrVec winv, w, scale;
};

struct PackedSharingParams {
    size_t l, n, t;                // packing factor, parties = 8l, threshold = l - 1   (pss.rs:38-64)
    Domain share, secret, secret2;
    // share_i = sum_j pack[i][j] secret_j (secrets zero-padded to 2l);  secret_j = sum_i unpack[j][i] share_i
    std::vector<FrVec> pack_matrix, unpack_matrix, unpack2_matrix;
    FrVec single_one_;  // pack_single(1)

    explicit PackedSharingParams(size_t l_) : l(l_), n(8 * l_), t(l_ - 1) {
        if (!l || (l & (l - 1))) throw std::invalid_argument("PackedSharingParams: l must be a power of two");
        share = Domain(n, Fr::one());
        secret = Domain(2 * l, fr_generator());
        secret2 = Domain(4 * l, fr_generator());
        auto unit = [](size_t m, size_t j) {
            FrVec u(m, Fr::zero());
            u[j] = Fr::one();
            return u;
        };
        pack_matrix.assign(n, FrVec(2 * l));
        for (size_t j = 0; j < 2 * l; ++j) {
            FrVec col = pack_from_public(unit(2 * l, j));
            for (size_t i = 0; i < n; ++i) pack_matrix[i][j] = col[i];
        }
        unpack_matrix.assign(l, FrVec(n));
        unpack2_matrix.assign(l, FrVec(n));
        for (size_t i = 0; i < n; ++i) {
            FrVec c1 = unpack(unit(n, i)), c2 = unpack2(unit(n, i));
            for (size_t j = 0; j < l; ++j) {
                unpack_matrix[j][i] = c1[j];
                unpack2_matrix[j][i] = c2[j];
            }
        }
        single_one_ = pack_single(Fr::one());
    }

    // pss.rs:69-73,93-99
    FrVec pack_from_public(const FrVec &secrets) const { return share.fft(secret.ifft(secrets)); }
    // pss.rs:103-113 -- packs and then packs the n-vector AGAIN (a quirk of the reference, kept literally)
    FrVec pack_single(const Fr &s) const { return pack_from_public(share.fft(secret.ifft(FrVec{s}))); }
    // pack_single is linear in its one argument: pack_single(s)[p] = s * pack_single(1)[p].  pss2ss (unpack.rs:72-97) needs entry p of
    // pack_single of every unpacked secret: one multiplication each instead of four transforms
    // (computed in the constructor: one PackedSharingParams is shared by all party threads of a process)
    const FrVec &pack_single_of_one() const { return single_one_; }
    // pss.rs:117-120,132-149
    FrVec unpack(const FrVec &shares) const {
        FrVec e = secret.fft(share.ifft(shares));
        e.resize(l);
        return e;
    }
    // pss.rs:124-128,153-171 (slots 0, 2, .., 2l-2)
    FrVec unpack2(const FrVec &shares) const {
        assert(shares.size() == n);
        FrVec e = secret2.fft(share.ifft(shares)), out(l);
        for (size_t j = 0; j < l; ++j) out[j] = e[2 * j];
        return out;
    }

    enum class Map { Pack, Unpack, Unpack2 };
    NttTables ntt_tables(Map kind) const {
        const Domain &src = kind == Map::Pack ? secret : share;
        const Domain &dst = kind == Map::Pack ? share : (kind == Map::Unpack ? secret : secret2);
        NttTables t;
        t.A = src.size, t.B = dst.size;
        t.n_in = kind == Map::Pack ? l : n;
        t.take = kind == Map::Pack ? n : l;
        t.step = kind == Map::Unpack2 ? 2 : 1;
        Fr ratio = dst.offset * src.offset.inverse(), ainv = Fr::from_u64(t.A).inverse();
        for (size_t i = 0; i < std::max<size_t>(t.A / 2, 1); ++i) t.winv.push_back(src.winv[i]);
        for (size_t i = 0; i < std::max<size_t>(t.B / 2, 1); ++i) t.w.push_back(dst.w[i]);
        Fr s = ainv;
        for (size_t i = 0; i < std::min(t.A, t.B); ++i) {
            t.scale.push_back(s);
            s *= ratio;
        }
        return t;
    }

    // --- coefficient rows used by the distributed primitives ---
    Fr lambda(size_t party) const {  // sum_j unpack2[j][party]: this party's weight in `unpack2 -> sum` (dmsm.rs:30-36)
        Fr s = Fr::zero();
        for (size_t j = 0; j < l; ++j) s += unpack2_matrix[j][party];
        return s;
    }
    Fr c(size_t party) const {  // pack_from_public([S; l])[party] = c_p S (dmsm.rs:37-39)
        Fr s = Fr::zero();
        for (size_t j = 0; j < l; ++j) s += pack_matrix[party][j];
        return s;
    }
    // d_msm leader closure for output party p: out_p = sum_i (c_p lambda_i) C_i
    FrVec dmsm_coeffs(size_t party) const {
        FrVec out(n);
        Fr cp = c(party);
        for (size_t i = 0; i < n; ++i) out[i] = cp * lambda(i);
        return out;
    }
    // this party's row of the composite pack o unpack2 (degree_reduce.rs:17-23)
    FrVec degree_reduce_row(size_t party) const {
        FrVec row(n, Fr::zero());
        for (size_t i = 0; i < n; ++i)
            for (size_t j = 0; j < l; ++j) row[i] += pack_matrix[party][j] * unpack2_matrix[j][i];
        return row;
    }
};

}  // namespace zkhost
