//---------------------------------------------------------------------------//
// zkhip host-side value types for the header-only shim.
//
// In a crypto3 tree the curve / field value types come from crypto3-algebra
// (`CurveType::scalar_field_type::value_type`, `CurveType::template g1_type<>::value_type`, ...).  That library
// is not part of crypto3-zk, so the shim is written against the small `curve_adapter` concept below and ships a
// self-contained implementation of it (`native_curve<ZKHIP_BLS12_381>`, `native_curve<ZKHIP_BN254>`), built on
// the same field / group-law headers the HIP kernels are compiled from (csrc/fu.hpp, csrc/curve.hpp).  A
// crypto3 maintainer specialises `curve_adapter<nil::crypto3::algebra::curves::bls12<381>>` with the four
// conversion functions instead (INTEGRATION.md shows it).
//
// Everything that crosses into the C ABI is CANONICAL little-endian 64-bit limbs.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_ALGEBRA_HPP
#define ZKHIP_SHIM_ALGEBRA_HPP

#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#define ZK_NOINLINE_MUL 1
#include "../../../../../csrc/curve.hpp"
#include "../../../../../../include/zkhip.h"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

namespace detail {
    template <int Curve>
    struct native_fields;
    template <>
    struct native_fields<ZKHIP_BLS12_381> {
        typedef zkhip::bls_fru fr;
        typedef zkhip::bls_fqu fq;
        typedef zkhip::bls_fqu2 fq2;
    };
    template <>
    struct native_fields<ZKHIP_BN254> {
        typedef zkhip::bn_fru fr;
        typedef zkhip::bn_fqu fq;
        typedef zkhip::bn_fqu2 fq2;
    };
}    // namespace detail

/// Scalar-field element: canonical integer in 4 x u64 limbs; the operators the prover's host glue needs
/// (`r * s` at prover.hpp:153, comparisons with zero / one at knowledge_commitment_multiexp.hpp:88-90).
template <int Curve>
struct fr_value {
    typedef typename detail::native_fields<Curve>::fr F;
    std::array<std::uint64_t, 4> limbs {};

    fr_value() = default;
    fr_value(std::uint64_t v) { limbs[0] = v; }
    static fr_value zero() { return fr_value(); }
    static fr_value one() { return fr_value(1); }
    bool is_zero() const { return (limbs[0] | limbs[1] | limbs[2] | limbs[3]) == 0; }
    bool operator==(const fr_value &o) const { return limbs == o.limbs; }
    bool operator!=(const fr_value &o) const { return !(*this == o); }

    // u64 <-> u32 limb views go through memcpy (no type punning through pointers)
    F mont() const {
        alignas(16) std::uint32_t w[8];
        std::memcpy(w, limbs.data(), 32);
        return zkhip::FieldOps<F>::from_canonical(w);
    }
    static fr_value from_mont(const F &m) {
        alignas(16) std::uint32_t w[8];
        zkhip::FieldOps<F>::to_canonical(w, m);
        fr_value r;
        std::memcpy(r.limbs.data(), w, 32);
        return r;
    }
    /// a * b on canonical limbs: (a R) * b / R with ONE conversion -- the Montgomery product of the converted left operand
    /// and the plain right operand is the canonical product (the same mixed-form trick the NTT kernels use).
    fr_value operator*(const fr_value &o) const {
        alignas(16) std::uint32_t w[8];
        std::memcpy(w, o.limbs.data(), 32);
        const F p = zkhip::fu_cond_sub_p(zkhip::fu_mul(mont(), zkhip::fu_unpack<typename F::params>(w)));
        zkhip::fu_pack<typename F::params>(w, p);
        fr_value r;
        std::memcpy(r.limbs.data(), w, 32);
        return r;
    }
    /// canonical + and -: plain 256-bit integer arithmetic with one conditional correction by r
    fr_value operator+(const fr_value &o) const {
        fr_value r;
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (unsigned __int128)limbs[i] + o.limbs[i];
            r.limbs[i] = (std::uint64_t)c;
            c >>= 64;
        }
        std::uint64_t d[4], mod[4];
        modulus(mod);
        if (!sub_limbs(d, r.limbs.data(), mod)) std::memcpy(r.limbs.data(), d, 32);    // r >= modulus (no carry out: 2 r < 2^256)
        return r;
    }
    fr_value operator-(const fr_value &o) const {
        fr_value r;
        if (sub_limbs(r.limbs.data(), limbs.data(), o.limbs.data())) {    // borrowed: add the modulus back
            std::uint64_t mod[4];
            modulus(mod);
            unsigned __int128 c = 0;
            for (int i = 0; i < 4; ++i) {
                c += (unsigned __int128)r.limbs[i] + mod[i];
                r.limbs[i] = (std::uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    static void modulus(std::uint64_t *out) {
        typedef typename F::params::sat P;
        for (int i = 0; i < 4; ++i) out[i] = (std::uint64_t)P::mod(2 * i) | ((std::uint64_t)P::mod(2 * i + 1) << 32);
    }
    /// d = a - b; returns the borrow
    static bool sub_limbs(std::uint64_t *d, const std::uint64_t *a, const std::uint64_t *b) {
        std::uint64_t borrow = 0;
        for (int i = 0; i < 4; ++i) {
            const std::uint64_t t = a[i] - b[i], t2 = t - borrow;
            borrow = (a[i] < b[i]) | (t < borrow);
            d[i] = t2;
        }
        return borrow != 0;
    }
    fr_value inversed() const { return from_mont(zkhip::FieldOps<F>::inv(mont())); }
};

/// Group element in the 3-coordinate shape of the reference's `G::value_type` (held as XYZZ internally).
/// `Coord` is the coordinate field (Fq for G1, Fq2 for G2), `Words64` the canonical u64 limbs per coordinate.
template <int Curve, int Group>
struct group_value {
    typedef typename std::conditional<Group == ZKHIP_G1, typename detail::native_fields<Curve>::fq,
                                      typename detail::native_fields<Curve>::fq2>::type F;
    typedef zkhip::FieldOps<F> O;
    static constexpr std::size_t coord_limbs = O::CANON_WORDS / 2;    // u64 limbs per coordinate
    zkhip::XYZZ<F> p = zkhip::XYZZ<F>::infinity();

    static group_value zero() { return group_value(); }
    bool is_zero() const { return p.is_inf(); }

    /// from canonical affine limbs (x | y) and an infinity flag
    static group_value from_affine(const std::uint64_t *xy, bool infinity = false) {
        group_value r;
        if (!infinity) {
            alignas(16) std::uint32_t w[2 * O::CANON_WORDS];
            std::memcpy(w, xy, sizeof(w));
            zkhip::Affine<F> a = {O::from_canonical(w), O::from_canonical(w + O::CANON_WORDS)};
            r.p = zkhip::XYZZ<F>::from_affine(a);
        }
        return r;
    }
    /// from the C ABI's Jacobian result (X | Y | Z canonical)
    static group_value from_jacobian(const std::uint64_t *xyz) {
        alignas(16) std::uint32_t w[3 * O::CANON_WORDS];
        std::memcpy(w, xyz, sizeof(w));
        zkhip::Jacobian<F> j = {O::from_canonical(w), O::from_canonical(w + O::CANON_WORDS), O::from_canonical(w + 2 * O::CANON_WORDS)};
        group_value r;
        r.p = zkhip::xyzz_from_jacobian(j);
        return r;
    }
    /// canonical affine limbs (x | y); returns false for the point at infinity (limbs zeroed)
    bool to_affine(std::uint64_t *xy) const {
        if (p.is_inf()) {
            std::memset(xy, 0, 2 * coord_limbs * 8);
            return false;
        }
        alignas(16) std::uint32_t w[2 * O::CANON_WORDS];
        /* a point that is already normalised (ZZ = ZZZ = 1: every element of a key after batch_to_special, generator.hpp:190-192)
           needs no inversion -- what makes handing a 2^20-constraint key to the device seconds instead of minutes */
        O::to_canonical(w, p.ZZ);
        O::to_canonical(w + O::CANON_WORDS, p.ZZZ);
        bool unit = w[0] == 1 && w[O::CANON_WORDS] == 1;
        for (int i = 1; i < O::CANON_WORDS && unit; ++i) unit = w[i] == 0 && w[O::CANON_WORDS + i] == 0;
        if (unit) {
            O::to_canonical(w, p.X);
            O::to_canonical(w + O::CANON_WORDS, p.Y);
        } else {
            zkhip::Affine<F> a = zkhip::xyzz_to_affine(p);
            O::to_canonical(w, a.x);
            O::to_canonical(w + O::CANON_WORDS, a.y);
        }
        std::memcpy(xy, w, sizeof(w));
        return true;
    }
    group_value operator+(const group_value &o) const {
        group_value r;
        r.p = zkhip::xyzz_add(p, o.p);
        return r;
    }
    group_value operator-() const {
        group_value r = *this;
        if (!r.p.is_inf()) r.p.Y = O::template sub<O::K2>(F::zero(), r.p.Y);
        return r;
    }
    group_value operator-(const group_value &o) const { return *this + (-o); }
    /// scalar multiplication (the handful of products at prover.hpp:142-155, two of them on a proof's critical path after the
    /// device results arrive): width-5 signed windows -- 255 doublings + ~43 additions against P, 3P, ..., 15P
    group_value operator*(const fr_value<Curve> &k) const {
        group_value r;
        if (p.is_inf() || k.is_zero()) return r;
        std::uint64_t e[5] = {k.limbs[0], k.limbs[1], k.limbs[2], k.limbs[3], 0};
        signed char naf[260];
        int len = 0;
        while (e[0] | e[1] | e[2] | e[3] | e[4]) {
            int d = 0;
            if (e[0] & 1) {
                d = (int)(e[0] & 31);
                if (d >= 16) d -= 32;
                if (d > 0) {    // e -= d: d is the low bits of e[0], no borrow
                    e[0] -= (std::uint64_t)d;
                } else {    // e += |d|
                    std::uint64_t c = (std::uint64_t)(-d);
                    for (int i = 0; i < 5 && c; ++i) {
                        e[i] += c;
                        c = e[i] < c ? 1 : 0;
                    }
                }
            }
            naf[len++] = (signed char)d;
            for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (e[i + 1] << 63);
            e[4] >>= 1;
        }
        zkhip::XYZZ<F> odd[8];
        odd[0] = p;
        const zkhip::XYZZ<F> twice = zkhip::xyzz_dbl(p);
        for (int i = 1; i < 8; ++i) odd[i] = zkhip::xyzz_add(odd[i - 1], twice);
        for (int i = len - 1; i >= 0; --i) {
            r.p = zkhip::xyzz_dbl(r.p);
            const int d = naf[i];
            if (d > 0) r.p = zkhip::xyzz_add(r.p, odd[(d - 1) >> 1]);
            else if (d < 0) {
                zkhip::XYZZ<F> m = odd[(-d - 1) >> 1];
                if (!m.is_inf()) m.Y = O::template sub<O::K2>(F::zero(), m.Y);
                r.p = zkhip::xyzz_add(r.p, m);
            }
        }
        return r;
    }
    friend group_value operator*(const fr_value<Curve> &k, const group_value &g) { return g * k; }
    bool operator==(const group_value &o) const {
        std::vector<std::uint64_t> a(2 * coord_limbs), b(2 * coord_limbs);
        bool ia = to_affine(a.data()), ib = o.to_affine(b.data());
        return ia == ib && a == b;
    }
};

/// The concept the shim is written against.  `native_curve<C>` is the self-contained model.
template <int Curve>
struct native_curve {
    static constexpr int id = Curve;
    typedef fr_value<Curve> scalar_value_type;
    typedef group_value<Curve, ZKHIP_G1> g1_value_type;
    typedef group_value<Curve, ZKHIP_G2> g2_value_type;
};

/// Adapter: how the shim reads canonical limbs out of / builds values of a curve's types.
template <typename CurveType>
struct curve_adapter;

template <int Curve>
struct curve_adapter<native_curve<Curve>> {
    typedef native_curve<Curve> curve_type;
    static constexpr int id = Curve;
    typedef typename curve_type::scalar_value_type scalar_value_type;
    typedef typename curve_type::g1_value_type g1_value_type;
    typedef typename curve_type::g2_value_type g2_value_type;
    static constexpr std::size_t g1_coord_limbs = g1_value_type::coord_limbs;
    static constexpr std::size_t g2_coord_limbs = g2_value_type::coord_limbs;

    /// fr_value IS four canonical little-endian u64 limbs: bulk uploads send the caller's vectors as they lie (backend.hpp, upload_scalars)
    static constexpr bool scalars_are_canonical_limbs = true;
    static void scalar_to_limbs(const scalar_value_type &s, std::uint64_t *out) { std::memcpy(out, s.limbs.data(), 32); }
    static scalar_value_type scalar_from_limbs(const std::uint64_t *in) {
        scalar_value_type s;
        std::memcpy(s.limbs.data(), in, 32);
        return s;
    }
    /// the scalar-field modulus r, canonical little-endian limbs (what rejection sampling of the blinders compares against)
    static void scalar_modulus(std::uint64_t *out) { scalar_value_type::modulus(out); }
    /// fields::arithmetic_params<scalar_field_type>: multiplicative_generator, two-adicity s and the primitive 2^n-th root of unity
    /// math::unity_root<F>(2^n) = generator^((r - 1) / 2^n) -- what the reference's domain classes read from crypto3-algebra / -math.
    /// With them the shim offers the reference's own ARITIES (no domain constants, no context in the call): standard_domain_params,
    /// the (pk) key constructor, witness_map(cs, x, w), the scheme classes' default roots.
    static constexpr bool has_field_constants = true;
    static constexpr unsigned two_adicity = Curve == ZKHIP_BLS12_381 ? 32 : 28;
    static scalar_value_type multiplicative_generator() { return scalar_value_type(Curve == ZKHIP_BLS12_381 ? 7 : 5); }
    static scalar_value_type root_of_unity(std::size_t log_n) {
        if (log_n > two_adicity) throw std::invalid_argument("root_of_unity: the scalar field has no 2^n-th root of unity for this n");
        const std::uint64_t top[2][4] = {{0x3829971f439f0d2bULL, 0xb63683508c2280b9ULL, 0xd09b681922c813b4ULL, 0x16a2a19edfe81f20ULL},
                                         {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL}};
        scalar_value_type w = scalar_from_limbs(top[Curve == ZKHIP_BLS12_381 ? 0 : 1]);    // generator^((r - 1) / 2^s)
        for (std::size_t k = log_n; k < two_adicity; ++k) w = w * w;
        return w;
    }
    template <typename G>
    static bool point_to_affine_limbs(const G &p, std::uint64_t *out) { return p.to_affine(out); }
    static g1_value_type g1_from_jacobian(const std::uint64_t *xyz) { return g1_value_type::from_jacobian(xyz); }
    static g2_value_type g2_from_jacobian(const std::uint64_t *xyz) { return g2_value_type::from_jacobian(xyz); }
};

typedef native_curve<ZKHIP_BLS12_381> bls12_381;
typedef native_curve<ZKHIP_BN254> alt_bn128_254;

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_ALGEBRA_HPP
