// Short-Weierstrass y^2 = x^3 + b (a = 0) group law in extended Jacobian ("XYZZ") coordinates,
// generic over the coordinate field F (Fq for G1, Fq2 for G2).
//
//   x = X / ZZ,  y = Y / ZZZ,  ZZ^3 = ZZZ^2;   infinity <=> ZZ == 0
//
// XYZZ mixed addition costs 8M + 2S (vs 7M + 4S for Jacobian) and needs no special-casing of Z = 1,
// which is why the bucket accumulators of the Pippenger kernels (msm.hip) use it.  All formulas are
// complete for the cases the bucket method meets: P + P (doubling), P + (-P) (infinity), and either
// operand at infinity.  The reference reaches the group law through crypto3-algebra
// (`G::value_type` operator+, `mixed_add`: knowledge_commitment_multiexp.hpp:91-97); the result of a
// sum is coordinate-system independent once normalised to affine, which is what parity is asserted on.
#pragma once
#include "fp.hpp"

namespace zkhip {

template <class F>
struct Affine {  // (0, 0) encodes infinity: it is never on y^2 = x^3 + b with b != 0
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    ZK_HD static Affine infinity() { return {F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
    F X, Y, ZZ, ZZZ;
    ZK_HD bool is_inf() const { return ZZ.is_zero(); }
    ZK_HD static XYZZ infinity() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_HD static XYZZ from_affine(const Affine<F> &p) {
        if (p.is_inf()) return infinity();
        return {p.x, p.y, F::one(), F::one()};
    }
};

template <class F>
struct Jacobian {
    F X, Y, Z;
};

// 2 * (affine p)
template <class F>
ZK_HD XYZZ<F> xyzz_dbl_affine(const Affine<F> &p) {
    if (p.is_inf() || p.y.is_zero()) return XYZZ<F>::infinity();
    F U = fp_dbl(p.y);
    F V = fp_sqr(U);
    F W = U * V;
    F S = p.x * V;
    F xx = fp_sqr(p.x);
    F M = fp_dbl(xx) + xx;
    F X3 = fp_sqr(M) - fp_dbl(S);
    F Y3 = M * (S - X3) - W * p.y;
    return {X3, Y3, V, W};
}

// 2 * a
template <class F>
ZK_HD XYZZ<F> xyzz_dbl(const XYZZ<F> &a) {
    if (a.is_inf() || a.Y.is_zero()) return XYZZ<F>::infinity();
    F U = fp_dbl(a.Y);
    F V = fp_sqr(U);
    F W = U * V;
    F S = a.X * V;
    F xx = fp_sqr(a.X);
    F M = fp_dbl(xx) + xx;
    F X3 = fp_sqr(M) - fp_dbl(S);
    F Y3 = M * (S - X3) - W * a.Y;
    return {X3, Y3, V * a.ZZ, W * a.ZZZ};
}

// a + (affine p); `negate` adds -p (signed-digit buckets)
template <class F>
ZK_HD XYZZ<F> xyzz_madd(const XYZZ<F> &a, const Affine<F> &p_in, bool negate = false) {
    if (p_in.is_inf()) return a;
    Affine<F> p = p_in;
    if (negate) p.y = fp_neg(p.y);
    if (a.is_inf()) return {p.x, p.y, F::one(), F::one()};
    F U2 = p.x * a.ZZ;
    F S2 = p.y * a.ZZZ;
    F Pd = U2 - a.X;
    F R = S2 - a.Y;
    if (Pd.is_zero()) {
        if (R.is_zero()) return xyzz_dbl_affine(p);
        return XYZZ<F>::infinity();
    }
    F PP = fp_sqr(Pd);
    F PPP = Pd * PP;
    F Q = a.X * PP;
    F X3 = fp_sqr(R) - PPP - fp_dbl(Q);
    F Y3 = R * (Q - X3) - a.Y * PPP;
    return {X3, Y3, a.ZZ * PP, a.ZZZ * PPP};
}

// a + b
template <class F>
ZK_HD XYZZ<F> xyzz_add(const XYZZ<F> &a, const XYZZ<F> &b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    F U1 = a.X * b.ZZ;
    F U2 = b.X * a.ZZ;
    F S1 = a.Y * b.ZZZ;
    F S2 = b.Y * a.ZZZ;
    F Pd = U2 - U1;
    F R = S2 - S1;
    if (Pd.is_zero()) {
        if (R.is_zero()) return xyzz_dbl(a);
        return XYZZ<F>::infinity();
    }
    F PP = fp_sqr(Pd);
    F PPP = Pd * PP;
    F Q = U1 * PP;
    F X3 = fp_sqr(R) - PPP - fp_dbl(Q);
    F Y3 = R * (Q - X3) - S1 * PPP;
    return {X3, Y3, a.ZZ * b.ZZ * PP, a.ZZZ * b.ZZZ * PPP};
}

template <class F>
ZK_HD XYZZ<F> xyzz_neg(const XYZZ<F> &a) {
    return {a.X, fp_neg(a.Y), a.ZZ, a.ZZZ};
}

// k * a for a small unsigned k (bucket-segment offsets, k < 2^32)
template <class F>
ZK_HD XYZZ<F> xyzz_mul_small(const XYZZ<F> &a, uint32_t k) {
    XYZZ<F> r = XYZZ<F>::infinity();
    for (int i = 31; i >= 0; --i) {
        r = xyzz_dbl(r);
        if ((k >> i) & 1) r = xyzz_add(r, a);
    }
    return r;
}

// XYZZ -> Jacobian without inversion: Z = ZZ*ZZZ, X' = X*ZZ^4, Y' = Y*ZZZ^4
// (Z^2 = ZZ^5 = ZZ * ZZ^4, Z^3 = ZZZ^5 = ZZZ * ZZZ^4, using ZZ^3 = ZZZ^2)
template <class F>
ZK_HD Jacobian<F> xyzz_to_jacobian(const XYZZ<F> &a) {
    if (a.is_inf()) return {F::one(), F::one(), F::zero()};
    F z2 = fp_sqr(a.ZZ);
    F z3 = fp_sqr(a.ZZZ);
    return {a.X * fp_sqr(z2), a.Y * fp_sqr(z3), a.ZZ * a.ZZZ};
}

template <class F>
ZK_HD Affine<F> xyzz_to_affine(const XYZZ<F> &a) {
    if (a.is_inf()) return Affine<F>::infinity();
    F i = fp_inv(a.ZZ * a.ZZZ);
    return {a.X * (i * a.ZZZ), a.Y * (i * a.ZZ)};
}

// memory layout helpers: Affine = x | y, XYZZ = X | Y | ZZ | ZZZ, each FieldIO<F>::NL u32 limbs
template <class F>
ZK_HD Affine<F> affine_load(const uint32_t *p) {
    return {FieldIO<F>::load(p), FieldIO<F>::load(p + FieldIO<F>::NL)};
}
template <class F>
ZK_HD void affine_store(uint32_t *p, const Affine<F> &a) {
    FieldIO<F>::store(p, a.x);
    FieldIO<F>::store(p + FieldIO<F>::NL, a.y);
}
template <class F>
ZK_HD XYZZ<F> xyzz_load(const uint32_t *p) {
    constexpr int NL = FieldIO<F>::NL;
    return {FieldIO<F>::load(p), FieldIO<F>::load(p + NL), FieldIO<F>::load(p + 2 * NL), FieldIO<F>::load(p + 3 * NL)};
}
template <class F>
ZK_HD void xyzz_store(uint32_t *p, const XYZZ<F> &a) {
    constexpr int NL = FieldIO<F>::NL;
    FieldIO<F>::store(p, a.X);
    FieldIO<F>::store(p + NL, a.Y);
    FieldIO<F>::store(p + 2 * NL, a.ZZ);
    FieldIO<F>::store(p + 3 * NL, a.ZZZ);
}

// (curve, group) -> coordinate field / scalar field
template <int CURVE, int GROUP>
struct CurveTraits;
template <>
struct CurveTraits<CURVE_BLS12_381, GROUP_G1> {
    typedef bls_fq F;
    typedef bls_fr S;
};
template <>
struct CurveTraits<CURVE_BLS12_381, GROUP_G2> {
    typedef bls_fq2 F;
    typedef bls_fr S;
};
template <>
struct CurveTraits<CURVE_BN254, GROUP_G1> {
    typedef bn_fq F;
    typedef bn_fr S;
};
template <>
struct CurveTraits<CURVE_BN254, GROUP_G2> {
    typedef bn_fq2 F;
    typedef bn_fr S;
};

}  // namespace zkhip
