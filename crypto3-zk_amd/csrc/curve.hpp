// Short-Weierstrass y^2 = x^3 + b (a = 0) group law in extended Jacobian ("XYZZ") coordinates,
// generic over the coordinate field F through FieldOps<F> (fu.hpp): the lazy 29-bit-limb types Fu / Fu2
// that the kernels compute in, or the exactly-reduced saturated types Fp / Fp2 (reference path).
//
//   x = X / ZZ,  y = Y / ZZZ,  ZZ^3 = ZZZ^2;   infinity <=> ZZ is the exact zero
//
// XYZZ mixed addition costs 8M + 2S (vs 7M + 4S for Jacobian) and needs no special-casing of Z = 1,
// which is why the bucket accumulators of the Pippenger kernels (msm.hip) use it.  All formulas are
// complete for the cases the bucket method meets: P + P (doubling), P + (-P) (infinity), and either
// operand at infinity.  (Points of order 2 do not exist: all four groups have odd order.)
// The reference reaches the group law through crypto3-algebra (`G::value_type` operator+, `mixed_add`:
// knowledge_commitment_multiexp.hpp:91-97); a sum is coordinate-system independent once normalised to
// affine, which is what parity is asserted on.
//
// Lazy-reduction bounds (units of p; MULB = bound of a product: 2 for Fu, 10 for Fu2):
//   stored X, Y < MULB + K1;  stored ZZ, ZZZ < MULB;  affine inputs < MULB
//   sub<K1>: subtrahend is a sum of at most three products (< 3 MULB <= K1 - 1)
//   sub<K2>: subtrahend is a stored X / Y or another K1-difference (< MULB + K1 <= K2 - 1)
//   mul_sub<K>(a, b, c, d) = a b - c d with c <= (K - 1) p: for Fu both products share ONE Montgomery reduction
//       (fu_mul2) and the result is < 2p; the other field types compute sub<K1>(a b, c d)
#pragma once
#include "fu.hpp"

namespace zkhip {

template <class F>
struct Affine {  // (0, 0) encodes infinity: it is never on y^2 = x^3 + b with b != 0
    F x, y;
    ZK_HD bool is_inf() const { return FieldOps<F>::is_exact_zero(x) && FieldOps<F>::is_exact_zero(y); }
    ZK_HD static Affine infinity() { return {F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
    F X, Y, ZZ, ZZZ;
    ZK_HD bool is_inf() const { return FieldOps<F>::is_exact_zero(ZZ); }
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
    typedef FieldOps<F> O;
    if (p.is_inf()) return XYZZ<F>::infinity();
    F U = O::add(p.y, p.y);
    F V = O::sqr(U);
    F W = O::mul(U, V);
    F S = O::mul(p.x, V);
    F xx = O::sqr(p.x);
    F M = O::add(O::add(xx, xx), xx);
    F X3 = O::template sub<O::K1>(O::sqr(M), O::add(S, S));
    F Y3 = O::template mul_sub<O::K2>(M, O::template sub<O::K2>(S, X3), W, p.y);
    return {X3, Y3, V, W};
}

// 2 * a
template <class F>
ZK_HD XYZZ<F> xyzz_dbl(const XYZZ<F> &a) {
    typedef FieldOps<F> O;
    if (a.is_inf()) return XYZZ<F>::infinity();
    F U = O::add(a.Y, a.Y);
    F V = O::sqr(U);
    F W = O::mul(U, V);
    F S = O::mul(a.X, V);
    F xx = O::sqr(a.X);
    F M = O::add(O::add(xx, xx), xx);
    F X3 = O::template sub<O::K1>(O::sqr(M), O::add(S, S));
    F Y3 = O::template mul_sub<O::K2>(M, O::template sub<O::K2>(S, X3), W, a.Y);
    return {X3, Y3, O::mul(V, a.ZZ), O::mul(W, a.ZZZ)};
}

// a + (affine p); `negate` adds -p (signed-digit buckets)
template <class F>
ZK_HD XYZZ<F> xyzz_madd(const XYZZ<F> &a, const Affine<F> &p_in, bool negate = false) {
    typedef FieldOps<F> O;
    if (p_in.is_inf()) return a;
    Affine<F> p = p_in;
    if (negate) p.y = O::template sub<O::K1>(F::zero(), p.y);
    if (a.is_inf()) return {p.x, p.y, F::one(), F::one()};
    F U2 = O::mul(p.x, a.ZZ);
    F S2 = O::mul(p.y, a.ZZZ);
    F Pd = O::template sub<O::K2>(U2, a.X);
    F R = O::template sub<O::K2>(S2, a.Y);
    F PP = O::sqr(Pd);
    if (O::is_zero_product(PP)) {  // Pd = 0 mod p  <=>  same x: doubling or cancellation (rare)
        if (O::is_zero(R)) return xyzz_dbl_affine(p);
        return XYZZ<F>::infinity();
    }
    F PPP = O::mul(Pd, PP);
    F Q = O::mul(a.X, PP);
    F X3 = O::template sub<O::K1>(O::sqr(R), O::add(PPP, O::add(Q, Q)));
    F Y3 = O::template mul_sub<O::K2>(R, O::template sub<O::K2>(Q, X3), a.Y, PPP);
    return {X3, Y3, O::mul(a.ZZ, PP), O::mul(a.ZZZ, PPP)};
}

// a + b
template <class F>
ZK_HD XYZZ<F> xyzz_add(const XYZZ<F> &a, const XYZZ<F> &b) {
    typedef FieldOps<F> O;
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    F U1 = O::mul(a.X, b.ZZ);
    F U2 = O::mul(b.X, a.ZZ);
    F S1 = O::mul(a.Y, b.ZZZ);
    F S2 = O::mul(b.Y, a.ZZZ);
    F Pd = O::template sub<O::K1>(U2, U1);
    F R = O::template sub<O::K1>(S2, S1);
    F PP = O::sqr(Pd);
    if (O::is_zero_product(PP)) {
        if (O::is_zero(R)) return xyzz_dbl(a);
        return XYZZ<F>::infinity();
    }
    F PPP = O::mul(Pd, PP);
    F Q = O::mul(U1, PP);
    F X3 = O::template sub<O::K1>(O::sqr(R), O::add(PPP, O::add(Q, Q)));
    F Y3 = O::template mul_sub<O::K1>(R, O::template sub<O::K2>(Q, X3), S1, PPP);
    return {X3, Y3, O::mul(O::mul(a.ZZ, b.ZZ), PP), O::mul(O::mul(a.ZZZ, b.ZZZ), PPP)};
}

// k * a for a small unsigned k (bucket weights / segment offsets, k < 2^32): left-to-right over 2-bit digits with the
// multiples a, 2a, 3a.  The lanes of a wave hold different k, so a bit-by-bit double-and-add executes its addition at
// EVERY bit (some lane always has it set): 2-bit digits halve the additions -- b doublings + b/2 + 2 additions
// instead of b + b for a b-bit k -- on a path whose length is what the MSM's tail costs (msm_core.hpp).
template <class F>
ZK_HD XYZZ<F> xyzz_mul_small(const XYZZ<F> &a, uint32_t k) {
    XYZZ<F> r = XYZZ<F>::infinity();
    if (k == 0 || a.is_inf()) return r;
    int top = 31;
    while (!((k >> top) & 1)) --top;
    if (top == 0) return a;
    const XYZZ<F> a2 = xyzz_dbl(a), a3 = xyzz_add(a2, a);
    for (int i = top >> 1; i >= 0; --i) {
        if (!r.is_inf()) r = xyzz_dbl(xyzz_dbl(r));
        const uint32_t d = (k >> (2 * i)) & 3u;
        if (d != 0) r = xyzz_add(r, d == 1 ? a : (d == 2 ? a2 : a3));
    }
    return r;
}

// XYZZ -> Jacobian without inversion: Z = ZZ*ZZZ, X' = X*ZZ^4, Y' = Y*ZZZ^4
// (Z^2 = ZZ^5 = ZZ * ZZ^4, Z^3 = ZZZ^5 = ZZZ * ZZZ^4, using ZZ^3 = ZZZ^2)
template <class F>
ZK_HD Jacobian<F> xyzz_to_jacobian(const XYZZ<F> &a) {
    typedef FieldOps<F> O;
    if (a.is_inf()) return {F::one(), F::one(), F::zero()};
    F z2 = O::sqr(a.ZZ);
    F z3 = O::sqr(a.ZZZ);
    return {O::mul(a.X, O::sqr(z2)), O::mul(a.Y, O::sqr(z3)), O::mul(a.ZZ, a.ZZZ)};
}

// Jacobian (X, Y, Z) is XYZZ (X, Y, Z^2, Z^3)
template <class F>
ZK_HD XYZZ<F> xyzz_from_jacobian(const Jacobian<F> &j) {
    typedef FieldOps<F> O;
    if (O::is_zero(j.Z)) return XYZZ<F>::infinity();
    F zz = O::sqr(j.Z);
    return {j.X, j.Y, zz, O::mul(zz, j.Z)};
}

template <class F>
ZK_HD Affine<F> xyzz_to_affine(const XYZZ<F> &a) {
    typedef FieldOps<F> O;
    if (a.is_inf()) return Affine<F>::infinity();
    F i = O::inv(O::mul(a.ZZ, a.ZZZ));
    return {O::mul(a.X, O::mul(i, a.ZZZ)), O::mul(a.Y, O::mul(i, a.ZZ))};
}

// device-buffer layout: Affine = x | y, XYZZ = X | Y | ZZ | ZZZ, each FieldOps<F>::WORDS u32 words
template <class F>
ZK_HD Affine<F> affine_load(const uint32_t *p) {
    typedef FieldOps<F> O;
    return {O::load(p), O::load(p + O::WORDS)};
}
template <class F>
ZK_HD void affine_store(uint32_t *p, const Affine<F> &a) {
    typedef FieldOps<F> O;
    O::store(p, a.x);
    O::store(p + O::WORDS, a.y);
}
template <class F>
ZK_HD XYZZ<F> xyzz_load(const uint32_t *p) {
    typedef FieldOps<F> O;
    constexpr int W = O::WORDS;
    return {O::load(p), O::load(p + W), O::load(p + 2 * W), O::load(p + 3 * W)};
}
template <class F>
ZK_HD void xyzz_store(uint32_t *p, const XYZZ<F> &a) {
    typedef FieldOps<F> O;
    constexpr int W = O::WORDS;
    O::store(p, a.X);
    O::store(p + W, a.Y);
    O::store(p + 2 * W, a.ZZ);
    O::store(p + 3 * W, a.ZZZ);
}

// (curve, group) -> coordinate field the kernels compute in
template <int CURVE, int GROUP>
struct CurveTraits;
template <>
struct CurveTraits<CURVE_BLS12_381, GROUP_G1> {
    typedef bls_fqu F;
};
template <>
struct CurveTraits<CURVE_BLS12_381, GROUP_G2> {
    typedef bls_fqu2 F;
};
template <>
struct CurveTraits<CURVE_BN254, GROUP_G1> {
    typedef bn_fqu F;
};
template <>
struct CurveTraits<CURVE_BN254, GROUP_G2> {
    typedef bn_fqu2 F;
};

}  // namespace zkhip
