// G1 group law spread over an even / odd LANE PAIR, for the latency-bound tail of the MSM (device only).
//
// The bucket reduction, the per-set sums and the final conversion are chains of full XYZZ additions and doublings
// that a wave works through alone on its SIMD: ~13.6 us per operation, 60-80 operations deep (msm_core.hpp,
// msm_tail_segment) -- a third of a 2^20-point MSM.  The chain cannot be shortened much, but one operation can: its
// 12M + 2S (addition) or 6M + 3S (doubling) have a dependency depth of only 5-6 products.  Here both lanes of a pair hold
// the SAME full XYZZ operands; at every step the even and the odd lane compute different products of the formula (one
// instruction stream, operands picked by v_cndmask) and swap results with v_mov_dpp quad_perm [1,0,3,2]:
//   addition  7 product steps instead of 14   doubling  4 instead of 8
// Twice the lanes do the same total work in about half the time -- exactly what a latency-bound phase wants (the
// accumulation kernel, bound by issue slots, keeps one lane per bucket).  G2 already splits Fq2 across a pair
// (fu2_pair.hpp); this is the same idea one level up, on the group law instead of the field.
// Both lanes of a pair always follow the same control flow: they hold the same values, so every test agrees.
#pragma once
#include "curve.hpp"
#include "fu2_pair.hpp"

namespace zkhip {

template <class U>
struct FuP {  // a base-field element held (identically) by both lanes of a pair
    typedef U params;
    Fu<U> v;
    ZK_D static bool odd() { return (threadIdx.x & 1u) != 0; }
    ZK_D static FuP zero() { return {Fu<U>::zero()}; }
    ZK_D static FuP one() { return {Fu<U>::one()}; }
};

// plain field interface (each lane computes the full operation: used off the hot path -- conversions, rare branches)
template <class U>
struct FieldOps<FuP<U>> {
    typedef FuP<U> F;
    typedef FieldOps<Fu<U>> B;
    static constexpr int K1 = B::K1, K2 = B::K2, K3 = B::K3;
    static constexpr int WORDS = U::SL;
    static constexpr int CANON_WORDS = U::NL;
    ZK_D static F mul(const F &a, const F &b) { return {fu_mul_call(a.v, b.v)}; }
    ZK_D static F sqr(const F &a) { return {fu_mul_call(a.v, a.v)}; }
    ZK_D static F add(const F &a, const F &b) { return {fu_add(a.v, b.v)}; }
    template <int K>
    ZK_D static F sub(const F &a, const F &b) { return {fu_sub<K>(a.v, b.v)}; }
    template <int K>
    ZK_D static F mul_sub(const F &a, const F &b, const F &c, const F &d) { return {fu_mul2_call(a.v, b.v, fu_sub<K>(Fu<U>::zero(), c.v), d.v)}; }
    ZK_D static bool is_zero(const F &a) { return fu_canon(a.v).limbs_zero(); }
    ZK_D static bool is_zero_product(const F &a) { return fu_is_zero_lt2p(a.v); }
    ZK_D static bool is_exact_zero(const F &a) { return a.v.limbs_zero(); }
    ZK_D static F load(const uint32_t *p) { return {fu_load<U>(p)}; }
    ZK_D static void store(uint32_t *p, const F &a) { fu_store<U>(p, a.v); }  // both lanes write the same words
    ZK_D static void to_canonical(uint32_t *sat, const F &a) { fu_to_canonical<U>(sat, a.v); }
};

// one product step of a pair: the even lane forms ae * be, the odd lane ao * bo; returns {even's result, odd's result} on BOTH lanes
template <class U>
struct PairProducts {
    Fu<U> even, odd;
};
template <class U>
ZK_D PairProducts<U> pair_mul(const Fu<U> &ae, const Fu<U> &be, const Fu<U> &ao, const Fu<U> &bo) {
    const bool odd = FuP<U>::odd();
    const Fu<U> mine = fu_mul(fu_select(odd, ao, ae), fu_select(odd, bo, be));
    const Fu<U> other = pair_swap(mine);
    return {fu_select(odd, other, mine), fu_select(odd, mine, other)};
}
// even: ae^2, odd: ao^2
template <class U>
ZK_D PairProducts<U> pair_sqr(const Fu<U> &ae, const Fu<U> &ao) {
    const bool odd = FuP<U>::odd();
    const Fu<U> mine = fu_sqr(fu_select(odd, ao, ae));
    const Fu<U> other = pair_swap(mine);
    return {fu_select(odd, other, mine), fu_select(odd, mine, other)};
}
// even: ae * be + ce * de;  odd: ao * bo  (the odd lane's second product is 0 * 0: same instruction stream)
template <class U>
ZK_D PairProducts<U> pair_mul2_mul(const Fu<U> &ae, const Fu<U> &be, const Fu<U> &ce, const Fu<U> &de, const Fu<U> &ao, const Fu<U> &bo) {
    const bool odd = FuP<U>::odd();
    const Fu<U> z = Fu<U>::zero();
    const Fu<U> mine = fu_mul2(fu_select(odd, ao, ae), fu_select(odd, bo, be), fu_select(odd, z, ce), fu_select(odd, z, de));
    const Fu<U> other = pair_swap(mine);
    return {fu_select(odd, other, mine), fu_select(odd, mine, other)};
}

// 2 a  (curve.hpp xyzz_dbl, same formulas and lazy bounds): 4 product steps
template <class U>
ZK_D XYZZ<FuP<U>> xyzz_dbl(const XYZZ<FuP<U>> &a) {
    typedef FieldOps<Fu<U>> O;
    if (a.is_inf()) return XYZZ<FuP<U>>::infinity();
    const Fu<U> Uu = fu_add(a.Y.v, a.Y.v);
    const PairProducts<U> s1 = pair_sqr(Uu, a.X.v);  // V = U^2 | XX = X^2
    const Fu<U> &V = s1.even, &XX = s1.odd;
    const PairProducts<U> s2 = pair_mul(Uu, V, a.X.v, V);  // W = U V | S = X V
    const Fu<U> &W = s2.even, &S = s2.odd;
    const Fu<U> M = fu_add(fu_add(XX, XX), XX);
    const PairProducts<U> s3 = pair_mul(M, M, V, a.ZZ.v);  // M^2 | ZZ3 = V ZZ
    const Fu<U> X3 = fu_sub<O::K1>(s3.even, fu_add(S, S));
    // Y3 = M (S - X3) - W Y | ZZZ3 = W ZZZ
    const PairProducts<U> s4 = pair_mul2_mul(M, fu_sub<O::K2>(S, X3), fu_sub<O::K2>(Fu<U>::zero(), W), a.Y.v, W, a.ZZZ.v);
    return {{X3}, {s4.even}, {s3.odd}, {s4.odd}};
}

// a + b  (curve.hpp xyzz_add): 7 product steps
template <class U>
ZK_D XYZZ<FuP<U>> xyzz_add(const XYZZ<FuP<U>> &a, const XYZZ<FuP<U>> &b) {
    typedef FieldOps<Fu<U>> O;
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    const PairProducts<U> s1 = pair_mul(a.X.v, b.ZZ.v, b.X.v, a.ZZ.v);    // U1 | U2
    const PairProducts<U> s2 = pair_mul(a.Y.v, b.ZZZ.v, b.Y.v, a.ZZZ.v);  // S1 | S2
    const Fu<U> &U1 = s1.even, &S1 = s2.even;
    const Fu<U> Pd = fu_sub<O::K1>(s1.odd, U1), R = fu_sub<O::K1>(s2.odd, S1);
    const PairProducts<U> s3 = pair_sqr(Pd, R);  // PP | R^2
    const Fu<U> &PP = s3.even;
    if (fu_is_zero_lt2p(PP)) {  // same x: doubling or cancellation (rare)
        if (fu_canon(R).limbs_zero()) return xyzz_dbl(a);
        return XYZZ<FuP<U>>::infinity();
    }
    const PairProducts<U> s4 = pair_mul(Pd, PP, U1, PP);  // PPP | Q
    const Fu<U> &PPP = s4.even, &Q = s4.odd;
    const Fu<U> X3 = fu_sub<O::K1>(s3.odd, fu_add(PPP, fu_add(Q, Q)));
    // Y3 = R (Q - X3) - S1 PPP | ZZab = a.ZZ b.ZZ
    const PairProducts<U> s5 = pair_mul2_mul(R, fu_sub<O::K2>(Q, X3), fu_sub<O::K1>(Fu<U>::zero(), S1), PPP, a.ZZ.v, b.ZZ.v);
    const PairProducts<U> s6 = pair_mul(a.ZZZ.v, b.ZZZ.v, s5.odd, PP);  // ZZZab | ZZ3 = ZZab PP
    const PairProducts<U> s7 = pair_mul(s6.even, PPP, s6.even, PPP);    // ZZZ3 = ZZZab PPP (both lanes)
    return {{X3}, {s5.even}, {s6.odd}, {s7.even}};
}

// k * a, bit by bit.  The generic 2-bit-digit version (curve.hpp) keeps a, 2a, 3a and the running value live: four full
// XYZZ points are 224 VGPRs on this type (every lane holds whole coordinates), the tail kernels spill and run slower than
// with the plain double-and-add (measured: bucket_red 0.88 ms against 0.73 at 2^19 buckets); G2's pair type holds half a
// coordinate per lane and takes the digits (1.92 -> 1.78 ms).
template <class U>
ZK_D XYZZ<FuP<U>> xyzz_mul_small(const XYZZ<FuP<U>> &a, uint32_t k) {
    XYZZ<FuP<U>> r = XYZZ<FuP<U>>::infinity();
    if (k == 0 || a.is_inf()) return r;
    int top = 31;
    while (!((k >> top) & 1)) --top;
    r = a;
    for (int i = top - 1; i >= 0; --i) {
        r = xyzz_dbl(r);
        if ((k >> i) & 1) r = xyzz_add(r, a);
    }
    return r;
}

// what a lane of the TAIL kernels holds for a bucket coordinate field F, and how many lanes share a point
template <class F>
struct TailLane : BucketLane<F> { };
template <class U>
struct TailLane<Fu<U>> {
    typedef FuP<U> type;
    static constexpr int LANES = 2;
};

}  // namespace zkhip
