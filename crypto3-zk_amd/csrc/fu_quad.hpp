// G1 group law spread over a LANE QUAD, for the latency-bound tail of SMALL MSMs (device only).
//
// fu_pair.hpp halves the dependent products of a tail operation by letting the two lanes of a pair compute different products
// of the formula.  Where the tail has lanes to spare even as pairs (below 2^18 buckets: a 2^16-point MSM reduces 2^14 buckets --
// 16 384 pairs are half a wave per SIMD) four lanes can share a point: all four hold the SAME full XYZZ operands, every step
// computes up to four different products (operands picked by two v_cndmask levels, results broadcast with
// v_mov_dpp quad_perm [k,k,k,k]):
//   addition  4 product steps (pairs: 7, one lane: 14)        doubling  3 (pairs: 4, one lane: 8)
// A step costs one Montgomery product (498 VALU instructions) + 84 selects + 56 DPP moves against the pair's 42 + 14: ~640
// against ~570 instructions, so an addition shrinks from ~4 000 to ~2 600 instructions per lane and a doubling from ~2 300 to
// ~1 900 -- on a path whose length IS the cost (a lone wave issues one VALU instruction per ~5 cycles whatever the dependencies).
// All four lanes follow the same control flow: they hold the same values, so every test agrees.
#pragma once
#include "fu_pair.hpp"

namespace zkhip {

template <class U>
struct FuQ {  // a base-field element held (identically) by all four lanes of a quad
    typedef U params;
    Fu<U> v;
    ZK_D static uint32_t lane() { return threadIdx.x & 3u; }
    ZK_D static FuQ zero() { return {Fu<U>::zero()}; }
    ZK_D static FuQ one() { return {Fu<U>::one()}; }
};

// plain field interface (each lane computes the full operation: off the hot path -- conversions, rare branches)
template <class U>
struct FieldOps<FuQ<U>> {
    typedef FuQ<U> F;
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
    ZK_D static void store(uint32_t *p, const F &a) { fu_store<U>(p, a.v); }  // the four lanes write the same words
    ZK_D static void to_canonical(uint32_t *sat, const F &a) { fu_to_canonical<U>(sat, a.v); }
};

// the value lane K of the quad holds, on all four lanes
// (The result is pinned in a VGPR of its own: left to itself LLVM's DPP-combine pass folds the move into the consuming VALU
// instruction, and for `p0 - p1` -- two different broadcasts of ONE source register meeting in one subtraction -- the folded code
// computed a wrong difference on gfx950 / ROCm 7.2: Y3 of both group operations, caught by tests/cpp/quadtest.hip.)
template <int K>
ZK_D uint32_t quad_bcast(uint32_t x) {
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, K * 0x55 /* quad_perm [K,K,K,K] */, 0xF, 0xF, false);
    asm volatile("" : "+v"(r));
    return r;
}
template <int K, class U>
ZK_D Fu<U> quad_bcast(const Fu<U> &a) {
    Fu<U> r;
#pragma unroll
    for (int i = 0; i < U::L; ++i) r.v[i] = quad_bcast<K>(a.v[i]);
    return r;
}
template <class U>
ZK_D Fu<U> quad_select(uint32_t lane, const Fu<U> &a0, const Fu<U> &a1, const Fu<U> &a2, const Fu<U> &a3) {
    const bool b0 = (lane & 1u) != 0, b1 = (lane & 2u) != 0;
    return fu_select(b1, fu_select(b0, a3, a2), fu_select(b0, a1, a0));
}

// one product step of a quad: lane k forms ak * bk; returns the four results on ALL lanes
template <class U>
struct QuadProducts {
    Fu<U> p0, p1, p2, p3;
};
template <class U>
ZK_D QuadProducts<U> quad_mul(const Fu<U> &a0, const Fu<U> &b0, const Fu<U> &a1, const Fu<U> &b1, const Fu<U> &a2, const Fu<U> &b2, const Fu<U> &a3,
                              const Fu<U> &b3) {
    const uint32_t lane = FuQ<U>::lane();
    const Fu<U> mine = fu_mul(quad_select(lane, a0, a1, a2, a3), quad_select(lane, b0, b1, b2, b3));
    return {quad_bcast<0>(mine), quad_bcast<1>(mine), quad_bcast<2>(mine), quad_bcast<3>(mine)};
}

// 2 a  (curve.hpp xyzz_dbl, same formulas; Y3 as a difference of two reduced products like the non-Fu field types): 3 product steps
template <class U>
ZK_D XYZZ<FuQ<U>> xyzz_dbl(const XYZZ<FuQ<U>> &a) {
    typedef FieldOps<Fu<U>> O;
    if (a.is_inf()) return XYZZ<FuQ<U>>::infinity();
    const Fu<U> Uu = fu_add(a.Y.v, a.Y.v);
    const QuadProducts<U> s1 = quad_mul(Uu, Uu, a.X.v, a.X.v, Uu, Uu, a.X.v, a.X.v);  // V = U^2 | XX = X^2 | (the same again)
    const Fu<U> &V = s1.p0, &XX = s1.p1;
    const Fu<U> M = fu_add(fu_add(XX, XX), XX);
    const QuadProducts<U> s2 = quad_mul(Uu, V, a.X.v, V, V, a.ZZ.v, M, M);  // W = U V | S = X V | ZZ3 = V ZZ | M^2
    const Fu<U> &W = s2.p0, &S = s2.p1;
    const Fu<U> X3 = fu_sub<O::K1>(s2.p3, fu_add(S, S));
    const Fu<U> D = fu_sub<O::K2>(S, X3);
    const QuadProducts<U> s3 = quad_mul(M, D, W, a.Y.v, W, a.ZZZ.v, M, D);  // M (S - X3) | W Y | ZZZ3 = W ZZZ | (idle)
    return {{X3}, {fu_sub<O::K1>(s3.p0, s3.p1)}, {s2.p2}, {s3.p2}};
}

// a + b  (curve.hpp xyzz_add): 4 product steps
template <class U>
ZK_D XYZZ<FuQ<U>> xyzz_add(const XYZZ<FuQ<U>> &a, const XYZZ<FuQ<U>> &b) {
    typedef FieldOps<Fu<U>> O;
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    const QuadProducts<U> s1 = quad_mul(a.X.v, b.ZZ.v, b.X.v, a.ZZ.v, a.Y.v, b.ZZZ.v, b.Y.v, a.ZZZ.v);  // U1 | U2 | S1 | S2
    const Fu<U> &U1 = s1.p0, &S1 = s1.p2;
    const Fu<U> Pd = fu_sub<O::K1>(s1.p1, U1), R = fu_sub<O::K1>(s1.p3, S1);
    const QuadProducts<U> s2 = quad_mul(Pd, Pd, R, R, a.ZZ.v, b.ZZ.v, a.ZZZ.v, b.ZZZ.v);  // PP | R^2 | ZZab | ZZZab
    const Fu<U> &PP = s2.p0;
    if (fu_is_zero_lt2p(PP)) {  // same x: doubling or cancellation (rare)
        if (fu_canon(R).limbs_zero()) return xyzz_dbl(a);
        return XYZZ<FuQ<U>>::infinity();
    }
    const QuadProducts<U> s3 = quad_mul(Pd, PP, U1, PP, s2.p2, PP, Pd, PP);  // PPP | Q | ZZ3 = ZZab PP | (idle)
    const Fu<U> &PPP = s3.p0, &Q = s3.p1;
    const Fu<U> X3 = fu_sub<O::K1>(s2.p1, fu_add(PPP, fu_add(Q, Q)));
    const Fu<U> D = fu_sub<O::K2>(Q, X3);
    const QuadProducts<U> s4 = quad_mul(R, D, S1, PPP, s2.p3, PPP, R, D);  // R (Q - X3) | S1 PPP | ZZZ3 = ZZZab PPP | (idle)
    return {{X3}, {fu_sub<O::K1>(s4.p0, s4.p1)}, {s3.p2}, {s4.p2}};
}

// k * a, bit by bit (the 2-bit-digit version of curve.hpp keeps four full points live: see fu_pair.hpp)
template <class U>
ZK_D XYZZ<FuQ<U>> xyzz_mul_small(const XYZZ<FuQ<U>> &a, uint32_t k) {
    XYZZ<FuQ<U>> r = XYZZ<FuQ<U>>::infinity();
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

// the quad type of a bucket coordinate field, where one exists (G1)
template <class F>
struct QuadLane {
    static constexpr bool AVAILABLE = false;
    typedef typename TailLane<F>::type type;
    static constexpr int LANES = TailLane<F>::LANES;
};
template <class U>
struct QuadLane<Fu<U>> {
    static constexpr bool AVAILABLE = true;
    typedef FuQ<U> type;
    static constexpr int LANES = 4;
};

}  // namespace zkhip
