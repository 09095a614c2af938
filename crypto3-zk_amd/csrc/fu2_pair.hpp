// Fq2 elements split across an even / odd lane pair (device only).
//
// One lane cannot hold a G2 mixed addition and still share its SIMD: the XYZZ formulas over Fu2 need ~430 VGPRs,
// one wave per SIMD, and a lone wave issues well below the SIMD's rate.  Here lane 2j holds the c0 component and
// lane 2j + 1 the c1 component of every Fq2 value of bucket j, so that a lane carries what a G1 lane carries
// (~170-250 VGPRs, 2-3 waves per SIMD) and twice as many lanes share the work:
//   add / sub      component-wise, no communication
//   mul            the pair swaps its halves (v_mov_dpp quad_perm [1,0,3,2]); the even lane forms a0 b0 - a1 b1, the
//                  odd lane a1 b0 + a0 b1, each as ONE shared-reduction product pair (fu_mul2): 3 L^2 mads per
//                  lane, 6 L^2 per pair -- what Karatsuba costs on a single lane
//   sqr            even: (a0 + a1)(a0 - a1), odd: (2 a1) a0 -- one product per lane
// Both lanes of a pair always follow the same control flow (same bucket, same trip count), which the cross-lane
// reads rely on.  The memory form is unchanged (c0 | c1, SL words each): FieldOps<Fu2h>::load / store address the
// lane's half, so the generic affine / XYZZ load-store helpers of curve.hpp work as they are.
#pragma once
#include "curve.hpp"

namespace zkhip {

template <class U>
struct Fu2h {
    typedef U params;
    Fu<U> v;  // this lane's component: c0 on even lanes, c1 on odd lanes
    ZK_D static bool odd() { return (threadIdx.x & 1u) != 0; }
    ZK_D static Fu2h zero() { return {Fu<U>::zero()}; }
    ZK_D static Fu2h one() { return {odd() ? Fu<U>::zero() : Fu<U>::one()}; }
};

// value held by the other lane of the pair
ZK_D uint32_t pair_swap(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false); }
template <class U>
ZK_D Fu<U> pair_swap(const Fu<U> &a) {
    Fu<U> r;
#pragma unroll
    for (int i = 0; i < U::L; ++i) r.v[i] = pair_swap(a.v[i]);
    return r;
}
template <class U>
ZK_D Fu<U> fu_select(bool c, const Fu<U> &a, const Fu<U> &b) {
    Fu<U> r;
#pragma unroll
    for (int i = 0; i < U::L; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// products out of line: a mixed addition holds eight of them and the bucket kernel's code has to stay inside the
// instruction cache; 28 argument registers fit the calling convention's 32
template <class U>
ZK_NOINLINE_D Fu<U> fu2h_mul_call(Fu<U> a, Fu<U> b) {
    const bool odd = Fu2h<U>::odd();
    const Fu<U> ao = pair_swap(a), bo = pair_swap(b);
    // even: a b - ao bo;  odd: a bo + ao b.  Operands are < 128p (see FieldOps<Fu2h>).
    const Fu<U> x2 = fu_select(odd, ao, fu_sub<128>(Fu<U>::zero(), ao));
    return fu_mul2(a, fu_select(odd, bo, b), x2, fu_select(odd, b, bo));
}
template <class U>
ZK_NOINLINE_D Fu<U> fu2h_sqr_call(Fu<U> a) {
    const bool odd = Fu2h<U>::odd();
    const Fu<U> ao = pair_swap(a);
    // even: (a + ao)(a - ao);  odd: (a + a) ao
    return fu_mul(fu_add(a, fu_select(odd, a, ao)), fu_select(odd, ao, fu_sub<128>(a, ao)));
}

// Bounds (units of p): products < 2, but the AFFINE INPUTS are whatever the single-lane Fu2 code stored in the bases
// and window tables: Fu2 products, < 10.  K1 therefore covers 10 (negating p.y, three products), stored X / Y are
// < 2 + K1 = 18 <= K2 - 1, and every product operand stays below 2 + K2 = 34 < 128 (the negation inside mul).
template <class U>
struct FieldOps<Fu2h<U>> {
    typedef Fu2h<U> F;
    static constexpr int K1 = 16, K2 = 32, K3 = 64;
    static constexpr int WORDS = 2 * U::SL;  // a full Fq2 coordinate in memory
    ZK_D static F mul(const F &a, const F &b) {
#ifdef ZK_PAIR_INLINE
        const bool odd = F::odd();
        const Fu<U> ao = pair_swap(a.v), bo = pair_swap(b.v);
        const Fu<U> x2 = fu_select(odd, ao, fu_sub<128>(Fu<U>::zero(), ao));
        return {fu_mul2(a.v, fu_select(odd, bo, b.v), x2, fu_select(odd, b.v, bo))};
#else
        return {fu2h_mul_call(a.v, b.v)};
#endif
    }
    ZK_D static F sqr(const F &a) {
#ifdef ZK_PAIR_INLINE
        const bool odd = F::odd();
        const Fu<U> ao = pair_swap(a.v);
        return {fu_mul(fu_add(a.v, fu_select(odd, a.v, ao)), fu_select(odd, ao, fu_sub<128>(a.v, ao)))};
#else
        return {fu2h_sqr_call(a.v)};
#endif
    }
    ZK_D static F add(const F &a, const F &b) { return {fu_add(a.v, b.v)}; }
    template <int K>
    ZK_D static F sub(const F &a, const F &b) { return {fu_sub<K>(a.v, b.v)}; }
    template <int K>
    ZK_D static F mul_sub(const F &a, const F &b, const F &c, const F &d) { return sub<K1>(mul(a, b), mul(c, d)); }
    // true iff it holds on both lanes of the pair; every lane executes the exchange (no short circuit around the DPP read)
    ZK_D static bool both(bool mine) {
        const uint32_t m = mine ? 1u : 0u, o = pair_swap(m);
        return (m & o) != 0;
    }
    ZK_D static bool is_zero(const F &a) { return both(fu_canon(a.v).limbs_zero()); }
    ZK_D static bool is_zero_product(const F &a) { return both(fu_is_zero_lt2p(a.v)); }
    ZK_D static bool is_exact_zero(const F &a) { return both(a.v.limbs_zero()); }
    static constexpr int CANON_WORDS = 2 * U::NL;  // canonical Fq2 at the boundary: c0 | c1
    ZK_D static void to_canonical(uint32_t *sat, const F &a) { fu_to_canonical<U>(sat + (F::odd() ? U::NL : 0), a.v); }
    ZK_D static F load(const uint32_t *p) { return {fu_load<U>(p + (F::odd() ? U::SL : 0))}; }
    ZK_D static void store(uint32_t *p, const F &a) { fu_store<U>(p + (F::odd() ? U::SL : 0), a.v); }
};

// coordinate type the bucket kernels compute in -> lanes per bucket and the type a lane holds
template <class F>
struct BucketLane {
    typedef F type;
    static constexpr int LANES = 1;
};
template <class U>
struct BucketLane<Fu2<U>> {
    typedef Fu2h<U> type;
    static constexpr int LANES = 2;
};

}  // namespace zkhip
