// Unsaturated-limb Montgomery field arithmetic: the compute representation of the gfx950 kernels.
//
// An element is L limbs of B = 29 bits held in u32 (value = sum v[i] 2^(B i)), Montgomery radix R = 2^(B L).
// Why: gfx950's widest integer multiply is v_mad_u64_u32 (32x32 + 64 -> 64).  With 32-bit limbs every limb
// product needs extra carry instructions (hipcc emits ~1.3k VALU ops for one 381-bit product, half of them
// moves and 64-bit adds); with 29-bit limbs up to 28 products of < 2^58 accumulate in one 64-bit register
// pair, so a column-wise (product-scanning) Montgomery product is exactly 2 L^2 v_mad_u64_u32 plus ~4 L cheap
// ops and measures 2.0x faster per wave (profiles/r01_mulbench.txt).
//
// Laziness: R has 25 (Fq) bits of slack above the modulus, so sums and differences are NOT reduced mod p;
// only limbs are re-normalised (carry propagation, 3 cheap ops per limb).  Contract:
//   fu mul      : limbs < 2^30 on both inputs, values a*b < R*p  ->  normalised limbs, value < 2p
//   fu add      : normalised inputs                               ->  normalised limbs, value = a + b
//   fu sub<K>   : b normalised with value(b) <= (K-1) p           ->  normalised limbs, value = a + K p - b
// Call sites pick K from the bound of the subtrahend (curve.hpp documents them); values never approach
// 2^(B L) (checked by the CPU tests with worst-case operands).  Equality / zero tests go through
// canonicalisation (one product by the Montgomery one and a conditional subtraction), or, for a value known
// to be < 2p, a comparison against {0, p}.
//
// At the library boundary and in HBM vectors of scalars everything stays canonical 32/64-bit limbs; bases,
// buckets and other device-internal points are stored in this form, padded to SL = 16 words (64 B) per
// coordinate so that a BLS12-381 G1 affine point is exactly one aligned 128-byte line.
#pragma once
#include "fp.hpp"

namespace zkhip {

template <class U>
struct Fu {
    typedef U params;
    static constexpr int L = U::L;
    static constexpr int B = U::B;
    static constexpr int SL = U::SL;  // storage words
    static constexpr uint32_t MASK = (1u << U::B) - 1;
    uint32_t v[L];

    ZK_HD static Fu zero() {
        Fu r;
ZK_UNROLL
        for (int i = 0; i < L; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fu one() {  // Montgomery form of 1 (R mod p)
        Fu r;
ZK_UNROLL
        for (int i = 0; i < L; ++i) r.v[i] = U::r1(i);
        return r;
    }
    ZK_HD static Fu r2() {
        Fu r;
ZK_UNROLL
        for (int i = 0; i < L; ++i) r.v[i] = U::r2(i);
        return r;
    }
    ZK_HD static Fu modulus() {
        Fu r;
ZK_UNROLL
        for (int i = 0; i < L; ++i) r.v[i] = U::mod(i);
        return r;
    }
    // all limbs zero (exact representation of 0; NOT a test for 0 mod p)
    ZK_HD bool limbs_zero() const {
        uint32_t o = 0;
ZK_UNROLL
        for (int i = 0; i < L; ++i) o |= v[i];
        return o == 0;
    }
    ZK_HD bool limbs_equal(const Fu &b) const {
        uint32_t o = 0;
ZK_UNROLL
        for (int i = 0; i < L; ++i) o |= v[i] ^ b.v[i];
        return o == 0;
    }
};

// carry propagation: limbs < 2^32 in, limbs < 2^B out (top limb keeps whatever is left)
template <class U>
ZK_HD void fu_norm(Fu<U> &a) {
    constexpr int L = U::L, B = U::B;
ZK_UNROLL
    for (int i = 0; i < L - 1; ++i) {
        a.v[i + 1] += a.v[i] >> B;
        a.v[i] &= Fu<U>::MASK;
    }
}

template <class U>
ZK_HD Fu<U> fu_add(const Fu<U> &a, const Fu<U> &b) {
    Fu<U> r;
ZK_UNROLL
    for (int i = 0; i < U::L; ++i) r.v[i] = a.v[i] + b.v[i];
    fu_norm(r);
    return r;
}

// a + K p - b;  b normalised, value(b) <= (K - 1) p
template <int K, class U>
ZK_HD Fu<U> fu_sub(const Fu<U> &a, const Fu<U> &b) {
    Fu<U> r;
ZK_UNROLL
    for (int i = 0; i < U::L; ++i) r.v[i] = a.v[i] + (U::template spread<K>(i) - b.v[i]);
    fu_norm(r);
    return r;
}

// Montgomery product, product scanning: column k accumulates a_i b_(k-i) and m_i q_(k-i) in one 64-bit register
template <class U>
ZK_HD Fu<U> fu_mul(const Fu<U> &a, const Fu<U> &b) {
    constexpr int L = U::L, B = U::B;
    constexpr uint32_t MASK = Fu<U>::MASK;
    uint32_t m[L];
    Fu<U> r;
    uint64_t acc = 0;
ZK_UNROLL
    for (int k = 0; k < L; ++k) {
ZK_UNROLL
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
ZK_UNROLL
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * U::mod(k - i);
        m[k] = ((uint32_t)acc * U::QINV) & MASK;
        acc += (uint64_t)m[k] * U::mod(0);
        acc >>= B;
    }
ZK_UNROLL
    for (int k = L; k < 2 * L - 1; ++k) {
ZK_UNROLL
        for (int i = k - L + 1; i < L; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
ZK_UNROLL
        for (int i = k - L + 1; i < L; ++i) acc += (uint64_t)m[i] * U::mod(k - i);
        r.v[k - L] = (uint32_t)acc & MASK;
        acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
}

// REDC(a b + c d): the two products share the column accumulators and ONE Montgomery reduction (3 L^2 mads
// instead of the 4 L^2 of two separate products).  Limbs normalised (< 2^B): a column holds at most 2L products
// < 2^58 plus L reduction terms, 3L 2^58 < 2^64 for L <= 21.
// a b + c d < R p  =>  result < 2p.
template <class U>
ZK_HD Fu<U> fu_mul2(const Fu<U> &a, const Fu<U> &b, const Fu<U> &c, const Fu<U> &d) {
    constexpr int L = U::L, B = U::B;
    constexpr uint32_t MASK = Fu<U>::MASK;
    static_assert(3 * L <= 63, "column accumulator would overflow");
    uint32_t m[L];
    Fu<U> r;
    uint64_t acc = 0;
ZK_UNROLL
    for (int k = 0; k < L; ++k) {
ZK_UNROLL
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
ZK_UNROLL
        for (int i = 0; i <= k; ++i) acc += (uint64_t)c.v[i] * d.v[k - i];
ZK_UNROLL
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * U::mod(k - i);
        m[k] = ((uint32_t)acc * U::QINV) & MASK;
        acc += (uint64_t)m[k] * U::mod(0);
        acc >>= B;
    }
ZK_UNROLL
    for (int k = L; k < 2 * L - 1; ++k) {
ZK_UNROLL
        for (int i = k - L + 1; i < L; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
ZK_UNROLL
        for (int i = k - L + 1; i < L; ++i) acc += (uint64_t)c.v[i] * d.v[k - i];
ZK_UNROLL
        for (int i = k - L + 1; i < L; ++i) acc += (uint64_t)m[i] * U::mod(k - i);
        r.v[k - L] = (uint32_t)acc & MASK;
        acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
}

// Montgomery square: the cross products a_i a_j (i != j) are taken once against the doubled operand, which
// saves L(L-1)/2 of the 2 L^2 mads.  Same contract as fu_mul, with normalised input limbs (< 2^B).
template <class U>
ZK_HD Fu<U> fu_sqr(const Fu<U> &a) {
    constexpr int L = U::L, B = U::B;
    constexpr uint32_t MASK = Fu<U>::MASK;
    uint32_t m[L], d[L];
ZK_UNROLL
    for (int i = 0; i < L; ++i) d[i] = a.v[i] << 1;
    Fu<U> r;
    uint64_t acc = 0;
ZK_UNROLL
    for (int k = 0; k < 2 * L - 1; ++k) {
        // a_i * (2 a_(k-i)) over i < k - i, both indices inside [0, L)
ZK_UNROLL
        for (int i = (k < L ? 0 : k - L + 1); 2 * i < k; ++i) acc += (uint64_t)a.v[i] * d[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
        if (k < L) {
ZK_UNROLL
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * U::mod(k - i);
            m[k] = ((uint32_t)acc * U::QINV) & MASK;
            acc += (uint64_t)m[k] * U::mod(0);
        } else {
ZK_UNROLL
            for (int i = k - L + 1; i < L; ++i) acc += (uint64_t)m[i] * U::mod(k - i);
            r.v[k - L] = (uint32_t)acc & MASK;
        }
        acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
}

// out-of-line copy; operands by value so that they travel in VGPRs, not through scratch
template <class U>
ZK_NOINLINE_HD Fu<U> fu_mul_call(Fu<U> a, Fu<U> b) {
    return fu_mul(a, b);
}
template <class U>
ZK_NOINLINE_HD Fu<U> fu_mul2_call(Fu<U> a, Fu<U> b, Fu<U> c, Fu<U> d) {
    return fu_mul2(a, b, c, d);
}

// value < 2p with normalised limbs: is it 0 mod p?
template <class U>
ZK_HD bool fu_is_zero_lt2p(const Fu<U> &a) {
    uint32_t z = 0, e = 0;
ZK_UNROLL
    for (int i = 0; i < U::L; ++i) {
        z |= a.v[i];
        e |= a.v[i] ^ U::mod(i);
    }
    return z == 0 || e == 0;
}

// value < 2p, normalised -> canonical representative in [0, p)
template <class U>
ZK_HD Fu<U> fu_cond_sub_p(const Fu<U> &a) {
    constexpr int L = U::L;
    Fu<U> d;
    uint32_t borrow = 0;
ZK_UNROLL
    for (int i = 0; i < L; ++i) {
        uint32_t t = a.v[i] - U::mod(i) - borrow;  // limbs are < 2^30: bit 31 of t is its sign
        borrow = t >> 31;
        d.v[i] = t & Fu<U>::MASK;
    }
    Fu<U> r;
ZK_UNROLL
    for (int i = 0; i < L; ++i) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}

// any value the contract allows -> canonical Montgomery representative in [0, p):
// a * (R mod p) * R^-1 = a, and the product is < 2p
template <class U>
ZK_HD Fu<U> fu_canon(const Fu<U> &a) {
#ifdef ZK_NOINLINE_MUL
    return fu_cond_sub_p(fu_mul_call(a, Fu<U>::one()));
#else
    return fu_cond_sub_p(fu_mul(a, Fu<U>::one()));
#endif
}

// ---- conversions between the boundary form (saturated canonical u32 limbs) and Fu ----------------------
// split NL 32-bit limbs into L B-bit limbs (value unchanged)
template <class U>
ZK_HD Fu<U> fu_unpack(const uint32_t *sat) {
    constexpr int L = U::L, B = U::B, NL = U::NL;
    Fu<U> r;
ZK_UNROLL
    for (int i = 0; i < L; ++i) {
        const int bit = i * B, w = bit >> 5, off = bit & 31;
        uint64_t x = w < NL ? sat[w] : 0u;
        if (w + 1 < NL) x |= (uint64_t)sat[w + 1] << 32;
        r.v[i] = (uint32_t)(x >> off) & Fu<U>::MASK;
    }
    return r;
}
// inverse of fu_unpack; the value must be < 2^(32 NL) and limbs normalised
template <class U>
ZK_HD void fu_pack(uint32_t *sat, const Fu<U> &a) {
    constexpr int L = U::L, B = U::B, NL = U::NL;
ZK_UNROLL
    for (int w = 0; w < NL; ++w) {
        const int bit = w * 32, i = bit / B, off = bit - i * B;
        uint64_t x = 0;
        if (i < L) x = (uint64_t)a.v[i] >> off;
        if (i + 1 < L) x |= (uint64_t)a.v[i + 1] << (B - off);
        if (i + 2 < L && 2 * B - off < 32) x |= (uint64_t)a.v[i + 2] << (2 * B - off);
        sat[w] = (uint32_t)x;
    }
}
// canonical integer (saturated limbs) -> Montgomery Fu, value < 2p
template <class U>
ZK_HD Fu<U> fu_from_canonical(const uint32_t *sat) {
    return fu_mul(fu_unpack<U>(sat), Fu<U>::r2());
}
// Montgomery Fu (any allowed value) -> canonical integer in [0, p), saturated limbs
template <class U>
ZK_HD void fu_to_canonical(uint32_t *sat, const Fu<U> &a) {
    Fu<U> o = Fu<U>::zero();
    o.v[0] = 1;
    fu_pack<U>(sat, fu_cond_sub_p(fu_mul(a, o)));
}

// device-buffer form: SL words per element (L limbs + zero padding), 16-byte aligned
template <class U>
ZK_HD Fu<U> fu_load(const uint32_t *p) {
    constexpr int L = U::L, SL = U::SL;
    Fu<U> r;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
ZK_UNROLL
    for (int i = 0; i < SL / 4; ++i) {
        uint4 t = q[i];
        if (4 * i + 0 < L) r.v[4 * i + 0] = t.x;
        if (4 * i + 1 < L) r.v[4 * i + 1] = t.y;
        if (4 * i + 2 < L) r.v[4 * i + 2] = t.z;
        if (4 * i + 3 < L) r.v[4 * i + 3] = t.w;
    }
    return r;
}
template <class U>
ZK_HD void fu_store(uint32_t *p, const Fu<U> &a) {
    constexpr int L = U::L, SL = U::SL;
    uint4 *q = reinterpret_cast<uint4 *>(p);
ZK_UNROLL
    for (int i = 0; i < SL / 4; ++i) {
        uint4 t;
        t.x = 4 * i + 0 < L ? a.v[4 * i + 0] : 0u;
        t.y = 4 * i + 1 < L ? a.v[4 * i + 1] : 0u;
        t.z = 4 * i + 2 < L ? a.v[4 * i + 2] : 0u;
        t.w = 4 * i + 3 < L ? a.v[4 * i + 3] : 0u;
        q[i] = t;
    }
}

// ---- quadratic extension over Fu:  Fu2 = Fu[u]/(u^2 + 1) ----------------------------------------------
template <class U>
struct Fu2 {
    typedef U params;
    Fu<U> c0, c1;
    ZK_HD static Fu2 zero() { return {Fu<U>::zero(), Fu<U>::zero()}; }
    ZK_HD static Fu2 one() { return {Fu<U>::one(), Fu<U>::zero()}; }
    ZK_HD bool limbs_zero() const { return c0.limbs_zero() && c1.limbs_zero(); }
};

// =========================================================================================================
// Generic field interface used by curve.hpp / msm.hip (FieldOps<F>): the same group-law source serves the
// saturated reference types (Fp, Fp2: exact reduction after every op, K ignored) and the lazy compute types
// (Fu, Fu2).  Bounds, in units of p, for the lazy types:
//   mul / sqr outputs  < MULB p        (Fu: 2; Fu2: 2 + 8 = 10, from c1 = s - (v0 + v1))
//   sub<K>(a, b)       needs b <= (K - 1) p, gives a + K p
// K1 / K2 / K3 are the spread levels a group-law formula needs when its subtrahend is, respectively,
// a sum of up to three products, a K1-difference, a K2-difference.
// =========================================================================================================
template <class F>
struct FieldOps;

template <class P>
struct FieldOps<Fp<P>> {
    typedef Fp<P> F;
    static constexpr int K1 = 2, K2 = 2, K3 = 2;
    static constexpr int WORDS = P::NL;  // device-buffer words per element
    ZK_HD static F mul(const F &a, const F &b) { return a * b; }
    ZK_HD static F sqr(const F &a) { return a * a; }
    ZK_HD static F add(const F &a, const F &b) { return a + b; }
    template <int K>
    ZK_HD static F sub(const F &a, const F &b) { return a - b; }
    template <int K>
    ZK_HD static F mul_sub(const F &a, const F &b, const F &c, const F &d) { return mul(a, b) - mul(c, d); }
    ZK_HD static bool is_zero(const F &a) { return a.is_zero(); }           // exact
    ZK_HD static bool is_zero_product(const F &a) { return a.is_zero(); }   // a is a mul/sqr output
    ZK_HD static bool is_exact_zero(const F &a) { return a.is_zero(); }     // representation of the constant 0
    ZK_HD static F load(const uint32_t *p) { return fp_load<P>(p); }
    ZK_HD static void store(uint32_t *p, const F &a) { fp_store<P>(p, a); }
    ZK_HD static F from_canonical(const uint32_t *sat) { return fp_to_mont(fp_load<P>(sat)); }
    ZK_HD static void to_canonical(uint32_t *sat, const F &a) { fp_store<P>(sat, fp_from_mont(a)); }
    static constexpr int CANON_WORDS = P::NL;
    ZK_HD static F inv(const F &a) { return fp_inv(a); }
};

template <class P>
struct FieldOps<Fp2<P>> {
    typedef Fp2<P> F;
    static constexpr int K1 = 2, K2 = 2, K3 = 2;
    static constexpr int WORDS = 2 * P::NL;
    ZK_HD static F mul(const F &a, const F &b) { return a * b; }
    ZK_HD static F sqr(const F &a) { return fp_sqr(a); }
    ZK_HD static F add(const F &a, const F &b) { return a + b; }
    template <int K>
    ZK_HD static F sub(const F &a, const F &b) { return a - b; }
    template <int K>
    ZK_HD static F mul_sub(const F &a, const F &b, const F &c, const F &d) { return mul(a, b) - mul(c, d); }
    ZK_HD static bool is_zero(const F &a) { return a.is_zero(); }
    ZK_HD static bool is_zero_product(const F &a) { return a.is_zero(); }
    ZK_HD static bool is_exact_zero(const F &a) { return a.is_zero(); }
    ZK_HD static F load(const uint32_t *p) { return {fp_load<P>(p), fp_load<P>(p + P::NL)}; }
    ZK_HD static void store(uint32_t *p, const F &a) {
        fp_store<P>(p, a.c0);
        fp_store<P>(p + P::NL, a.c1);
    }
    ZK_HD static F from_canonical(const uint32_t *sat) { return {fp_to_mont(fp_load<P>(sat)), fp_to_mont(fp_load<P>(sat + P::NL))}; }
    ZK_HD static void to_canonical(uint32_t *sat, const F &a) {
        fp_store<P>(sat, fp_from_mont(a.c0));
        fp_store<P>(sat + P::NL, fp_from_mont(a.c1));
    }
    static constexpr int CANON_WORDS = 2 * P::NL;
    ZK_HD static F inv(const F &a) { return fp_inv(a); }
};

template <class U>
ZK_HD Fu<U> fu_mul_sel(const Fu<U> &a, const Fu<U> &b) {
#ifdef ZK_NOINLINE_MUL
    return fu_mul_call(a, b);
#else
    return fu_mul(a, b);
#endif
}

// products inside the quadratic extension: out of line also when only ZK_NOINLINE_MUL2 is set (the G2 group
// law holds 30+ base-field products per operation)
template <class U>
ZK_HD Fu<U> fu_mul_sel2(const Fu<U> &a, const Fu<U> &b) {
#if defined(ZK_NOINLINE_MUL) || defined(ZK_NOINLINE_MUL2)
    return fu_mul_call(a, b);
#else
    return fu_mul(a, b);
#endif
}

template <class U>
ZK_HD Fu<U> fu_mul2_sel(const Fu<U> &a, const Fu<U> &b, const Fu<U> &c, const Fu<U> &d) {
#ifdef ZK_NOINLINE_MUL
    return fu_mul2_call(a, b, c, d);
#else
    return fu_mul2(a, b, c, d);
#endif
}
// a^(p-2): serial, never on a per-element path
template <class U>
ZK_HD Fu<U> fu_inv(const Fu<U> &a) {
    constexpr int NL = U::NL;
    uint32_t e[NL];
    uint64_t br = 2;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) {
        uint64_t t = (uint64_t)U::sat::mod(i) - br;
        e[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    Fu<U> r = Fu<U>::one();
    for (int i = NL * 32 - 1; i >= 0; --i) {
        r = fu_mul_call(r, r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = fu_mul_call(r, a);
    }
    return r;
}

template <class U>
struct FieldOps<Fu<U>> {
    typedef Fu<U> F;
    static constexpr int K1 = 8, K2 = 16, K3 = 32;
    static constexpr int WORDS = U::SL;
    ZK_HD static F mul(const F &a, const F &b) { return fu_mul_sel(a, b); }
    ZK_HD static F sqr(const F &a) {
#ifdef ZK_NOINLINE_MUL
        return fu_mul_call(a, a);
#else
        return fu_sqr(a);
#endif
    }
    ZK_HD static F add(const F &a, const F &b) { return fu_add(a, b); }
    template <int K>
    ZK_HD static F sub(const F &a, const F &b) { return fu_sub<K>(a, b); }
    // a b - c d with c <= (K - 1) p: one reduction for both products, result < 2p
    template <int K>
    ZK_HD static F mul_sub(const F &a, const F &b, const F &c, const F &d) {
        return fu_mul2_sel(a, b, fu_sub<K>(F::zero(), c), d);
    }
    ZK_HD static bool is_zero(const F &a) { return fu_canon(a).limbs_zero(); }
    ZK_HD static bool is_zero_product(const F &a) { return fu_is_zero_lt2p(a); }
    ZK_HD static bool is_exact_zero(const F &a) { return a.limbs_zero(); }
    ZK_HD static F load(const uint32_t *p) { return fu_load<U>(p); }
    ZK_HD static void store(uint32_t *p, const F &a) { fu_store<U>(p, a); }
    ZK_HD static F from_canonical(const uint32_t *sat) { return fu_from_canonical<U>(sat); }
    ZK_HD static void to_canonical(uint32_t *sat, const F &a) { fu_to_canonical<U>(sat, a); }
    static constexpr int CANON_WORDS = U::NL;
    ZK_HD static F inv(const F &a) { return fu_inv(a); }
};

template <class U>
struct FieldOps<Fu2<U>> {
    typedef Fu2<U> F;
    typedef FieldOps<Fu<U>> B;
    // products are < 10p per component, so the levels sit a factor ~5 above the base field's
    static constexpr int K1 = 32, K2 = 64, K3 = 128;
    static constexpr int WORDS = 2 * U::SL;
    // Karatsuba: v0, v1, s < 2p; c0 < 6p, c1 < 10p.  Measured and rejected for the G2 bucket kernel (14.2 ms): schoolbook
    // over fu_mul2 (four products, two reductions: the same 6 L^2 mads; 16.0 ms) and Karatsuba with lazy reduction
    // (three products formed column-wise feeding two reductions, 5 L^2 mads; 15.9 ms) -- both need 56 argument
    // registers per out-of-line call, 24 of which the calling convention passes through scratch, and inlining the
    // products overflows the instruction cache.
    ZK_HD static F mul(const F &a, const F &b) {
        Fu<U> v0 = fu_mul_sel2(a.c0, b.c0), v1 = fu_mul_sel2(a.c1, b.c1);
        Fu<U> s = fu_mul_sel2(fu_add(a.c0, a.c1), fu_add(b.c0, b.c1));
        return {fu_sub<4>(v0, v1), fu_sub<8>(s, fu_add(v0, v1))};
    }
    ZK_HD static F sqr(const F &a) {  // (c0 + c1)(c0 - c1), 2 c0 c1; operands up to ~150p stay far below sqrt(R p)
        Fu<U> t = fu_mul_sel2(fu_add(a.c0, a.c1), fu_sub<128>(a.c0, a.c1));
        Fu<U> m = fu_mul_sel2(a.c0, a.c1);
        return {t, fu_add(m, m)};
    }
    ZK_HD static F add(const F &a, const F &b) { return {fu_add(a.c0, b.c0), fu_add(a.c1, b.c1)}; }
    template <int K>
    ZK_HD static F sub(const F &a, const F &b) { return {fu_sub<K>(a.c0, b.c0), fu_sub<K>(a.c1, b.c1)}; }
    template <int K>
    ZK_HD static F mul_sub(const F &a, const F &b, const F &c, const F &d) { return sub<K1>(mul(a, b), mul(c, d)); }
    ZK_HD static bool is_zero(const F &a) { return fu_canon(a.c0).limbs_zero() && fu_canon(a.c1).limbs_zero(); }
    ZK_HD static bool is_zero_product(const F &a) { return is_zero(a); }
    ZK_HD static bool is_exact_zero(const F &a) { return a.limbs_zero(); }
    ZK_HD static F load(const uint32_t *p) { return {fu_load<U>(p), fu_load<U>(p + U::SL)}; }
    ZK_HD static void store(uint32_t *p, const F &a) {
        fu_store<U>(p, a.c0);
        fu_store<U>(p + U::SL, a.c1);
    }
    ZK_HD static F from_canonical(const uint32_t *sat) { return {fu_from_canonical<U>(sat), fu_from_canonical<U>(sat + U::NL)}; }
    ZK_HD static void to_canonical(uint32_t *sat, const F &a) {
        fu_to_canonical<U>(sat, a.c0);
        fu_to_canonical<U>(sat + U::NL, a.c1);
    }
    static constexpr int CANON_WORDS = 2 * U::NL;
    ZK_HD static F inv(const F &a) {
        Fu<U> n = fu_inv(fu_add(fu_mul_sel2(a.c0, a.c0), fu_mul_sel2(a.c1, a.c1)));
        return {fu_mul_sel2(a.c0, n), fu_sub<4>(Fu<U>::zero(), fu_mul_sel2(a.c1, n))};
    }
};

typedef Fu<BlsFqU> bls_fqu;
typedef Fu<BnFqU> bn_fqu;
typedef Fu<BlsFrU> bls_fru;
typedef Fu<BnFrU> bn_fru;
typedef Fu2<BlsFqU> bls_fqu2;
typedef Fu2<BnFqU> bn_fqu2;

}  // namespace zkhip
