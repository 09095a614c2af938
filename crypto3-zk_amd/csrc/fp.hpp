// Montgomery prime-field arithmetic on 32-bit limbs for the gfx950 VALU.
//
// In HBM every element is stored as 64-bit little-endian limbs (4 x u64 for Fr, 6 x u64 for BLS12-381 Fq);
// a u64 limb is two consecutive u32 limbs, so the same bytes are addressed here as NL x u32.  The VALU is
// 32-bit: products are v_mad_u64_u32 (32x32+64).  Everything is __host__ __device__ so that the CPU test
// suite can check this exact code against the oracle without a GPU (tests/test_host_arith.py).
//
// Replaces (on the device) the field arithmetic the reference reaches through crypto3-algebra
// (`typename FieldType::value_type` operators used throughout zk/snark/reductions/r1cs_to_qap.hpp:245-321).
#pragma once
#include "field_consts.hpp"
#include "zk_defs.hpp"

namespace zkhip {

template <class P>
struct Fp {
    static constexpr int NL = P::NL;
    typedef P params;
    uint32_t v[NL];

    ZK_HD static Fp zero() {
        Fp r;
ZK_UNROLL
        for (int i = 0; i < NL; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fp one() {  // Montgomery form of 1
        Fp r;
ZK_UNROLL
        for (int i = 0; i < NL; ++i) r.v[i] = P::r1(i);
        return r;
    }
    ZK_HD static Fp r2() {
        Fp r;
ZK_UNROLL
        for (int i = 0; i < NL; ++i) r.v[i] = P::r2(i);
        return r;
    }
    ZK_HD bool is_zero() const {
        uint32_t o = 0;
ZK_UNROLL
        for (int i = 0; i < NL; ++i) o |= v[i];
        return o == 0;
    }
    ZK_HD bool operator==(const Fp &b) const {
        uint32_t o = 0;
ZK_UNROLL
        for (int i = 0; i < NL; ++i) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    ZK_HD bool operator!=(const Fp &b) const { return !(*this == b); }
};

// r = a - p if a >= p else a   (a < 2p)
template <class P>
ZK_HD void fp_reduce_once(Fp<P> &a) {
    constexpr int NL = P::NL;
    uint32_t d[NL];
    uint64_t br = 0;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) {
        uint64_t t = (uint64_t)a.v[i] - P::mod(i) - br;
        d[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    bool keep = br != 0;  // a < p
ZK_UNROLL
    for (int i = 0; i < NL; ++i) a.v[i] = keep ? a.v[i] : d[i];
}

template <class P>
ZK_HD Fp<P> operator+(const Fp<P> &a, const Fp<P> &b) {
    constexpr int NL = P::NL;
    Fp<P> r;
    uint64_t c = 0;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // all four moduli leave the top bit of limb NL-1 clear, so a + b < 2p < 2^(32 NL): no carry out
    fp_reduce_once(r);
    return r;
}

template <class P>
ZK_HD Fp<P> operator-(const Fp<P> &a, const Fp<P> &b) {
    constexpr int NL = P::NL;
    Fp<P> r;
    uint64_t br = 0;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
        r.v[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;  // all ones when a < b
    uint64_t c = 0;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) {
        c += (uint64_t)r.v[i] + (P::mod(i) & mask);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class P>
ZK_HD Fp<P> fp_neg(const Fp<P> &a) {
    return Fp<P>::zero() - a;
}
template <class P>
ZK_HD Fp<P> fp_dbl(const Fp<P> &a) {
    return a + a;
}

// Montgomery product a*b*R^-1 mod p, R = 2^(32 NL).  CIOS with the two carry chains merged
// (valid because the top limb of every modulus here is < 2^31).
template <class P>
ZK_HD Fp<P> fp_mul_inline(const Fp<P> &a, const Fp<P> &b) {
    constexpr int NL = P::NL;
    uint32_t t[NL];
ZK_UNROLL
    for (int i = 0; i < NL; ++i) t[i] = 0;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) {
        uint64_t A = (uint64_t)a.v[0] * b.v[i] + t[0];
        uint32_t m = (uint32_t)A * P::INV;
        uint64_t C = (uint64_t)m * P::mod(0) + (uint32_t)A;
        A >>= 32;
        C >>= 32;
ZK_UNROLL
        for (int j = 1; j < NL; ++j) {
            A += (uint64_t)a.v[j] * b.v[i] + t[j];
            C += (uint64_t)m * P::mod(j) + (uint32_t)A;
            t[j - 1] = (uint32_t)C;
            A >>= 32;
            C >>= 32;
        }
        t[NL - 1] = (uint32_t)(A + C);
    }
    Fp<P> r;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) r.v[i] = t[i];
    fp_reduce_once(r);
    return r;
}

// Out-of-line copy: the elliptic-curve kernels call the product instead of inlining it ~10-40 times per
// group operation (one fully unrolled product is ~1.2k instructions; inlined group laws do not fit the
// instruction cache and take hipcc tens of minutes to schedule).  Translation units that want the
// product inlined (the NTT butterflies) leave ZK_NOINLINE_MUL undefined.
template <class P>
ZK_NOINLINE_HD Fp<P> fp_mul_call(const Fp<P> &a, const Fp<P> &b) {
    return fp_mul_inline(a, b);
}

template <class P>
ZK_HD Fp<P> operator*(const Fp<P> &a, const Fp<P> &b) {
#ifdef ZK_NOINLINE_MUL
    return fp_mul_call(a, b);
#else
    return fp_mul_inline(a, b);
#endif
}

template <class P>
ZK_HD Fp<P> fp_sqr(const Fp<P> &a) {
    return a * a;
}

// canonical <-> Montgomery
template <class P>
ZK_HD Fp<P> fp_to_mont(const Fp<P> &c) {
    return c * Fp<P>::r2();
}
template <class P>
ZK_HD Fp<P> fp_from_mont(const Fp<P> &m) {
    Fp<P> o = Fp<P>::zero();
    o.v[0] = 1;
    return m * o;
}

// a^e for a small exponent (e < 2^64); not on any hot path
template <class P>
ZK_HD Fp<P> fp_pow_u64(const Fp<P> &a, uint64_t e) {
    Fp<P> r = Fp<P>::one();
    Fp<P> b = a;
    while (e) {
        if (e & 1) r = r * b;
        b = b * b;
        e >>= 1;
    }
    return r;
}

// a^(p-2) (Fermat inverse); serial, used once per MSM result / table build, never per element
template <class P>
ZK_HD Fp<P> fp_inv(const Fp<P> &a) {
    constexpr int NL = P::NL;
    // exponent e = p - 2, limb-wise with borrow
    uint32_t e[NL];
    uint64_t br = 2;
ZK_UNROLL
    for (int i = 0; i < NL; ++i) {
        uint64_t t = (uint64_t)P::mod(i) - br;
        e[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    Fp<P> r = Fp<P>::one();
    for (int i = NL * 32 - 1; i >= 0; --i) {
        r = r * r;
        if ((e[i >> 5] >> (i & 31)) & 1) r = r * a;
    }
    return r;
}

// 16-byte vector load/store of a field element that is 16-byte aligned in memory
template <class P>
ZK_HD Fp<P> fp_load(const uint32_t *p) {
    constexpr int NL = P::NL;
    static_assert(NL % 4 == 0, "limb count must be a multiple of 4 for dwordx4 access");
    Fp<P> r;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
ZK_UNROLL
    for (int i = 0; i < NL / 4; ++i) {
        uint4 t = q[i];
        r.v[4 * i + 0] = t.x;
        r.v[4 * i + 1] = t.y;
        r.v[4 * i + 2] = t.z;
        r.v[4 * i + 3] = t.w;
    }
    return r;
}
template <class P>
ZK_HD void fp_store(uint32_t *p, const Fp<P> &a) {
    constexpr int NL = P::NL;
    uint4 *q = reinterpret_cast<uint4 *>(p);
ZK_UNROLL
    for (int i = 0; i < NL / 4; ++i) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

// ---------------------------------------------------------------------------------------------
// quadratic extension Fq2 = Fq[u]/(u^2 + 1)  (BLS12-381 and BN254 both use u^2 = -1)
// ---------------------------------------------------------------------------------------------
template <class P>
struct Fp2 {
    typedef P params;
    static constexpr int NL = 2 * P::NL;
    Fp<P> c0, c1;
    ZK_HD static Fp2 zero() { return {Fp<P>::zero(), Fp<P>::zero()}; }
    ZK_HD static Fp2 one() { return {Fp<P>::one(), Fp<P>::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_HD bool operator==(const Fp2 &b) const { return c0 == b.c0 && c1 == b.c1; }
    ZK_HD bool operator!=(const Fp2 &b) const { return !(*this == b); }
};
template <class P>
ZK_HD Fp2<P> operator+(const Fp2<P> &a, const Fp2<P> &b) {
    return {a.c0 + b.c0, a.c1 + b.c1};
}
template <class P>
ZK_HD Fp2<P> operator-(const Fp2<P> &a, const Fp2<P> &b) {
    return {a.c0 - b.c0, a.c1 - b.c1};
}
template <class P>
ZK_HD Fp2<P> fp_neg(const Fp2<P> &a) {
    return {fp_neg(a.c0), fp_neg(a.c1)};
}
template <class P>
ZK_HD Fp2<P> fp_dbl(const Fp2<P> &a) {
    return {fp_dbl(a.c0), fp_dbl(a.c1)};
}
template <class P>
ZK_HD Fp2<P> operator*(const Fp2<P> &a, const Fp2<P> &b) {  // Karatsuba, 3 base-field products
    Fp<P> v0 = a.c0 * b.c0, v1 = a.c1 * b.c1;
    Fp<P> s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {v0 - v1, s - v0 - v1};
}
template <class P>
ZK_HD Fp2<P> fp_sqr(const Fp2<P> &a) {  // complex squaring, 2 base-field products
    Fp<P> t = (a.c0 + a.c1) * (a.c0 - a.c1);
    Fp<P> m = a.c0 * a.c1;
    return {t, fp_dbl(m)};
}
template <class P>
ZK_HD Fp2<P> fp_inv(const Fp2<P> &a) {
    Fp<P> n = fp_inv(fp_sqr(a.c0) + fp_sqr(a.c1));
    return {a.c0 * n, fp_neg(a.c1 * n)};
}
template <class P>
ZK_HD Fp2<P> fp_to_mont(const Fp2<P> &c) {
    return {fp_to_mont(c.c0), fp_to_mont(c.c1)};
}
template <class P>
ZK_HD Fp2<P> fp_from_mont(const Fp2<P> &c) {
    return {fp_from_mont(c.c0), fp_from_mont(c.c1)};
}

// uniform load/store for Fp and Fp2 (Fp2 = c0 limbs then c1 limbs)
template <class F>
struct FieldIO;
template <class P>
struct FieldIO<Fp<P>> {
    static constexpr int NL = P::NL;
    ZK_HD static Fp<P> load(const uint32_t *p) { return fp_load<P>(p); }
    ZK_HD static void store(uint32_t *p, const Fp<P> &a) { fp_store<P>(p, a); }
};
template <class P>
struct FieldIO<Fp2<P>> {
    static constexpr int NL = 2 * P::NL;
    ZK_HD static Fp2<P> load(const uint32_t *p) { return {fp_load<P>(p), fp_load<P>(p + P::NL)}; }
    ZK_HD static void store(uint32_t *p, const Fp2<P> &a) {
        fp_store<P>(p, a.c0);
        fp_store<P>(p + P::NL, a.c1);
    }
};

typedef Fp<BlsFq> bls_fq;
typedef Fp<BlsFr> bls_fr;
typedef Fp<BnFq> bn_fq;
typedef Fp<BnFr> bn_fr;
typedef Fp2<BlsFq> bls_fq2;
typedef Fp2<BnFq> bn_fq2;

}  // namespace zkhip
