// Proving-key points straight from their wire form: BLS12-381 compressed encodings are decoded
// ON THE DEVICE into the resident base tables, so that loading a key costs one H2D of the byte blob instead of a host
// decompression (a square root in Fq / Fq2 per point) plus the conversion to limbs.
//
// Format: what the reference's serializers emit per point -- one Fq (48 B) for G1, two (96 B, x.c1 then x.c0) for G2,
// big-endian, three flag bits in byte 0: 0x80 compressed, 0x40 infinity, 0x20 y is the lexicographically larger root
// (g16/marshalling.hpp:111-112, 178-201; pinned by the literal vectors of
// test/systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark_aggregation_conformity.cpp:932-1010).
// y is recovered from y^2 = x^3 + 4 (G1) or x^3 + 4 (1 + u) (G2); p = 3 mod 4, so a square root in Fq is one
// exponentiation by (p + 1) / 4, and a square root in Fq2 = Fq[u]/(u^2 + 1) takes the "complex method" (two or three
// Fq square roots and one inversion).  No subgroup check (the reference's deserializers do not make one either).
#include "ctx.hpp"
#include "curve.hpp"

using namespace zkhip;

namespace {

// a^((p + 1) / 4)
template <class U>
ZK_D Fu<U> fu_sqrt_candidate(const Fu<U> &a) {
    constexpr int NL = U::NL;
    uint32_t e[NL];
    uint64_t carry = 1;
#pragma unroll
    for (int i = 0; i < NL; ++i) {  // p + 1
        uint64_t t = (uint64_t)U::sat::mod(i) + carry;
        e[i] = (uint32_t)t;
        carry = t >> 32;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) e[i] = (e[i] >> 2) | (i + 1 < NL ? e[i + 1] << 30 : 0u);
    Fu<U> r = Fu<U>::one();
    for (int i = NL * 32 - 1; i >= 0; --i) {
        r = fu_mul_call(r, r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = fu_mul_call(r, a);
    }
    return r;
}

template <class U>
ZK_D bool fu_equal(const Fu<U> &a, const Fu<U> &b) {
    return fu_canon(fu_sub<4>(a, b)).limbs_zero();  // operands below 2p
}

// root of a (value < 2p), if it has one
template <class U>
ZK_D bool fu_sqrt(const Fu<U> &a, Fu<U> &r) {
    r = fu_sqrt_candidate(a);
    return fu_equal(fu_mul_call(r, r), a);
}

// canonical value of y (Montgomery in) above (p - 1) / 2 ?
template <class U>
ZK_D bool fu_is_larger_half(const Fu<U> &y) {
    uint32_t c[U::NL], n[U::NL];
    fu_to_canonical<U>(c, y);
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < U::NL; ++i) {  // n = p - y
        uint64_t d = (uint64_t)U::sat::mod(i) - c[i] - borrow;
        n[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    bool larger = false, decided = false;
#pragma unroll
    for (int i = U::NL - 1; i >= 0; --i)
        if (!decided && c[i] != n[i]) {
            larger = c[i] > n[i];
            decided = true;
        }
    return larger;
}
template <class U>
ZK_D bool fu_is_zero_mod(const Fu<U> &a) { return fu_canon(a).limbs_zero(); }

// 48 big-endian bytes (flag bits already cleared) -> canonical little-endian u32 limbs; false if the value is >= p
template <class U>
ZK_D bool be48_to_canonical(const uint8_t *b, uint32_t *w) {
#pragma unroll
    for (int k = 0; k < U::NL; ++k) {
        const int hi = 4 * U::NL - 1 - 4 * k;  // index of the least significant byte of limb k
        w[k] = (uint32_t)b[hi] | ((uint32_t)b[hi - 1] << 8) | ((uint32_t)b[hi - 2] << 16) | ((uint32_t)b[hi - 3] << 24);
    }
    bool less = false, decided = false;
#pragma unroll
    for (int i = U::NL - 1; i >= 0; --i)
        if (!decided && w[i] != U::sat::mod(i)) {
            less = w[i] < U::sat::mod(i);
            decided = true;
        }
    return less;
}

template <class U>
ZK_D Fu<U> fu_small(uint32_t v) {
    uint32_t w[U::NL] = {0};
    w[0] = v;
    return fu_from_canonical<U>(w);
}

// one lane per point; err[0] counts rejected encodings
template <class U>
__global__ __launch_bounds__(64) void wire_decompress_g1(const uint8_t *__restrict__ octets, uint32_t n, uint32_t *__restrict__ pts,
                                                         uint32_t *__restrict__ err) {
    typedef Fu<U> F;
    constexpr int NB = 4 * U::NL, NLW = FieldOps<F>::WORDS;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t b[NB];
    for (int k = 0; k < NB; ++k) b[k] = octets[(size_t)i * NB + k];
    const uint32_t flags = b[0] >> 5;
    b[0] &= 0x1F;
    Affine<F> out = Affine<F>::infinity();
    bool bad = !(flags & 4);
    if (!bad && (flags & 2)) {  // infinity: everything else must be zero
        bad = (flags & 1) != 0;
        for (int k = 0; k < NB; ++k) bad = bad || b[k] != 0;
    } else if (!bad) {
        uint32_t w[U::NL];
        bad = !be48_to_canonical<U>(b, w);
        if (!bad) {
            F x = fu_from_canonical<U>(w);
            F y2 = fu_add(fu_mul_call(fu_mul_call(x, x), x), fu_small<U>(4)), y;
            bad = !fu_sqrt(fu_cond_sub_p(fu_mul_call(y2, F::one())), y);
            if (!bad) {
                if (fu_is_larger_half(y) != ((flags & 1) != 0)) y = fu_sub<4>(F::zero(), y);
                out = {fu_cond_sub_p(fu_mul_call(x, F::one())), fu_cond_sub_p(fu_mul_call(y, F::one()))};
            }
        }
    }
    if (bad) atomicAdd(err, 1u);
    affine_store<F>(pts + (size_t)i * (2 * NLW), out);
}

template <class U>
__global__ __launch_bounds__(64) void wire_decompress_g2(const uint8_t *__restrict__ octets, uint32_t n, uint32_t *__restrict__ pts,
                                                         uint32_t *__restrict__ err) {
    typedef Fu<U> B;
    typedef Fu2<U> F;
    typedef FieldOps<F> O;
    constexpr int NB = 4 * U::NL, NLW = O::WORDS;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t b1[NB], b0[NB];  // x.c1 comes first on the wire
    for (int k = 0; k < NB; ++k) {
        b1[k] = octets[(size_t)i * 2 * NB + k];
        b0[k] = octets[(size_t)i * 2 * NB + NB + k];
    }
    const uint32_t flags = b1[0] >> 5;
    b1[0] &= 0x1F;
    Affine<F> out = Affine<F>::infinity();
    bool bad = !(flags & 4);
    if (!bad && (flags & 2)) {
        bad = (flags & 1) != 0;
        for (int k = 0; k < NB; ++k) bad = bad || b1[k] != 0 || b0[k] != 0;
    } else if (!bad) {
        uint32_t w0[U::NL], w1[U::NL];
        bad = !be48_to_canonical<U>(b0, w0);
        bad = !be48_to_canonical<U>(b1, w1) || bad;
        if (!bad) {
            F x = {fu_from_canonical<U>(w0), fu_from_canonical<U>(w1)};
            const B four = fu_small<U>(4), one = B::one();
            F a = O::mul(O::sqr(x), x);  // components < 10p
            B a0 = fu_cond_sub_p(fu_mul_call(fu_add(a.c0, four), one)), a1 = fu_cond_sub_p(fu_mul_call(fu_add(a.c1, four), one));  // canonical reps
            B y0 = B::zero(), y1 = B::zero();
            if (a1.limbs_zero()) {
                if (!fu_sqrt(a0, y0)) {  // a0 is not a square: sqrt(a) = sqrt(-a0) u
                    y0 = B::zero();
                    bad = !fu_sqrt(fu_cond_sub_p(fu_sub<2>(B::zero(), a0)), y1);
                }
            } else {
                B s, t, inv2 = fu_inv(fu_add(one, one));
                bad = !fu_sqrt(fu_cond_sub_p(fu_mul_call(fu_add(fu_mul_call(a0, a0), fu_mul_call(a1, a1)), one)), s);  // sqrt of the norm
                if (!bad) {
                    t = fu_cond_sub_p(fu_mul_call(fu_add(a0, s), inv2));
                    if (!fu_sqrt(t, y0)) {
                        t = fu_cond_sub_p(fu_mul_call(fu_sub<4>(a0, s), inv2));
                        bad = !fu_sqrt(t, y0);
                    }
                    bad = bad || fu_is_zero_mod(y0);
                    if (!bad) y1 = fu_mul_call(a1, fu_inv(fu_add(y0, y0)));
                }
            }
            if (!bad) {  // (y0 + y1 u)^2 == a ?
                F y = {fu_cond_sub_p(fu_mul_call(y0, one)), fu_cond_sub_p(fu_mul_call(y1, one))};
                F yy = O::sqr(y);
                bad = !fu_equal(fu_cond_sub_p(fu_mul_call(yy.c0, one)), a0) || !fu_equal(fu_cond_sub_p(fu_mul_call(yy.c1, one)), a1);
                if (!bad) {
                    const bool larger = fu_is_zero_mod(y.c1) ? fu_is_larger_half(y.c0) : fu_is_larger_half(y.c1);
                    if (larger != ((flags & 1) != 0)) y = {fu_cond_sub_p(fu_sub<2>(B::zero(), y.c0)), fu_cond_sub_p(fu_sub<2>(B::zero(), y.c1))};
                    out = {{fu_cond_sub_p(fu_mul_call(x.c0, one)), fu_cond_sub_p(fu_mul_call(x.c1, one))}, y};
                }
            }
        }
    }
    if (bad) atomicAdd(err, 1u);
    affine_store<F>(pts + (size_t)i * (2 * NLW), out);
}

}  // namespace

// table 0 of `b` <- the n points encoded at d_octets (device); *d_err (device, zeroed by the caller) counts rejected ones
int zk_bases_decompress(zkhip_ctx *ctx, zkhip_bases *b, const uint8_t *d_octets, uint32_t *d_err) {
    if (b->curve != CURVE_BLS12_381) return ZKHIP_ERR_INVALID;  // only this curve's wire format is pinned by the reference's vectors
    if (b->n == 0) return 0;
    dim3 grid((unsigned)((b->n + 63) / 64)), block(64);
    if (b->group == GROUP_G1) ZK_LAUNCH(ctx, "wire_decompress", wire_decompress_g1<BlsFqU>, grid, block, 0, d_octets, (uint32_t)b->n, b->d, d_err);
    else ZK_LAUNCH(ctx, "wire_decompress", wire_decompress_g2<BlsFqU>, grid, block, 0, d_octets, (uint32_t)b->n, b->d, d_err);
    return 0;
}
