// TEST SHIM (never loaded by the product path): runs the __host__ __device__ field / curve / recoding
// code of fp.hpp, curve.hpp and msm_recode.hpp on the CPU so the no-GPU test-suite can compare the exact
// device arithmetic against the oracle.  Built with --offload-host-only into libzkhip_hosttest.so.
#include <cstring>
#include <vector>

#define ZK_NOINLINE_MUL 1
#include "curve.hpp"
#include "msm_recode.hpp"

using namespace zkhip;

namespace {

template <class F>
F load_canon(const uint32_t *p) {
    return fp_to_mont(FieldIO<F>::load(p));
}
template <class F>
void store_canon(uint32_t *p, const F &a) {
    FieldIO<F>::store(p, fp_from_mont(a));
}

// op: 0 mul, 1 add, 2 sub, 3 inv(a), 4 sqr(a), 5 neg(a), 6 dbl(a)
template <class F>
int field_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
    F x = load_canon<F>(a), y = b ? load_canon<F>(b) : F::zero(), r;
    switch (op) {
        case 0: r = x * y; break;
        case 1: r = x + y; break;
        case 2: r = x - y; break;
        case 3: r = fp_inv(x); break;
        case 4: r = fp_sqr(x); break;
        case 5: r = fp_neg(x); break;
        case 6: r = fp_dbl(x); break;
        default: return -1;
    }
    store_canon<F>(out, r);
    return 0;
}

template <class F>
Affine<F> load_aff(const uint32_t *p, int inf) {
    if (inf) return Affine<F>::infinity();
    constexpr int NL = FieldIO<F>::NL;
    return {load_canon<F>(p), load_canon<F>(p + NL)};
}
template <class F>
void store_aff(uint32_t *p, uint8_t *inf, const XYZZ<F> &a) {
    constexpr int NL = FieldIO<F>::NL;
    Affine<F> r = xyzz_to_affine(a);
    *inf = a.is_inf() ? 1 : 0;
    store_canon<F>(p, r.x);
    store_canon<F>(p + NL, r.y);
}

// sum_i (+/-) pts[i] accumulated with xyzz_madd in order, then optionally doubled `dbls` times,
// multiplied by `k` (xyzz_mul_small) and added to itself via xyzz_add; result affine canonical.
// mode: 0 = madd chain; 1 = chain then xyzz_add(acc, acc2) where acc2 = chain over the second half;
//       2 = chain then xyzz_mul_small(acc, k); 3 = chain, via xyzz_to_jacobian -> canonical Jacobian out (3 coords)
template <class F>
int point_chain(const uint32_t *pts, const uint8_t *inf, const uint8_t *neg, size_t n, int mode, uint32_t k, uint32_t *out, uint8_t *out_inf) {
    constexpr int NL = FieldIO<F>::NL;
    size_t split = mode == 1 ? n / 2 : n;
    XYZZ<F> acc = XYZZ<F>::infinity(), acc2 = XYZZ<F>::infinity();
    for (size_t i = 0; i < split; ++i) acc = xyzz_madd(acc, load_aff<F>(pts + i * 2 * NL, inf ? inf[i] : 0), neg ? neg[i] != 0 : false);
    for (size_t i = split; i < n; ++i) acc2 = xyzz_madd(acc2, load_aff<F>(pts + i * 2 * NL, inf ? inf[i] : 0), neg ? neg[i] != 0 : false);
    if (mode == 1) acc = xyzz_add(acc, acc2);
    if (mode == 2) acc = xyzz_mul_small(acc, k);
    if (mode == 3) {
        Jacobian<F> j = xyzz_to_jacobian(acc);
        store_canon<F>(out, j.X);
        store_canon<F>(out + NL, j.Y);
        store_canon<F>(out + 2 * NL, j.Z);
        *out_inf = acc.is_inf() ? 1 : 0;
        return 0;
    }
    store_aff<F>(out, out_inf, acc);
    return 0;
}

}  // namespace

#define FIELD_SWITCH(field, ...)                      \
    switch (field) {                                  \
        case 0: { typedef bls_fq F; __VA_ARGS__; } break;  \
        case 1: { typedef bls_fr F; __VA_ARGS__; } break;  \
        case 2: { typedef bn_fq F; __VA_ARGS__; } break;   \
        case 3: { typedef bn_fr F; __VA_ARGS__; } break;   \
        case 4: { typedef bls_fq2 F; __VA_ARGS__; } break; \
        case 5: { typedef bn_fq2 F; __VA_ARGS__; } break;  \
        default: return -1;                           \
    }

extern "C" {

// field: 0 BLS Fq, 1 BLS Fr, 2 BN Fq, 3 BN Fr, 4 BLS Fq2, 5 BN Fq2; canonical u32 limbs in and out
int zkt_field_op(int field, int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
    FIELD_SWITCH(field, return field_op<F>(op, a, b, out));
    return -1;
}

// curve 0/1, group 1/2
int zkt_point_chain(int curve, int group, const uint32_t *pts, const uint8_t *inf, const uint8_t *neg, size_t n, int mode, uint32_t k,
                    uint32_t *out, uint8_t *out_inf) {
    int field = curve == 0 ? (group == 1 ? 0 : 4) : (group == 1 ? 2 : 5);
    FIELD_SWITCH(field, return point_chain<F>(pts, inf, neg, n, mode, k, out, out_inf));
    return -1;
}

// digits[w] for one scalar: value = sum_w digit_w * 2^(c w), digit as signed int32 (0 when none)
int zkt_recode(const uint32_t *scalar, int c, int W, int32_t *digits) {
    uint32_t carry = 0;
    for (int w = 0; w < W; ++w) {
        uint32_t d = msm_recode(scalar, w, c, carry);
        if (d == DIG_NONE) digits[w] = 0;
        else {
            int32_t mag = (int32_t)(d & 0x7FFFFFFFu) + 1;
            digits[w] = (d >> 31) ? -mag : mag;
        }
    }
    return (int)carry;
}

}  // extern "C"
