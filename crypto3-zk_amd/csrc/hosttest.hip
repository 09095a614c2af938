// TEST SHIM (never loaded by the product path): runs the __host__ __device__ field / curve / recoding
// code of fp.hpp, fu.hpp, curve.hpp and msm_recode.hpp on the CPU so the no-GPU test-suite can compare the
// exact device arithmetic against the oracle.  Built with --offload-host-only into libzkhip_hosttest.so.
#include <cstring>
#include <vector>

#define ZK_NOINLINE_MUL 1  // keeps this shim's build time short; fu_sqr is reached through op 9
#include "curve.hpp"
#include "fu_safegcd.hpp"
#include "msm_recode.hpp"

using namespace zkhip;

namespace {

// op: 0 mul, 1 add, 2 sub<K1>, 3 inv(a), 4 sqr(a), 5 neg(a) = sub<K1>(0, a), 6 dbl(a), 7 sub<K2>, 9 fu_sqr(a + b),
//     10 mul_sub<K2>(a, a + b, sub<K1>(0, b), b) = a (a + b) + b^2  (one shared reduction for the lazy base field),
//     8 bound stress: mul(sub<K2>(mul(a,b), X), sub<K2>(sqr(b), X)),  X = sub<K1>(sqr(a), ab + 2 b^2)
//       -- the deepest lazy chain of the group law, with every operand at its contract bound
// the dedicated Montgomery square the kernels inline (FieldOps::sqr routes to the out-of-line product in this build)
template <class F>
F sqr_direct(const F &x) { return FieldOps<F>::sqr(x); }
template <class U>
Fu<U> sqr_direct(const Fu<U> &x) { return fu_sqr(x); }

// op 11: the safegcd inverse the grand products take once per call (fu_safegcd.hpp); the saturated reference types have none of their own
template <class F>
F inv_gcd(const F &x) { return FieldOps<F>::inv(x); }
template <class U>
Fu<U> inv_gcd(const Fu<U> &x) { return fu_inv_gcd(fu_canon(x)); }

template <class F>
int field_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
    typedef FieldOps<F> O;
    F x = O::from_canonical(a), y = b ? O::from_canonical(b) : F::zero(), r;
    switch (op) {
        case 0: r = O::mul(x, y); break;
        case 1: r = O::add(x, y); break;
        case 2: r = O::template sub<O::K1>(x, y); break;
        case 3: r = O::inv(x); break;
        case 4: r = O::sqr(x); break;
        case 5: r = O::template sub<O::K1>(F::zero(), x); break;
        case 6: r = O::add(x, x); break;
        case 7: r = O::template sub<O::K2>(x, y); break;
        case 9: r = sqr_direct(O::add(x, y)); break;  // (a + b)^2 through fu_sqr, operand not reduced
        case 10: r = O::template mul_sub<O::K2>(x, O::add(x, y), O::template sub<O::K1>(F::zero(), y), y); break;
        case 11: r = inv_gcd(x); break;
        case 8: {
            F ab = O::mul(x, y), bb = O::sqr(y);
            F X = O::template sub<O::K1>(O::sqr(x), O::add(ab, O::add(bb, bb)));
            r = O::mul(O::template sub<O::K2>(ab, X), O::template sub<O::K2>(bb, X));
            break;
        }
        default: return -1;
    }
    O::to_canonical(out, r);
    return 0;
}

template <class F>
Affine<F> load_aff(const uint32_t *p, int inf) {
    typedef FieldOps<F> O;
    if (inf) return Affine<F>::infinity();
    return {O::from_canonical(p), O::from_canonical(p + O::CANON_WORDS)};
}
template <class F>
void store_aff(uint32_t *p, uint8_t *inf, const XYZZ<F> &a) {
    typedef FieldOps<F> O;
    Affine<F> r = xyzz_to_affine(a);
    *inf = a.is_inf() ? 1 : 0;
    O::to_canonical(p, r.x);
    O::to_canonical(p + O::CANON_WORDS, r.y);
}

// sum_i (+/-) pts[i] accumulated with xyzz_madd in order; result affine canonical.
// mode: 0 = madd chain; 1 = xyzz_add of the chains over the two halves; 2 = chain then xyzz_mul_small(acc, k);
//       3 = chain, xyzz_to_jacobian -> canonical Jacobian out (3 coords); 4 = like 0 but every partial sum goes
//       through the device-buffer store/load round trip (xyzz_store / xyzz_load, affine_store / affine_load)
template <class F>
int point_chain(const uint32_t *pts, const uint8_t *inf, const uint8_t *neg, size_t n, int mode, uint32_t k, uint32_t *out, uint8_t *out_inf) {
    typedef FieldOps<F> O;
    constexpr int CW = O::CANON_WORDS;
    size_t split = mode == 1 ? n / 2 : n;
    XYZZ<F> acc = XYZZ<F>::infinity(), acc2 = XYZZ<F>::infinity();
    std::vector<uint32_t> buf(4 * O::WORDS + 2 * O::WORDS + 8);
    uint32_t *b16 = (uint32_t *)(((uintptr_t)buf.data() + 15) & ~(uintptr_t)15);
    for (size_t i = 0; i < split; ++i) {
        Affine<F> p = load_aff<F>(pts + i * 2 * CW, inf ? inf[i] : 0);
        if (mode == 4) {
            affine_store<F>(b16, p);
            p = affine_load<F>(b16);
        }
        acc = xyzz_madd(acc, p, neg ? neg[i] != 0 : false);
        if (mode == 4) {
            xyzz_store<F>(b16, acc);
            acc = xyzz_load<F>(b16);
        }
    }
    for (size_t i = split; i < n; ++i) acc2 = xyzz_madd(acc2, load_aff<F>(pts + i * 2 * CW, inf ? inf[i] : 0), neg ? neg[i] != 0 : false);
    if (mode == 1) acc = xyzz_add(acc, acc2);
    if (mode == 2) acc = xyzz_mul_small(acc, k);
    if (mode == 3) {
        Jacobian<F> j = xyzz_to_jacobian(acc);
        O::to_canonical(out, j.X);
        O::to_canonical(out + CW, j.Y);
        O::to_canonical(out + 2 * CW, j.Z);
        *out_inf = acc.is_inf() ? 1 : 0;
        return 0;
    }
    store_aff<F>(out, out_inf, acc);
    return 0;
}

}  // namespace

#define FIELD_SWITCH(field, ...)                             \
    switch (field) {                                         \
        case 0: { typedef bls_fq F; __VA_ARGS__; } break;    \
        case 1: { typedef bls_fr F; __VA_ARGS__; } break;    \
        case 2: { typedef bn_fq F; __VA_ARGS__; } break;     \
        case 3: { typedef bn_fr F; __VA_ARGS__; } break;     \
        case 4: { typedef bls_fq2 F; __VA_ARGS__; } break;   \
        case 5: { typedef bn_fq2 F; __VA_ARGS__; } break;    \
        case 6: { typedef bls_fqu F; __VA_ARGS__; } break;   \
        case 7: { typedef bn_fqu F; __VA_ARGS__; } break;    \
        case 8: { typedef bls_fru F; __VA_ARGS__; } break;   \
        case 9: { typedef bn_fru F; __VA_ARGS__; } break;    \
        case 10: { typedef bls_fqu2 F; __VA_ARGS__; } break; \
        case 11: { typedef bn_fqu2 F; __VA_ARGS__; } break;  \
        default: return -1;                                  \
    }

extern "C" {

// field: 0 BLS Fq, 1 BLS Fr, 2 BN Fq, 3 BN Fr, 4 BLS Fq2, 5 BN Fq2 (saturated reference types);
//        6 BLS Fq, 7 BN Fq, 8 BLS Fr, 9 BN Fr, 10 BLS Fq2, 11 BN Fq2 (lazy 29-bit-limb compute types).
// canonical u32 limbs in and out
int zkt_field_op(int field, int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
    FIELD_SWITCH(field, return field_op<F>(op, a, b, out));
    return -1;
}

// coordinate field id as above (0/2/4/5 saturated, 6/7/10/11 lazy)
int zkt_point_chain(int field, const uint32_t *pts, const uint8_t *inf, const uint8_t *neg, size_t n, int mode, uint32_t k,
                    uint32_t *out, uint8_t *out_inf) {
    if (field == 1 || field == 3 || field == 8 || field == 9) return -1;
    FIELD_SWITCH(field, return point_chain<F>(pts, inf, neg, n, mode, k, out, out_inf));
    return -1;
}

// digits[w] for one scalar: value = sum_w digit_w * 2^off(w), digit as signed int32 (0 when none); returns the carry
// out of the top window
static int recode_windows(const uint32_t *scalar, MsmWindows win, int32_t *digits) {
    uint32_t carry = 0;
    for (int w = 0; w < win.W; ++w) {
        uint32_t d = msm_recode(scalar, win.off(w), win.width(w), carry);
        if (d == DIG_NONE) digits[w] = 0;
        else {
            int32_t mag = (int32_t)(d & 0x7FFFFFFFu) + 1;
            digits[w] = (d >> 31) ? -mag : mag;
        }
    }
    return (int)carry;
}

// W uniform windows of c bits (off(w) = c w)
int zkt_recode(const uint32_t *scalar, int c, int W, int32_t *digits) { return recode_windows(scalar, msm_make_windows(c * W, W), digits); }

// the recoding msm_digits_only performs: fold to |s| <= (r - 1) / 2, then signed digits over the balanced windows
// MsmWindows{bitlen(r), ceil(bitlen(r) / c)}: value = sum_w digit_w * 2^floor(w * bitlen(r) / W).
// Returns the window count, or -1 if a carry left the top window (must not happen).  curve: 0 BLS12-381, 1 BN254.
int zkt_recode_folded(int curve, const uint32_t *scalar, int c, int32_t *digits) {
    uint32_t s[8];
    const bool flip = curve == 0 ? msm_fold_scalar<BlsFr>(scalar, s) : msm_fold_scalar<BnFr>(scalar, s);
    const int tb = curve == 0 ? 255 : 254, W = msm_windows(tb, c);
    if (recode_windows(s, msm_make_windows(tb, W), digits) != 0) return -1;
    if (flip)
        for (int w = 0; w < W; ++w) digits[w] = -digits[w];
    return W;
}

}  // extern "C"
