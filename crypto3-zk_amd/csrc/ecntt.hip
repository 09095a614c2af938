// Radix-2 (inverse) DFT over GROUP ELEMENTS: out[j] = sum_i omega^(i j) P_i  (inverse: omega^-1 and a final 1/m).
//
// This is evaluation_domain<Fr, G>::evaluate_all_lagrange_polynomials(powers_begin, powers_end) as the powers-of-tau
// result uses it (zk/commitments/detail/polynomial/powers_of_tau/result.hpp:81-94): with P_i = tau^i G the inverse
// transform yields L_j(tau) G for every Lagrange basis polynomial of the domain.  Setup-side
// work: every butterfly multiplies a point by a full-width twiddle (double-and-add, ~255 doublings + ~128 additions),
// m/2 log2 m of them; the butterflies of a stage are independent (one lane each).
//
// Boundary form: canonical Jacobian points (X, Y, Z; Z = 0 for infinity), as zkhip_msm_dev returns them.
#include <algorithm>

#include "ctx.hpp"
#include "curve.hpp"

using namespace zkhip;

namespace {

// consts: [0] = effective root (omega, or omega^-1 for the inverse) canonical; [1] = 1/m canonical
template <class U>
__global__ void ec_ntt_setup(const uint32_t *__restrict__ omega_c, uint32_t log_m, int inverse, uint32_t *__restrict__ consts) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Fu<U> w = fu_from_canonical<U>(omega_c);
    if (inverse) w = fu_inv(w);
    fu_to_canonical<U>(consts, w);
    Fu<U> two = fu_add(Fu<U>::one(), Fu<U>::one()), mm = Fu<U>::one();
    for (uint32_t i = 0; i < log_m; ++i) mm = fu_mul_call(mm, two);
    fu_to_canonical<U>(consts + U::NL, fu_inv(mm));
}

// tw[j] = w^j for j < count (canonical scalars, 8 words each): square-and-multiply over the bits of j
template <class U>
__global__ __launch_bounds__(256) void ec_ntt_twiddles(const uint32_t *__restrict__ consts, uint32_t count, uint32_t *__restrict__ tw) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fu<U> pw = fu_from_canonical<U>(consts), acc = Fu<U>::one();
    for (uint32_t e = j; e; e >>= 1) {
        if (e & 1) acc = fu_mul_call(acc, pw);
        pw = fu_mul_call(pw, pw);
    }
    fu_to_canonical<U>(tw + (size_t)j * U::NL, acc);
}

template <class F>
ZK_D XYZZ<F> ec_scalar_mul(const XYZZ<F> &p, const uint32_t *__restrict__ s, int bits) {
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (int b = bits - 1; b >= 0; --b) {
        acc = xyzz_dbl(acc);
        if ((s[b >> 5] >> (b & 31)) & 1) acc = xyzz_add(acc, p);
    }
    return acc;
}

// canonical Jacobian -> device XYZZ at the bit-reversed position
template <class F>
__global__ __launch_bounds__(64) void ec_ntt_load(const uint32_t *__restrict__ jac, uint32_t log_m, uint32_t *__restrict__ pts) {
    typedef FieldOps<F> O;
    constexpr int CW = O::CANON_WORDS, NL = O::WORDS;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, m = 1u << log_m;
    if (i >= m) return;
    const uint32_t *p = jac + (size_t)i * 3 * CW;
    Jacobian<F> j = {O::from_canonical(p), O::from_canonical(p + CW), O::from_canonical(p + 2 * CW)};
    const uint32_t r = log_m ? (__brev(i) >> (32 - log_m)) : 0;
    xyzz_store<F>(pts + (size_t)r * (4 * NL), xyzz_from_jacobian(j));
}

// stage s (1-based) of the decimation-in-time network: butterflies (u, v) at distance 2^(s-1), v multiplied by w^(k stride)
template <class F>
__global__ __launch_bounds__(64) void ec_ntt_stage(uint32_t *__restrict__ pts, const uint32_t *__restrict__ tw, uint32_t log_m, uint32_t s, int scalar_bits) {
    typedef FieldOps<F> O;
    constexpr int NL = O::WORDS;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, m = 1u << log_m;
    if (b >= m / 2) return;
    const uint32_t half = 1u << (s - 1), k = b & (half - 1), i = ((b >> (s - 1)) << s) + k, j = i + half;
    XYZZ<F> u = xyzz_load<F>(pts + (size_t)i * (4 * NL)), v = xyzz_load<F>(pts + (size_t)j * (4 * NL));
    if (k != 0) v = ec_scalar_mul(v, tw + (size_t)(k << (log_m - s)) * 8, scalar_bits);
    xyzz_store<F>(pts + (size_t)i * (4 * NL), xyzz_add(u, v));
    if (!v.is_inf()) v.Y = O::template sub<O::K2>(F::zero(), v.Y);
    xyzz_store<F>(pts + (size_t)j * (4 * NL), xyzz_add(u, v));
}

// device XYZZ -> canonical Jacobian, multiplied by 1/m for the inverse transform
template <class F>
__global__ __launch_bounds__(64) void ec_ntt_store(const uint32_t *__restrict__ pts, uint32_t m, const uint32_t *__restrict__ minv, int scalar_bits,
                                                   uint32_t *__restrict__ jac) {
    typedef FieldOps<F> O;
    constexpr int CW = O::CANON_WORDS, NL = O::WORDS;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    XYZZ<F> p = xyzz_load<F>(pts + (size_t)i * (4 * NL));
    if (minv) p = ec_scalar_mul(p, minv, scalar_bits);
    Jacobian<F> j = xyzz_to_jacobian(p);
    uint32_t *o = jac + (size_t)i * 3 * CW;
    O::to_canonical(o, j.X);
    O::to_canonical(o + CW, j.Y);
    O::to_canonical(o + 2 * CW, j.Z);
}

template <class F, class U>
int ec_ntt_t(zkhip_ctx *ctx, uint32_t *d_jac, size_t log_m, const uint64_t *omega, int inverse, int scalar_bits) {
    constexpr int NL = FieldOps<F>::WORDS;
    const uint32_t m = 1u << log_m, ntw = std::max<uint32_t>(1, m / 2);
    size_t need = zkhip_ctx::ws_round((size_t)m * 4 * NL * 4) + zkhip_ctx::ws_round((size_t)ntw * 32) + zkhip_ctx::ws_round(256);
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    uint32_t *pts = ctx->ws_take<uint32_t>((size_t)m * 4 * NL);
    uint32_t *tw = ctx->ws_take<uint32_t>((size_t)ntw * 8);
    uint32_t *consts = ctx->ws_take<uint32_t>(64);
    uint32_t *d_w = consts + 32;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_w, omega, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "ec_ntt_setup", ec_ntt_setup<U>, dim3(1), dim3(64), 0, d_w, (uint32_t)log_m, inverse, consts);
    ZK_LAUNCH(ctx, "ec_ntt_twiddles", ec_ntt_twiddles<U>, dim3((ntw + 255) / 256), dim3(256), 0, consts, ntw, tw);
    ZK_LAUNCH(ctx, "ec_ntt_load", ec_ntt_load<F>, dim3((m + 63) / 64), dim3(64), 0, d_jac, (uint32_t)log_m, pts);
    for (uint32_t s = 1; s <= log_m; ++s)
        ZK_LAUNCH(ctx, "ec_ntt_stage", ec_ntt_stage<F>, dim3((m / 2 + 63) / 64), dim3(64), 0, pts, tw, (uint32_t)log_m, s, scalar_bits);
    ZK_LAUNCH(ctx, "ec_ntt_store", ec_ntt_store<F>, dim3((m + 63) / 64), dim3(64), 0, pts, m, inverse ? consts + U::NL : (const uint32_t *)nullptr,
              scalar_bits, d_jac);
    return ZKHIP_OK;
}

}  // namespace

extern "C" int zkhip_ec_ntt_dev(zkhip_ctx *ctx, int curve, int group, void *d_jacobian, size_t log_m, const uint64_t *omega, int inverse) {
    if (!ctx || !d_jacobian || !omega) return ZKHIP_ERR_INVALID;
    if (log_m > 26) return ZKHIP_ERR_RANGE;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint32_t *d = (uint32_t *)d_jacobian;
    if (curve == CURVE_BLS12_381 && group == GROUP_G1) return ec_ntt_t<CurveTraits<CURVE_BLS12_381, GROUP_G1>::F, BlsFrU>(ctx, d, log_m, omega, inverse, 255);
    if (curve == CURVE_BLS12_381 && group == GROUP_G2) return ec_ntt_t<CurveTraits<CURVE_BLS12_381, GROUP_G2>::F, BlsFrU>(ctx, d, log_m, omega, inverse, 255);
    if (curve == CURVE_BN254 && group == GROUP_G1) return ec_ntt_t<CurveTraits<CURVE_BN254, GROUP_G1>::F, BnFrU>(ctx, d, log_m, omega, inverse, 254);
    if (curve == CURVE_BN254 && group == GROUP_G2) return ec_ntt_t<CurveTraits<CURVE_BN254, GROUP_G2>::F, BnFrU>(ctx, d, log_m, omega, inverse, 254);
    return ZKHIP_ERR_INVALID;
}
