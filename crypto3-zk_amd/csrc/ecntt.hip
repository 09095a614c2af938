// Radix-2 (inverse) DFT over GROUP ELEMENTS: out[j] = sum_i omega^(i j) P_i  (inverse: omega^-1 and a final 1/m).
//
// This is evaluation_domain<Fr, G>::evaluate_all_lagrange_polynomials(powers_begin, powers_end) as the powers-of-tau
// result uses it (zk/commitments/detail/polynomial/powers_of_tau/result.hpp:81-94): with P_i = tau^i G the inverse
// transform yields L_j(tau) G for every Lagrange basis polynomial of the domain.  Setup-side
// work: every butterfly multiplies a point by a full-width twiddle, m/2 log2 m of them.
//
// Shape (round 3; round 2 ran one plain double-and-add per butterfly -- 255 doublings and, because the lanes of a wave hold
// DIFFERENT twiddles, an addition at every bit: ~1.6 M instructions per lane):
//   * a stage is a MULTIPLICATION pass (one lane per point that takes a twiddle) and a cheap butterfly pass (u + v, u - v);
//   * the multiplication is a FIXED 4-bit signed-window ladder -- every lane adds at the same 1-in-4 positions, so divergence costs
//     nothing: the lane's multiples P .. 8P (4 doublings + 3 additions) go to a table in HBM (the lane's own 8 x 4 coordinates,
//     contiguous), the twiddle is recoded once per table entry into nibbles in [-8, 7];
//   * on G1 the twiddle is split with the curve's endomorphism phi(x, y) = (beta x, y) = lambda P (GLV): k = k1 + k2 lambda with
//     |k1|, |k2| < 2^129, decomposed on the device when the twiddle records are built, so the ladder is 33 windows of 4 doublings +
//     2 additions (from the table of P and the table of phi(P): the same entries with X multiplied by beta) instead of 65 windows of
//     4 + 1: ~0.6 M instructions per lane; G2 takes the 65-window ladder (~2 x fewer additions than round 2);
//   * the 1/m of the inverse transform is folded into the LAST stage (u / m + (w^k / m) v: its own record table) -- m
//     multiplications in one parallel pass instead of m / 2 in the stage and m more afterwards.
//
// Boundary form: canonical Jacobian points (X, Y, Z; Z = 0 for infinity), as zkhip_msm_dev returns them.
#include <algorithm>

#include "ctx.hpp"
#include "curve.hpp"

using namespace zkhip;

namespace {

constexpr int REC_WORDS = 12;  // a twiddle record: GLV  [0..4] nibbles of |k1|, [5..9] nibbles of |k2|, [10] bit h = half h negative
                               //                   plain [0..8] nibbles of k (65 used), [10] = 0

// ---- GLV constants: lambda^2 + lambda + 1 = 0 (mod r), phi(P) = (beta x, y) = lambda P, lattice basis (a1, b1), (a2, b2) with
// a + b lambda = 0 (mod r); c1 = floor(k g1 / 2^256), c2 = floor(k g2 / 2^256) (g1 = floor(2^256 |b2| / r), g2 = floor(2^256 |b1| / r)),
//   k1 = k - c1 |a1| - c2 |a2|,  k2 = + c1 |b1| - c2 |b2|     (the signs below are those of both bases as chosen here)
// Any (c1, c2) gives k1 + k2 lambda = k (mod r); these keep |k1|, |k2| below 2^128 (checked over 3 x 10^5 scalars and the corners).
struct NoGlv {
    static constexpr bool ENABLED = false;
};
struct BlsGlv {  // lambda = z^2 - 1 = 0xac45a4010001a40200000000ffffffff; basis (lambda, -1), (1, lambda + 1)
    static constexpr bool ENABLED = true;
    ZK_HD static uint32_t g1(int i) { constexpr uint32_t t[5] = {0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u}; return t[i]; }
    ZK_HD static uint32_t g2(int i) { constexpr uint32_t t[5] = {0x00000002u, 0, 0, 0, 0}; return t[i]; }
    ZK_HD static uint32_t a1(int i) { constexpr uint32_t t[5] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u, 0}; return t[i]; }
    ZK_HD static uint32_t a2(int i) { constexpr uint32_t t[5] = {0x00000001u, 0, 0, 0, 0}; return t[i]; }
    ZK_HD static uint32_t b1(int i) { constexpr uint32_t t[5] = {0x00000001u, 0, 0, 0, 0}; return t[i]; }
    ZK_HD static uint32_t b2(int i) { constexpr uint32_t t[5] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u, 0}; return t[i]; }
    ZK_HD static uint32_t beta(int i) {  // canonical, 12 words
        constexpr uint32_t t[12] = {0x0000aaacu, 0x8bfd0000u, 0x4f49fffdu, 0x409427ebu, 0x0fb85f9bu, 0x897d2965u,
                                    0x89759ad4u, 0xaa0d857du, 0x63d4de85u, 0xec024086u, 0x397fe699u, 0x1a0111eau};
        return t[i];
    }
};
struct BnGlv {  // lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd; basis (a1, -b1), (a2, b2) from the extended Euclid on (r, lambda)
    static constexpr bool ENABLED = true;
    ZK_HD static uint32_t g1(int i) { constexpr uint32_t t[5] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u, 0, 0}; return t[i]; }
    ZK_HD static uint32_t g2(int i) { constexpr uint32_t t[5] = {0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u}; return t[i]; }
    ZK_HD static uint32_t a1(int i) { constexpr uint32_t t[5] = {0x94d213e3u, 0x89d32568u, 0, 0, 0}; return t[i]; }
    ZK_HD static uint32_t a2(int i) { constexpr uint32_t t[5] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u, 0}; return t[i]; }
    ZK_HD static uint32_t b1(int i) { constexpr uint32_t t[5] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u, 0}; return t[i]; }
    ZK_HD static uint32_t b2(int i) { constexpr uint32_t t[5] = {0x94d213e3u, 0x89d32568u, 0, 0, 0}; return t[i]; }
    ZK_HD static uint32_t beta(int i) {  // canonical, 8 words
        constexpr uint32_t t[8] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0, 0};
        return t[i];
    }
};

// t (6 words) +/-= a (5 words) * b (5 words)   modulo 2^192
ZK_D void acc_mul_192(uint32_t *t, const uint32_t *a, const uint32_t *b, bool subtract) {
    uint32_t prod[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) {
        uint64_t carry = 0;
        for (int j = 0; i + j < 6 && j < 5; ++j) {
            const uint64_t cur = (uint64_t)prod[i + j] + (uint64_t)a[i] * b[j] + carry;
            prod[i + j] = (uint32_t)cur;
            carry = cur >> 32;
        }
        if (i + 5 < 6) prod[i + 5] = (uint32_t)carry;
    }
    uint64_t c = subtract ? 1 : 0;  // t - prod = t + ~prod + 1
    for (int i = 0; i < 6; ++i) {
        const uint64_t cur = (uint64_t)t[i] + (subtract ? (uint32_t)~prod[i] : prod[i]) + c;
        t[i] = (uint32_t)cur;
        c = cur >> 32;
    }
}

// words 8..12 of k (8 words) * g (5 words)
template <class GetG>
ZK_D void mul_hi_256(uint32_t *out5, const uint32_t *k, GetG g) {
    uint32_t t[13];
    for (int i = 0; i < 13; ++i) t[i] = 0;
    for (int i = 0; i < 8; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 5; ++j) {
            const uint64_t cur = (uint64_t)t[i + j] + (uint64_t)k[i] * g(j) + carry;
            t[i + j] = (uint32_t)cur;
            carry = cur >> 32;
        }
        t[i + 5] = (uint32_t)carry;
    }
    for (int i = 0; i < 5; ++i) out5[i] = t[8 + i];
}

// signed 4-bit windows of an unsigned magnitude: digit w in [-8, 7] as a two's-complement nibble, sum_w d_w 16^w = mag
ZK_D void recode_nibbles(uint32_t *out, int out_words, const uint32_t *mag, int mag_words, int windows) {
    for (int i = 0; i < out_words; ++i) out[i] = 0;
    uint32_t carry = 0;
    for (int w = 0; w < windows; ++w) {
        const int word = w >> 3;
        uint32_t d = (word < mag_words ? (mag[word] >> ((w & 7) * 4)) & 15u : 0u) + carry;
        carry = d >= 8 ? 1 : 0;  // d - 16 in [-8, 0]: the nibble (d & 15) read as two's complement
        out[word] |= (d & 15u) << ((w & 7) * 4);
    }
}

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

// rec[j] = the window record of w^j (times `scale` when given) for j < count: square-and-multiply over the bits of j, then the
// GLV split (G1) and the signed-nibble recoding
template <class U, class G>
__global__ __launch_bounds__(256) void ec_ntt_records(const uint32_t *__restrict__ consts, const uint32_t *__restrict__ scale_c, uint32_t count,
                                                       uint32_t *__restrict__ rec) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fu<U> pw = fu_from_canonical<U>(consts), acc = Fu<U>::one();
    for (uint32_t e = j; e; e >>= 1) {
        if (e & 1) acc = fu_mul_call(acc, pw);
        pw = fu_mul_call(pw, pw);
    }
    if (scale_c) acc = fu_mul_call(acc, fu_from_canonical<U>(scale_c));
    uint32_t k[8];
    fu_to_canonical<U>(k, acc);
    uint32_t *out = rec + (size_t)j * REC_WORDS;
    if constexpr (G::ENABLED) {
        uint32_t c1[5], c2[5], cst[5], k1[6], k2[6] = {0, 0, 0, 0, 0, 0};
        mul_hi_256(c1, k, [](int i) { return G::g1(i); });
        mul_hi_256(c2, k, [](int i) { return G::g2(i); });
        for (int i = 0; i < 6; ++i) k1[i] = k[i];  // k modulo 2^192: k1 and k2 are small, the arithmetic is exact in two's complement
        for (int i = 0; i < 5; ++i) cst[i] = G::a1(i);
        acc_mul_192(k1, c1, cst, true);
        for (int i = 0; i < 5; ++i) cst[i] = G::a2(i);
        acc_mul_192(k1, c2, cst, true);
        for (int i = 0; i < 5; ++i) cst[i] = G::b1(i);
        acc_mul_192(k2, c1, cst, false);
        for (int i = 0; i < 5; ++i) cst[i] = G::b2(i);
        acc_mul_192(k2, c2, cst, true);
        uint32_t signs = 0;
        uint32_t *half[2] = {k1, k2};
        for (int h = 0; h < 2; ++h) {
            if (half[h][5] >> 31) {  // negative: take the magnitude
                signs |= 1u << h;
                uint64_t c = 1;
                for (int i = 0; i < 6; ++i) {
                    const uint64_t cur = (uint64_t)(uint32_t)~half[h][i] + c;
                    half[h][i] = (uint32_t)cur;
                    c = cur >> 32;
                }
            }
            recode_nibbles(out + 5 * h, 5, half[h], 6, 33);
        }
        out[10] = signs;
        out[11] = 0;
    } else {
        recode_nibbles(out, 9, k, 8, 65);
        out[9] = out[10] = out[11] = 0;
    }
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

template <class F>
ZK_D XYZZ<F> xyzz_dbl4(XYZZ<F> a) {
    a = xyzz_dbl(a);
    a = xyzz_dbl(a);
    a = xyzz_dbl(a);
    return xyzz_dbl(a);
}

// The multiplication pass.  mode 0: stage s (1-based) of the decimation-in-time network, lane t = butterfly t multiplies its lower
// input v by w^(k stride) (k = 0: nothing to do); mode 1: the last stage of the INVERSE transform, lane t = point t, upper inputs take
// 1/m, lower inputs w^k / m.  Lanes first .. first + gridDim.x * 64 of `total`; `tbl` holds 8 * HALVES points per lane of the launch.
template <class F, class G>
__global__ __launch_bounds__(64, G::ENABLED ? 2 : 1) void ec_ntt_mul_pass(uint32_t *__restrict__ pts, const uint32_t *__restrict__ rec_stage, const uint32_t *__restrict__ rec_last,
                                                      const uint32_t *__restrict__ rec_minv, uint32_t log_m, uint32_t s, int mode, uint32_t first,
                                                      uint32_t total, uint32_t *__restrict__ tbl) {
    typedef FieldOps<F> O;
    constexpr int NL = O::WORDS, PW = 4 * NL, HALVES = G::ENABLED ? 2 : 1, NW = G::ENABLED ? 33 : 65;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, t = first + slot;
    if (t >= total) return;
    const uint32_t half = 1u << (s - 1);
    uint32_t j;
    const uint32_t *rec;
    if (mode == 0) {
        const uint32_t k = t & (half - 1);
        if (k == 0) return;
        j = ((t >> (s - 1)) << s) + k + half;
        rec = rec_stage + (size_t)(k << (log_m - s)) * REC_WORDS;
    } else {
        j = t;
        rec = (t & half) ? rec_last + (size_t)(t & (half - 1)) * REC_WORDS : rec_minv;
    }
    uint32_t *pj = pts + (size_t)j * PW;
    const XYZZ<F> p = xyzz_load<F>(pj);
    if (p.is_inf()) return;
    // the lane's table: e P for e = 1 .. 8 (and phi(e P) behind them); every entry is re-read from memory where it is needed again
    uint32_t *T = tbl + (size_t)slot * (8 * HALVES) * PW;
    auto entry = [&](int e) { return T + (size_t)(e - 1) * PW; };
    {
        xyzz_store<F>(entry(1), p);
        XYZZ<F> e2 = xyzz_dbl(p);
        xyzz_store<F>(entry(2), e2);
        xyzz_store<F>(entry(3), xyzz_add(e2, p));
        e2 = xyzz_dbl(e2);
        xyzz_store<F>(entry(4), e2);
        xyzz_store<F>(entry(5), xyzz_add(e2, p));
        xyzz_store<F>(entry(8), xyzz_dbl(e2));
        e2 = xyzz_dbl(xyzz_load<F>(entry(3)));
        xyzz_store<F>(entry(6), e2);
        xyzz_store<F>(entry(7), xyzz_add(e2, p));
    }
    if constexpr (G::ENABLED) {
        uint32_t bc[O::CANON_WORDS];
        for (int i = 0; i < O::CANON_WORDS; ++i) bc[i] = G::beta(i);
        const F beta = O::from_canonical(bc);
        for (int e = 1; e <= 8; ++e) {
            XYZZ<F> q = xyzz_load<F>(entry(e));
            q.X = O::mul(q.X, beta);
            xyzz_store<F>(entry(8 + e), q);
        }
    }
    const uint32_t signs = rec[10];
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (int w = NW - 1; w >= 0; --w) {
        acc = xyzz_dbl4(acc);
#pragma unroll 1
        for (int h = 0; h < HALVES; ++h) {
            const int d = (int)(((rec[h * 5 + (w >> 3)] >> ((w & 7) * 4)) & 15u) ^ 8u) - 8;
            if (d != 0) {
                XYZZ<F> e = xyzz_load<F>(entry(8 * h + (d < 0 ? -d : d)));
                if ((d < 0) != (((signs >> h) & 1u) != 0)) e.Y = O::template sub<O::K2>(F::zero(), e.Y);
                acc = xyzz_add(acc, e);
            }
        }
    }
    xyzz_store<F>(pj, acc);
}

// the butterflies of stage s: (u, v) at distance 2^(s-1) -> (u + v, u - v)
template <class F>
__global__ __launch_bounds__(64) void ec_ntt_butterflies(uint32_t *__restrict__ pts, uint32_t log_m, uint32_t s) {
    typedef FieldOps<F> O;
    constexpr int NL = O::WORDS;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, m = 1u << log_m;
    if (b >= m / 2) return;
    const uint32_t half = 1u << (s - 1), k = b & (half - 1), i = ((b >> (s - 1)) << s) + k, j = i + half;
    XYZZ<F> u = xyzz_load<F>(pts + (size_t)i * (4 * NL)), v = xyzz_load<F>(pts + (size_t)j * (4 * NL));
    xyzz_store<F>(pts + (size_t)i * (4 * NL), xyzz_add(u, v));
    if (!v.is_inf()) v.Y = O::template sub<O::K2>(F::zero(), v.Y);
    xyzz_store<F>(pts + (size_t)j * (4 * NL), xyzz_add(u, v));
}

// device XYZZ -> canonical Jacobian
template <class F>
__global__ __launch_bounds__(64) void ec_ntt_store(const uint32_t *__restrict__ pts, uint32_t m, uint32_t *__restrict__ jac) {
    typedef FieldOps<F> O;
    constexpr int CW = O::CANON_WORDS, NL = O::WORDS;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Jacobian<F> j = xyzz_to_jacobian(xyzz_load<F>(pts + (size_t)i * (4 * NL)));
    uint32_t *o = jac + (size_t)i * 3 * CW;
    O::to_canonical(o, j.X);
    O::to_canonical(o + CW, j.Y);
    O::to_canonical(o + 2 * CW, j.Z);
}

template <class F, class U, class G>
int ec_ntt_t(zkhip_ctx *ctx, uint32_t *d_jac, size_t log_m, const uint64_t *omega, int inverse) {
    constexpr int NL = FieldOps<F>::WORDS, PW = 4 * NL, HALVES = G::ENABLED ? 2 : 1;
    const uint32_t m = 1u << log_m, ntw = std::max<uint32_t>(1, m / 2);
    // table slots: the lanes of one multiplication launch (<= 1 GiB of tables; a pass over more lanes runs in several launches)
    const size_t slot_bytes = (size_t)8 * HALVES * PW * 4;
    const size_t slot_cap = ctx->opt_ec_ntt_table_lanes ? ((size_t)ctx->opt_ec_ntt_table_lanes + 63) / 64 * 64 : ((size_t)1 << 30) / slot_bytes / 64 * 64;
    const uint32_t slots = (uint32_t)std::max<size_t>(64, std::min<size_t>(((size_t)m + 63) / 64 * 64, slot_cap));
    const size_t need = zkhip_ctx::ws_round((size_t)m * PW * 4) + 2 * zkhip_ctx::ws_round((size_t)ntw * REC_WORDS * 4) + zkhip_ctx::ws_round(512) +
                        zkhip_ctx::ws_round((size_t)slots * slot_bytes);
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    uint32_t *pts = ctx->ws_take<uint32_t>((size_t)m * PW);
    uint32_t *rec = ctx->ws_take<uint32_t>((size_t)ntw * REC_WORDS);
    uint32_t *rec_last = ctx->ws_take<uint32_t>((size_t)ntw * REC_WORDS);
    uint32_t *consts = ctx->ws_take<uint32_t>(128);
    uint32_t *tbl = ctx->ws_take<uint32_t>((size_t)slots * slot_bytes / 4);
    uint32_t *d_w = consts + 32, *d_one = consts + 48, *rec_minv = consts + 64;
    uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_w, omega, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "ec_ntt_setup", ec_ntt_setup<U>, dim3(1), dim3(64), 0, d_w, (uint32_t)log_m, inverse, consts);
    ZK_LAUNCH(ctx, "ec_ntt_records", (ec_ntt_records<U, G>), dim3((ntw + 255) / 256), dim3(256), 0, consts, (const uint32_t *)nullptr, ntw, rec);
    if (inverse && log_m > 0) {
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_one, one, 32, hipMemcpyHostToDevice, ctx->stream));
        ZK_LAUNCH(ctx, "ec_ntt_records", (ec_ntt_records<U, G>), dim3((ntw + 255) / 256), dim3(256), 0, consts, consts + U::NL, ntw, rec_last);
        ZK_LAUNCH(ctx, "ec_ntt_records", (ec_ntt_records<U, G>), dim3(1), dim3(256), 0, d_one, consts + U::NL, 1u, rec_minv);
    }
    ZK_LAUNCH(ctx, "ec_ntt_load", ec_ntt_load<F>, dim3((m + 63) / 64), dim3(64), 0, d_jac, (uint32_t)log_m, pts);
    for (uint32_t s = 1; s <= log_m; ++s) {
        const int mode = (inverse && s == log_m) ? 1 : 0;
        const uint32_t total = mode ? m : m / 2;
        if (mode == 1 || s > 1)  // stage 1 has no twiddle but 1
            for (uint32_t first = 0; first < total; first += slots) {
                const uint32_t lanes = std::min(slots, total - first);
                ZK_LAUNCH(ctx, "ec_ntt_mul_pass", (ec_ntt_mul_pass<F, G>), dim3((lanes + 63) / 64), dim3(64), 0, pts, rec, rec_last, rec_minv, (uint32_t)log_m, s,
                          mode, first, total, tbl);
            }
        ZK_LAUNCH(ctx, "ec_ntt_butterflies", ec_ntt_butterflies<F>, dim3((m / 2 + 63) / 64), dim3(64), 0, pts, (uint32_t)log_m, s);
    }
    ZK_LAUNCH(ctx, "ec_ntt_store", ec_ntt_store<F>, dim3((m + 63) / 64), dim3(64), 0, pts, m, d_jac);
    return ZKHIP_OK;
}

}  // namespace

extern "C" int zkhip_ec_ntt_dev(zkhip_ctx *ctx, int curve, int group, void *d_jacobian, size_t log_m, const uint64_t *omega, int inverse) {
    if (!ctx || !d_jacobian || !omega) return ZKHIP_ERR_INVALID;
    if (log_m > 26) return ZKHIP_ERR_RANGE;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint32_t *d = (uint32_t *)d_jacobian;
    if (curve == CURVE_BLS12_381 && group == GROUP_G1) return ec_ntt_t<CurveTraits<CURVE_BLS12_381, GROUP_G1>::F, BlsFrU, BlsGlv>(ctx, d, log_m, omega, inverse);
    if (curve == CURVE_BLS12_381 && group == GROUP_G2) return ec_ntt_t<CurveTraits<CURVE_BLS12_381, GROUP_G2>::F, BlsFrU, NoGlv>(ctx, d, log_m, omega, inverse);
    if (curve == CURVE_BN254 && group == GROUP_G1) return ec_ntt_t<CurveTraits<CURVE_BN254, GROUP_G1>::F, BnFrU, BnGlv>(ctx, d, log_m, omega, inverse);
    if (curve == CURVE_BN254 && group == GROUP_G2) return ec_ntt_t<CurveTraits<CURVE_BN254, GROUP_G2>::F, BnFrU, NoGlv>(ctx, d, log_m, omega, inverse);
    return ZKHIP_ERR_INVALID;
}
