// Bernstein-Yang "safegcd" modular inverse on the kernels' own limb shape (signed 29-bit limbs): batches of 29 constant-time half-delta
// divsteps on the low words of (f, g), whose 2 x 2 transition matrix is then applied to (f, g) and, modulo p, to (d, e).  No
// data-dependent branch: the lanes of a wave do not diverge.  Built and measured in round 3 (tools/invbench.hip: a 381-bit inverse in 43 k
// VALU instructions against Fermat's 430 k); since round 5 it is the ONE inversion of a grand-product call (perm.hip), which sits alone on
// that call's critical path: 21 batches for the 255-bit scalar fields, ~15 k instructions against the 70 k of a^(p - 2).
// Checked on the host against the oracle's inverse (tests/test_host_arith.py, op 11) and on the device through every -m gpu test of the
// permutation / lookup arguments.
#pragma once
#include "fu.hpp"

namespace zkhip {

template <class U>
struct SafeGcd {
    static constexpr int L = U::L, B = U::B;
    static constexpr int32_t M = (1 << B) - 1;
    // half-delta divsteps that bring g to 0 for a modulus of `bits` bits: ceil((45907 bits + 26313) / 19929) (Bernstein-Yang, as refined by
    // Pornin / Wuille); bits = 32 NL bounds the modulus; more steps than needed change nothing (g stays 0, f stays +-1)
    static constexpr int BITS = 32 * U::NL, STEPS = (45907 * BITS + 26313 + 19928) / 19929, BATCHES = (STEPS + B - 1) / B;

    struct Mat {
        int32_t u, v, q, r;
    };
    // B half-delta divsteps on the low bits of f, g; zeta = -(delta + 1/2)
    ZK_HD static int32_t divsteps(int32_t zeta, uint32_t f0, uint32_t g0, Mat &t) {
        uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
ZK_UNROLL
        for (int i = 0; i < B; ++i) {
            uint32_t c1 = (uint32_t)(zeta >> 31);  // all ones iff zeta < 0
            const uint32_t c2 = -(g & 1u);
            const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // conditionally negated f, u, v
            g += x & c2;
            q += y & c2;
            r += z & c2;
            c1 &= c2;
            zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
            f += g & c1;
            u += q & c1;
            v += r & c1;
            g >>= 1;
            u <<= 1;
            v <<= 1;
        }
        t.u = (int32_t)u, t.v = (int32_t)v, t.q = (int32_t)q, t.r = (int32_t)r;
        return zeta;
    }
    // (f, g) <- t (f, g) / 2^B  (exact)
    ZK_HD static void update_fg(int32_t (&f)[L], int32_t (&g)[L], const Mat &t) {
        int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0], cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
        cf >>= B;
        cg >>= B;
ZK_UNROLL
        for (int i = 1; i < L; ++i) {
            cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
            cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
            f[i - 1] = (int32_t)cf & M;
            g[i - 1] = (int32_t)cg & M;
            cf >>= B;
            cg >>= B;
        }
        f[L - 1] = (int32_t)cf;
        g[L - 1] = (int32_t)cg;
    }
    // (d, e) <- t (d, e) / 2^B mod p, kept in (-2p, p)
    ZK_HD static void update_de(int32_t (&d)[L], int32_t (&e)[L], const Mat &t) {
        constexpr uint32_t PINV = (0u - U::QINV) & (uint32_t)M;  // p^-1 mod 2^B (QINV = -p^-1)
        const int32_t sd = d[L - 1] >> 31, se = e[L - 1] >> 31;
        int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
        int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0], ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
        md -= (int32_t)((PINV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M);
        me -= (int32_t)((PINV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M);
        cd += (int64_t)(int32_t)U::mod(0) * md;
        ce += (int64_t)(int32_t)U::mod(0) * me;
        cd >>= B;
        ce >>= B;
ZK_UNROLL
        for (int i = 1; i < L; ++i) {
            cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)(int32_t)U::mod(i) * md;
            ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)(int32_t)U::mod(i) * me;
            d[i - 1] = (int32_t)cd & M;
            e[i - 1] = (int32_t)ce & M;
            cd >>= B;
            ce >>= B;
        }
        d[L - 1] = (int32_t)cd;
        e[L - 1] = (int32_t)ce;
    }
    // a (plain integer, normalised 29-bit limbs, 0 < a < p)  ->  a^-1 mod p in [0, p)
    ZK_HD static Fu<U> inverse(const Fu<U> &a) {
        int32_t f[L], g[L], d[L], e[L];
ZK_UNROLL
        for (int i = 0; i < L; ++i) {
            f[i] = (int32_t)U::mod(i);
            g[i] = (int32_t)a.v[i];
            d[i] = 0;
            e[i] = i == 0 ? 1 : 0;
        }
        int32_t zeta = -1;
        for (int b = 0; b < BATCHES; ++b) {
            Mat t;
            zeta = divsteps(zeta, (uint32_t)f[0], (uint32_t)g[0], t);
            update_de(d, e, t);
            update_fg(f, g, t);
        }
        // f = +-1; d = +-a^-1 in (-2p, p): negate when f < 0, then bring into [0, p)
        const int32_t sf = f[L - 1] >> 31;
        int32_t carry = 0;
ZK_UNROLL
        for (int i = 0; i < L; ++i) {  // d = sf ? -d : d
            int32_t x = (d[i] ^ sf) - sf + carry;
            carry = i + 1 < L ? x >> B : 0;
            d[i] = i + 1 < L ? x & M : x;
        }
        for (int rep = 0; rep < 2; ++rep) {  // while d < 0: d += p  (at most twice)
            const int32_t neg = d[L - 1] >> 31;
            carry = 0;
ZK_UNROLL
            for (int i = 0; i < L; ++i) {
                int32_t x = d[i] + ((int32_t)U::mod(i) & neg) + carry;
                carry = i + 1 < L ? x >> B : 0;
                d[i] = i + 1 < L ? x & M : x;
            }
        }
        Fu<U> r;
ZK_UNROLL
        for (int i = 0; i < L; ++i) r.v[i] = (uint32_t)d[i];
        return fu_cond_sub_p(r);
    }
};

// Montgomery form in (a R, 0 < a < p, normalised limbs, value < p), Montgomery form of the inverse out (a^-1 R, canonical):
// safegcd gives (a R)^-1 = a^-1 R^-1 as a plain integer; times R^2 = one Montgomery product by R^3
template <class U>
ZK_HD Fu<U> fu_inv_gcd(const Fu<U> &a_mont) {
    const Fu<U> r3 = fu_mul(Fu<U>::r2(), Fu<U>::r2());
    return fu_cond_sub_p(fu_mul(SafeGcd<U>::inverse(a_mont), r3));
}

}  // namespace zkhip
