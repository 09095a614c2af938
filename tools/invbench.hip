// what does a modular inversion cost on this arithmetic?  (The batched-affine bucket accumulation needs one
// per k additions; the kill criterion is 12 k VALU instructions.)
//
// A Bernstein-Yang "safegcd" inverse (constant-time divsteps, no data-dependent branches: lanes of a wave do not diverge) on the
// kernels' own limb shape: signed 29-bit limbs, batches of 29 divsteps whose 2 x 2 transition matrix is applied to (f, g) and,
// modulo p, to (d, e).  31 batches cover the 879 half-delta divsteps a 381-bit modulus needs (ceil((45907 * 381 + 26313) / 19929)).
// Checked against the product a * a^-1 = 1 on the host and on the device, then timed per wave next to the Montgomery product.
// Build: hipcc --offload-arch=gfx950 -O3 -I crypto3-zk_amd/csrc tools/invbench.hip -o tools/invbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "fu.hpp"

using namespace zkhip;

template <class U>
struct SafeGcd {
    static constexpr int L = U::L, B = U::B;
    static constexpr int32_t M = (1 << B) - 1;
    static constexpr int BATCHES = 31;

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

// a (plain) * inv (plain) == 1 mod p ?
template <class U>
ZK_HD bool is_inverse(const Fu<U> &a, const Fu<U> &inv) {
    const Fu<U> one = fu_cond_sub_p(fu_mul(fu_mul(a, inv), Fu<U>::r2()));  // (a inv / R) R^2 / R = a inv
    bool ok = one.v[0] == 1;
    for (int i = 1; i < U::L; ++i) ok = ok && one.v[i] == 0;
    return ok;
}

#define ITERS 20
template <class U, int WHAT>
__global__ void k(uint32_t *out, const uint32_t *in, uint64_t *cyc, uint32_t *bad) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fu<U> a;
    for (int i = 0; i < U::L; ++i) a.v[i] = in[(tid * U::L + i) & 1023] & ((1u << 28) - 1);
    a.v[U::L - 1] &= 7;  // < p
    a.v[0] |= 1;
    const Fu<U> a0 = a;
    uint64_t t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (WHAT == 0) a = SafeGcd<U>::inverse(a);  // a dependent chain
        else if constexpr (WHAT == 1) a = fu_mul(a, a0);
        else a = fu_cond_sub_p(fu_inv(a));  // Fermat, for scale (Montgomery in, Montgomery out)
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (WHAT == 0 && (ITERS % 2) == 0 && !a.limbs_equal(a0)) atomicAdd(bad, 1u);  // an even number of inversions returns a
    if (WHAT == 0) {
        const Fu<U> i1 = SafeGcd<U>::inverse(a0);
        if (!is_inverse(a0, i1)) atomicAdd(bad, 1u);
    }
    for (int i = 0; i < U::L; ++i) out[tid * U::L + i] = a.v[i];
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

template <class U, int WHAT>
double run(const char *name, int wps) {
    uint32_t *din, *dout, *dbad;
    uint64_t *dc;
    hipMalloc(&din, 4096);
    hipMalloc(&dout, 256 * 1024 * U::L * 4);
    hipMalloc(&dc, 256 * 16 * 8);
    hipMalloc(&dbad, 4);
    hipMemset(dbad, 0, 4);
    uint32_t h[1024];
    for (int i = 0; i < 1024; i++) h[i] = i * 2654435761u + 12345;
    hipMemcpy(din, h, 4096, hipMemcpyHostToDevice);
    const int threads = 256 * wps;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<U, WHAT>), dim3(256), dim3(threads), 0, 0, dout, din, dc, dbad);
        hipDeviceSynchronize();
    }
    uint64_t c = 0;
    uint32_t bad = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    const double per = (double)c / ITERS;
    printf("%-28s waves/SIMD=%d  cycles per wave-op = %9.1f   wrong results: %u\n", name, wps, per, bad);
    hipFree(din), hipFree(dout), hipFree(dc), hipFree(dbad);
    return per;
}

int main() {
    // host check first: the same code, a few hundred values
    int bad = 0;
    uint32_t s = 12345;
    for (int t = 0; t < 300; ++t) {
        Fu<BlsFqU> a;
        for (int i = 0; i < BlsFqU::L; ++i) {
            s = s * 1664525u + 1013904223u;
            a.v[i] = (s >> 3) & ((1u << 29) - 1);
        }
        a.v[BlsFqU::L - 1] &= 7;
        if (t == 0) {
            for (int i = 0; i < BlsFqU::L; ++i) a.v[i] = i == 0 ? 1 : 0;  // 1
        }
        if (t == 1) {
            for (int i = 0; i < BlsFqU::L; ++i) a.v[i] = BlsFqU::mod(i);  // p - 1
            a.v[0] -= 1;
        }
        if (!is_inverse(a, SafeGcd<BlsFqU>::inverse(a))) ++bad;
    }
    printf("host check, 300 values (incl. 1 and p - 1): %d wrong\n", bad);
    const double mul = run<BlsFqU, 1>("fu_mul (Montgomery product)", 1);
    const double inv = run<BlsFqU, 0>("safegcd inverse (31 x 29)", 1);
    const double fer = run<BlsFqU, 2>("fu_inv (Fermat)", 1);
    run<BlsFqU, 0>("safegcd inverse (31 x 29)", 3);
    printf("inverse / product = %.1f;  Fermat / product = %.1f\n", inv / mul, fer / mul);
    printf("VALU instructions per inverse ~ cycles / 5 (a lone wave issues one VALU instruction per ~5 cycles, DESIGN.md section 4) = %.0f\n", inv / 5);
    return bad != 0;
}
