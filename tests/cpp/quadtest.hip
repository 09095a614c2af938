// The MSM tail's group law over lane PAIRS (csrc/fu_pair.hpp) and lane QUADS (csrc/fu_quad.hpp) against the one-lane formulas of
// csrc/curve.hpp, on arbitrary field elements (the formulas are algebraic identities; no curve membership needed), both base fields:
// addition, doubling and the small-scalar multiple, every coordinate compared after canonicalisation.
// (This is the test that caught LLVM's DPP-combine fold of two broadcasts into one subtraction: fu_quad.hpp, quad_bcast.)
// Exit code 0 and "mismatch mask 0x0" on both lines = pass.  Device only: run on the GPU box (tests/test_gpu_msm.py).
#include <hip/hip_runtime.h>

#include <cstdio>

#include "fu_quad.hpp"

using namespace zkhip;

template <class U>
__global__ void k(const uint32_t *in, uint32_t *bad) {
    const int q = (blockIdx.x * blockDim.x + threadIdx.x) / 4;
    XYZZ<Fu<U>> a, b;
    Fu<U> *fa[4] = {&a.X, &a.Y, &a.ZZ, &a.ZZZ}, *fb[4] = {&b.X, &b.Y, &b.ZZ, &b.ZZZ};
    for (int c = 0; c < 4; ++c) {
        for (int i = 0; i < U::L; ++i) {
            fa[c]->v[i] = in[(q * 97 + c * 31 + i) & 4095] & ((1u << 29) - 1);
            fb[c]->v[i] = in[(q * 89 + c * 37 + i + 1000) & 4095] & ((1u << 29) - 1);
        }
        fa[c]->v[U::L - 1] = 0, fb[c]->v[U::L - 1] = 0;  // below p, then one Montgomery product: < 2p like real coordinates
        if (U::L == 10) fa[c]->v[8] &= 0xfffff, fb[c]->v[8] &= 0xfffff;
        *fa[c] = fu_mul(*fa[c], Fu<U>::r2());
        *fb[c] = fu_mul(*fb[c], Fu<U>::r2());
    }
    const uint32_t kk = 1 + (in[q & 4095] & 0xfffff);
    // the multiple by the pair / quad code's own chain (double, then add where the bit is set): off the curve -- these are arbitrary
    // field elements -- two DIFFERENT addition chains need not meet, so the one-lane reference follows the same one
    XYZZ<Fu<U>> m1 = a;
    for (int i = 31 - __builtin_clz(kk) - 1; i >= 0; --i) {
        m1 = xyzz_dbl(m1);
        if ((kk >> i) & 1) m1 = xyzz_add(m1, a);
    }
    const XYZZ<Fu<U>> s1 = xyzz_add(a, b), d1 = xyzz_dbl(a);
    const XYZZ<FuQ<U>> aq {{a.X}, {a.Y}, {a.ZZ}, {a.ZZZ}}, bq {{b.X}, {b.Y}, {b.ZZ}, {b.ZZZ}};
    const XYZZ<FuQ<U>> s4 = xyzz_add(aq, bq), d4 = xyzz_dbl(aq), m4 = xyzz_mul_small(aq, kk);
    const XYZZ<FuP<U>> ap {{a.X}, {a.Y}, {a.ZZ}, {a.ZZZ}}, bp {{b.X}, {b.Y}, {b.ZZ}, {b.ZZZ}};
    const XYZZ<FuP<U>> s2 = xyzz_add(ap, bp), d2 = xyzz_dbl(ap), m2 = xyzz_mul_small(ap, kk);
    auto same = [](const Fu<U> &x, const Fu<U> &y) { return fu_canon(x).limbs_equal(fu_canon(y)); };
    auto same_point = [&](const XYZZ<Fu<U>> &p, const Fu<U> &X, const Fu<U> &Y, const Fu<U> &ZZ, const Fu<U> &ZZZ) {
        return same(p.X, X) && same(p.Y, Y) && same(p.ZZ, ZZ) && same(p.ZZZ, ZZZ);
    };
    unsigned m = 0;
    if (!same(s1.X, s4.X.v) || !same(s1.Y, s4.Y.v) || !same(s1.ZZ, s4.ZZ.v) || !same(s1.ZZZ, s4.ZZZ.v)) m |= 1;    // quad addition
    if (!same(d1.X, d4.X.v) || !same(d1.Y, d4.Y.v) || !same(d1.ZZ, d4.ZZ.v) || !same(d1.ZZZ, d4.ZZZ.v)) m |= 2;    // quad doubling
    if (!same_point(m1, m4.X.v, m4.Y.v, m4.ZZ.v, m4.ZZZ.v)) m |= 4;                                                   // quad multiple
    if (!same(s1.X, s2.X.v) || !same(s1.Y, s2.Y.v) || !same(s1.ZZ, s2.ZZ.v) || !same(s1.ZZZ, s2.ZZZ.v)) m |= 16;   // pair addition
    if (!same(d1.X, d2.X.v) || !same(d1.Y, d2.Y.v) || !same(d1.ZZ, d2.ZZ.v) || !same(d1.ZZZ, d2.ZZZ.v)) m |= 32;   // pair doubling
    if (!same_point(m1, m2.X.v, m2.Y.v, m2.ZZ.v, m2.ZZZ.v)) m |= 64;                                                  // pair multiple
    if (m) atomicOr(bad, m);
}

template <class U>
unsigned run(const char *name) {
    uint32_t *din, *dbad, h[4096], bad = 0xffffffffu;
    for (int i = 0; i < 4096; ++i) h[i] = i * 2654435761u + 977;
    if (hipMalloc(&din, sizeof(h)) != hipSuccess || hipMalloc(&dbad, 4) != hipSuccess) return bad;
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice), (void)hipMemset(dbad, 0, 4);
    hipLaunchKernelGGL(k<U>, dim3(16), dim3(256), 0, 0, din, dbad);
    if (hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost) != hipSuccess) bad = 0xffffffffu;
    printf("%s: mismatch mask 0x%x (quad add / dbl / multiple = 1 / 2 / 4, pair = 16 / 32 / 64)\n", name, bad);
    (void)hipFree(din), (void)hipFree(dbad);
    return bad;
}

int main() {
    const unsigned a = run<BlsFqU>("BLS12-381 Fq"), b = run<BnFqU>("BN254 Fq");
    return (a | b) ? 1 : 0;
}
