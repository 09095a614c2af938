// quad / pair group law against the one-lane formulas on arbitrary field elements (the formulas are algebraic identities)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fu_quad.hpp"
using namespace zkhip;
template <class U>
__global__ void k(const uint32_t *in, uint32_t *bad, uint32_t *dump) {
    const int q = (blockIdx.x * blockDim.x + threadIdx.x) / 4;
    XYZZ<Fu<U>> a, b;
    Fu<U> *fa[4] = {&a.X, &a.Y, &a.ZZ, &a.ZZZ}, *fb[4] = {&b.X, &b.Y, &b.ZZ, &b.ZZZ};
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < U::L; ++i) {
            fa[c]->v[i] = in[(q * 97 + c * 31 + i) & 4095] & ((1u << 29) - 1);
            fb[c]->v[i] = in[(q * 89 + c * 37 + i + 1000) & 4095] & ((1u << 29) - 1);
        }
    for (int c = 0; c < 4; ++c) {  // below p: clear the top limbs, then one Montgomery product makes them < 2p like real coordinates
        fa[c]->v[U::L - 1] = 0, fb[c]->v[U::L - 1] = 0;
        if (U::L == 10) fa[c]->v[8] &= 0xfffff, fb[c]->v[8] &= 0xfffff;
        *fa[c] = fu_mul(*fa[c], Fu<U>::r2());
        *fb[c] = fu_mul(*fb[c], Fu<U>::r2());
    }
    const XYZZ<Fu<U>> s1 = xyzz_add(a, b), d1 = xyzz_dbl(a);
    XYZZ<FuQ<U>> aq {{a.X}, {a.Y}, {a.ZZ}, {a.ZZZ}}, bq {{b.X}, {b.Y}, {b.ZZ}, {b.ZZZ}};
    const XYZZ<FuQ<U>> s4 = xyzz_add(aq, bq), d4 = xyzz_dbl(aq);
    const XYZZ<FuQ<U>> d4e = zkhip::xyzz_dbl<U>(aq);    // my overload, named explicitly
    XYZZ<FuP<U>> ap {{a.X}, {a.Y}, {a.ZZ}, {a.ZZZ}}, bp {{b.X}, {b.Y}, {b.ZZ}, {b.ZZZ}};
    const XYZZ<FuP<U>> s2 = xyzz_add(ap, bp), d2 = xyzz_dbl(ap);
    auto same = [](const Fu<U> &x, const Fu<U> &y) { return fu_canon(x).limbs_equal(fu_canon(y)); };
    unsigned m = 0;
    if (!same(s1.X, s4.X.v)) m |= 1;
    if (!same(s1.Y, s4.Y.v)) m |= 2;
    if (!same(s1.ZZ, s4.ZZ.v)) m |= 4;
    if (!same(s1.ZZZ, s4.ZZZ.v)) m |= 8;
    if (!same(d1.X, d4.X.v)) m |= 16;
    if (!same(d1.Y, d4.Y.v)) m |= 32;
    if (!same(d1.ZZ, d4.ZZ.v)) m |= 64;
    if (!same(d1.ZZZ, d4.ZZZ.v)) m |= 128;
    if (!fu_canon(d1.Y).limbs_equal(fu_canon(d4e.Y.v))) m |= 1u << 9;
    if (!d4e.Y.v.limbs_equal(d4.Y.v)) m |= 1u << 10;    // the unqualified call computed something else than the explicit one
    if (!same(s1.X, s2.X.v) || !same(s1.Y, s2.Y.v) || !same(d1.Y, d2.Y.v)) m |= 256;
    {   // the pieces of the doubling's Y3, one by one
        typedef FieldOps<Fu<U>> O;
        const Fu<U> Uu = fu_add(a.Y, a.Y), V = fu_mul(Uu, Uu), XX = fu_mul(a.X, a.X), M = fu_add(fu_add(XX, XX), XX), W = fu_mul(Uu, V), S = fu_mul(a.X, V);
        const Fu<U> X3 = fu_sub<O::K1>(fu_mul(M, M), fu_add(S, S)), D = fu_sub<O::K2>(S, X3);
        const QuadProducts<U> s3 = quad_mul(M, D, W, a.Y, W, a.ZZZ, M, D);
        if (!same(s3.p0, fu_mul(M, D))) m |= 1u << 12;
        if (!same(s3.p1, fu_mul(W, a.Y))) m |= 1u << 13;
        if (!same(s3.p2, fu_mul(W, a.ZZZ))) m |= 1u << 14;
        const Fu<U> y_sep = fu_sub<O::K1>(fu_mul(M, D), fu_mul(W, a.Y));
        if (!same(y_sep, d1.Y)) m |= 1u << 15;                                   // separate reductions against mul2, one lane
        if (!same(fu_sub<O::K1>(s3.p0, s3.p1), y_sep)) m |= 1u << 16;
        if (!same(X3, d1.X)) m |= 1u << 17;
    }
    {   // fu_mul2 inline / out of line / as two products
        const Fu<U> t2 = fu_mul2(a.X, a.Y, a.ZZ, a.ZZZ), t2c = fu_mul2_call(a.X, a.Y, a.ZZ, a.ZZZ), ts = fu_add(fu_mul(a.X, a.Y), fu_mul(a.ZZ, a.ZZZ));
        if (!same(t2, ts)) m |= 1u << 18;
        if (!same(t2c, ts)) m |= 1u << 19;
    }
    {   // the quad doubling again, step by step, every intermediate against the one-lane value
        typedef FieldOps<Fu<U>> O;
        const Fu<U> Uu = fu_add(a.Y, a.Y);
        const QuadProducts<U> s1 = quad_mul(Uu, Uu, a.X, a.X, Uu, Uu, a.X, a.X);
        const Fu<U> V = s1.p0, XX = s1.p1;
        const Fu<U> M = fu_add(fu_add(XX, XX), XX);
        const QuadProducts<U> s2 = quad_mul(Uu, V, a.X, V, V, a.ZZ, M, M);
        const Fu<U> W = s2.p0, S = s2.p1;
        const Fu<U> X3 = fu_sub<O::K1>(s2.p3, fu_add(S, S));
        const Fu<U> D = fu_sub<O::K2>(S, X3);
        const QuadProducts<U> s3 = quad_mul(M, D, W, a.Y, W, a.ZZZ, M, D);
        const Fu<U> Y3 = fu_sub<O::K1>(s3.p0, s3.p1);
        const Fu<U> rV = fu_mul(Uu, Uu), rXX = fu_mul(a.X, a.X), rM = fu_add(fu_add(rXX, rXX), rXX), rW = fu_mul(Uu, rV), rS = fu_mul(a.X, rV);
        const Fu<U> rX3 = fu_sub<O::K1>(fu_mul(rM, rM), fu_add(rS, rS)), rD = fu_sub<O::K2>(rS, rX3);
        if (!same(V, rV)) m |= 1u << 20;
        if (!same(XX, rXX)) m |= 1u << 21;
        if (!same(W, rW)) m |= 1u << 22;
        if (!same(S, rS)) m |= 1u << 23;
        if (!same(X3, rX3)) m |= 1u << 24;
        if (!same(D, rD)) m |= 1u << 25;
        if (!same(s3.p0, fu_mul(rM, rD))) m |= 1u << 26;
        if (!same(s3.p1, fu_mul(rW, a.Y))) m |= 1u << 27;
        if (!same(Y3, d1.Y)) m |= 1u << 28;
        if (!same(Y3, d4.Y.v)) m |= 1u << 29;
        if (!Y3.limbs_equal(d4e.Y.v)) m |= 1u << 30;     // the explicit overload against its own replica, bit for bit
        if (!X3.limbs_equal(d4e.X.v)) m |= 1u << 31;
        if (!s2.p2.limbs_equal(d4e.ZZ.v)) m |= 1u << 11;
    }
    if (m && q == 0 && (threadIdx.x & 3) == 0 && dump) {
        const Fu<U> y1 = fu_canon(d1.Y), y4 = fu_canon(d4.Y.v), raw = d4.Y.v;
        for (int i = 0; i < U::L; ++i) dump[i] = y1.v[i], dump[16 + i] = y4.v[i], dump[32 + i] = raw.v[i], dump[48 + i] = d1.Y.v[i];
    }
    if (m) atomicOr(bad, m);
}
template <class U>
void run(const char *name) {
    uint32_t *din, *dbad, *ddump, h[4096], bad = 0, dump[64] = {0};
    hipMalloc(&ddump, 256), hipMemset(ddump, 0, 256);
    for (int i = 0; i < 4096; ++i) h[i] = i * 2654435761u + 977;
    hipMalloc(&din, sizeof(h)), hipMalloc(&dbad, 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice), hipMemset(dbad, 0, 4);
    hipLaunchKernelGGL(k<U>, dim3(16), dim3(256), 0, 0, din, dbad, ddump);
    hipMemcpy(dump, ddump, 256, hipMemcpyDeviceToHost);
    hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    if (dump[0] | dump[16]) {
        for (int r = 0; r < 4; ++r) {
            printf("  %s:", r == 0 ? "single canon" : r == 1 ? "quad canon  " : r == 2 ? "quad raw    " : "single raw  ");
            for (int i = 0; i < U::L; ++i) printf(" %08x", dump[16 * r + i]);
            printf("\n");
        }
    }
    printf("%s: mismatch mask 0x%x (add X/Y/ZZ/ZZZ = 1/2/4/8, dbl = 16/32/64/128, pair = 256)\n", name, bad);
}
int main() {
    run<BlsFqU>("BLS12-381 Fq");
    run<BnFqU>("BN254 Fq");
    return 0;
}
