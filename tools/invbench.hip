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

#include "fu_safegcd.hpp"  // SafeGcd<U> lives with the kernels since round 5

using namespace zkhip;

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
