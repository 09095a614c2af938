// Round 4: does a 13 x 30-bit SIGNED-limb Montgomery product beat the shipped 14 x 29-bit unsigned one, and does writing the
// product as one inline-asm block (tools/gen_mont_asm.py) remove the per-column v_lshl_add_u64 LLVM inserts?
// Timed by WALL CLOCK (hipEvents over a long dependent chain), at exactly w waves per SIMD for w = 1 ... 8 where the
// registers allow (dynamic LDS sized so that exactly w one-wave-per-SIMD workgroups fit a CU), plus the s_memtime tick
// count of the same run -- the ratio calibrates s_memtime (it does NOT tick at the shader clock on gfx950).
// Round 5: k14x29_asm = the same 14 x 29-bit product with ONE Karatsuba level (gen_mont_asm.py KBlock).
// Build: make -C tools mulbench4 ; run on the GPU box.  Prints CHECK lines that tools/mulbench4_check.py verifies.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "mont_asm.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return; } } while (0)

__device__ __constant__ uint32_t kPu14[14] = {0x1fffaaabu, 0xff7ffffu, 0x14ffffeeu, 0x17fffd62u, 0xf6241eau, 0x9507b58u, 0xafd9cc3u, 0x109e70a2u,
                                              0x1764774bu, 0x121a5d66u, 0x12c6e9edu, 0x12ffcd34u, 0x111ea3u, 0xdu};
static const int32_t hPs13[13] = {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811,
                                  316751073, -321428361, 517541167, -375082566, -91332614, 1704210};

template <int L, int B>
__device__ __forceinline__ void umul_cpp(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L], const uint32_t (&q)[L], uint32_t qinv) {
    constexpr uint32_t MASK = (1u << B) - 1;
    uint32_t m[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * q[k - i];
        m[k] = ((uint32_t)acc * qinv) & MASK;
        acc += (uint64_t)m[k] * q[0];
        acc >>= B;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * q[k - i];
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= B;
    }
    r[L - 1] = (uint32_t)acc;
}

template <int L, int B>
__device__ __forceinline__ void smul_cpp(int32_t (&r)[L], const int32_t (&a)[L], const int32_t (&b)[L], const int32_t (&p)[L], uint32_t ninv) {
    int32_t q[L];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (int64_t)q[i] * p[k - i];
        q[k] = (int32_t)(((uint32_t)acc * ninv) << (32 - B)) >> (32 - B);
        acc += (int64_t)q[k] * p[0];
        acc >>= B;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (int64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (int64_t)q[i] * p[k - i];
        r[k - L] = (int32_t)((uint32_t)acc << (32 - B)) >> (32 - B);
        acc = (acc + (1ll << (B - 1))) >> B;
    }
    r[L - 1] = (int32_t)acc;
}

#define ITERS 1500
enum { V_U14_CPP = 0, V_S13_CPP = 1, V_S13_ASM = 2, V_U14_ASM = 3, V_S13_ASM_MUL2 = 4, V_K14_ASM = 5 };

template <int V>
__global__ __launch_bounds__(256) void kmul(uint32_t *out, const uint32_t *in, uint64_t *cyc, const int32_t *ps) {
    extern __shared__ uint32_t lds_dummy[];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t t0, t1;
    if constexpr (V == V_U14_CPP || V == V_U14_ASM || V == V_K14_ASM) {
        constexpr int L = 14;
        uint32_t a[L], b[L], q[L];
        for (int i = 0; i < L; i++) {
            a[i] = in[(tid * L + i) & 4095] & ((1u << 29) - 1);
            b[i] = in[(tid * L + i + 7) & 4095] & ((1u << 29) - 1);
            q[i] = kPu14[i];
        }
        a[L - 1] &= 15; b[L - 1] &= 15;
        uint32_t a0[L];
        for (int i = 0; i < L; i++) a0[i] = a[i];
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < ITERS; ++it) {
            uint32_t r[L];
            if constexpr (V == V_U14_CPP) umul_cpp<L, 29>(r, a, b, q, 0x1ffcfffdu); else if constexpr (V == V_U14_ASM) mont_u14_mul(r, a, b, q, 0x1ffcfffdu); else mont_k14_mul(r, a, b, q, 0x1ffcfffdu);
            for (int i = 0; i < L; i++) a[i] = r[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (tid < 4) {  // one product for the checker
            uint32_t r[L];
            if constexpr (V == V_U14_CPP) umul_cpp<L, 29>(r, a0, b, q, 0x1ffcfffdu); else if constexpr (V == V_U14_ASM) mont_u14_mul(r, a0, b, q, 0x1ffcfffdu); else mont_k14_mul(r, a0, b, q, 0x1ffcfffdu);
            for (int i = 0; i < L; i++) { out[tid * 64 + i] = a0[i]; out[tid * 64 + 16 + i] = b[i]; out[tid * 64 + 32 + i] = r[i]; }
        }
        if (tid >= 4) out[1024 + tid] = a[0];
    } else {
        constexpr int L = 13;
        int32_t a[L], b[L], p[L];
        for (int i = 0; i < L; i++) {
            a[i] = (int32_t)in[(tid * L + i) & 4095] >> 2;
            b[i] = (int32_t)in[(tid * L + i + 7) & 4095] >> 2;
            p[i] = ps[i];
        }
        a[L - 1] >>= 8; b[L - 1] >>= 8;
        int32_t a0[L];
        for (int i = 0; i < L; i++) a0[i] = a[i];
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < ITERS; ++it) {
            int32_t r[L];
            if constexpr (V == V_S13_CPP) smul_cpp<L, 30>(r, a, b, p, 0x3ffcfffdu);
            else if constexpr (V == V_S13_ASM) mont_s13_mul(r, a, b, p, 0x3ffcfffdu);
            else mont_s13_mul2(r, a, b, b, a, p, 0x3ffcfffdu);
            for (int i = 0; i < L; i++) a[i] = r[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (tid < 4) {
            int32_t r[L];
            if constexpr (V == V_S13_CPP) smul_cpp<L, 30>(r, a0, b, p, 0x3ffcfffdu);
            else if constexpr (V == V_S13_ASM) mont_s13_mul(r, a0, b, p, 0x3ffcfffdu);
            else mont_s13_mul2(r, a0, b, b, a0, p, 0x3ffcfffdu);
            for (int i = 0; i < L; i++) { out[tid * 64 + i] = (uint32_t)a0[i]; out[tid * 64 + 16 + i] = (uint32_t)b[i]; out[tid * 64 + 32 + i] = (uint32_t)r[i]; }
        }
        if (tid >= 4) out[1024 + tid] = (uint32_t)a[0];
    }
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
    if (tid == 0x7fffffff) lds_dummy[0] = 1;
}

template <int V>
void run(const char *name, int L, int mads) {
    uint32_t *din, *dout; uint64_t *dc; int32_t *dps;
    const int ncu = 256;
    CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, (1024 + ncu * 8 * 256 + 64) * 4)); CK(hipMalloc(&dc, ncu * 8 * 4 * 8)); CK(hipMalloc(&dps, 13 * 4));
    std::vector<uint32_t> h(4096);
    uint32_t s = 12345; for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s ^ (s >> 13); }
    CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dps, hPs13, 13 * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)kmul<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void *)kmul<V>));
    printf("# %s: %d VGPRs, %zu B scratch\n", name, fa.numRegs, (size_t)fa.localSizeBytes);
    for (int wps = 1; wps <= 8; ++wps) {
        if (fa.numRegs * wps > 512) break;
        size_t lds = (size_t)(160 * 1024 / wps) & ~(size_t)1023; if (wps == 1) lds = 96 * 1024;
        hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
        hipLaunchKernelGGL((kmul<V>), dim3(ncu * wps), dim3(256), lds, 0, dout, din, dc, dps);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ea));
        hipLaunchKernelGGL((kmul<V>), dim3(ncu * wps), dim3(256), lds, 0, dout, din, dc, dps);
        CK(hipEventRecord(eb));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        std::vector<uint64_t> c(ncu * wps * 4); CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end());
        const double ticks = (double)c[c.size() / 2];
        const double ns_per_prod_simd = ms * 1e6 / ITERS / wps;  // one SIMD issues wps * ITERS products in ms
        printf("%-14s wps=%d  %8.1f ns per wave-product per SIMD  (%.3f ns per multiply-add)  ticks/product/SIMD %.1f  tick rate %.3f GHz  kernel %.3f ms\n", name, wps,
               ns_per_prod_simd, ns_per_prod_simd / mads, ticks / ITERS / wps, ticks / (ms * 1e6), ms);
        CK(hipEventDestroy(ea)); CK(hipEventDestroy(eb));
    }
    std::vector<uint32_t> o(256); CK(hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < 4; ++t) {
        printf("CHECK %s L=%d", name, L);
        for (int part = 0; part < 3; ++part) for (int i = 0; i < L; ++i) printf(" %d", (int32_t)o[t * 64 + part * 16 + i]);
        printf("\n");
    }
    hipFree(din); hipFree(dout); hipFree(dc); hipFree(dps);
}

int main() {
    run<V_U14_CPP>("u14x29_cpp", 14, 392);
    run<V_U14_ASM>("u14x29_asm", 14, 392);
    run<V_K14_ASM>("k14x29_asm", 14, 343);  // round 5: one Karatsuba level, same limbs, same reduction half
    run<V_U14_ASM>("u14x29_asm", 14, 392);  // again, after: the clock drifts over a run
    run<V_S13_CPP>("s13x30_cpp", 13, 338);
    run<V_S13_ASM>("s13x30_asm", 13, 338);
    run<V_S13_ASM_MUL2>("s13x30_asm_mul2", 13, 507);
    return 0;
}
