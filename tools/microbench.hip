// Instruction-throughput microbenchmark for the integer/f64 VALU paths that bound modular arithmetic on
// gfx950.  Prints cycles per wave-instruction per SIMD (s_memtime based) at 1, 2, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define ITERS 512
#define CHAINS 8

#define KERNEL_BEGIN(name)                                                                  \
    __global__ void name(uint64_t *out, uint64_t *cycles, uint32_t seed) {                   \
        uint32_t a = seed * 2654435761u + threadIdx.x, b = a ^ 0x9e3779b9u;                  \
        uint64_t acc[CHAINS];                                                                \
        for (int i = 0; i < CHAINS; ++i) acc[i] = ((uint64_t)(a + i) << 32) | (b + i);        \
        uint64_t t0, t1;                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
        for (int it = 0; it < ITERS; ++it) {
#define KERNEL_END                                                                           \
        }                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
        uint64_t s = 0;                                                                      \
        for (int i = 0; i < CHAINS; ++i) s ^= acc[i];                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 32x32+64 -> 64
#define OP_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
KERNEL_BEGIN(k_mad_u64_u32) REP8(OP_MAD64) KERNEL_END

#define OP_MULLO(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
KERNEL_BEGIN(k_mul_lo_u32) REP8(OP_MULLO) KERNEL_END

#define OP_MULHI(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
KERNEL_BEGIN(k_mul_hi_u32) REP8(OP_MULHI) KERNEL_END

#define OP_MAD24(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
KERNEL_BEGIN(k_mad_u32_u24) REP8(OP_MAD24) KERNEL_END

#define OP_MULHI24(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
KERNEL_BEGIN(k_mul_hi_u32_u24) REP8(OP_MULHI24) KERNEL_END

#define OP_FMA64(i) { double d = __longlong_as_double(acc[i]); double x = 1.0000001, y = 1e-9; asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y)); acc[i] = __double_as_longlong(d); }
KERNEL_BEGIN(k_fma_f64) REP8(OP_FMA64) KERNEL_END

#define OP_FMA32(i) { float d = __uint_as_float((uint32_t)acc[i]); float x = 1.0000001f, y = 1e-9f; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y)); acc[i] = __float_as_uint(d); }
KERNEL_BEGIN(k_fma_f32) REP8(OP_FMA32) KERNEL_END

#define OP_ADD32(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
KERNEL_BEGIN(k_add_u32) REP8(OP_ADD32) KERNEL_END

#define OP_ADDCO(i) { uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc"); acc[i] = ((uint64_t)hi << 32) | lo; }
KERNEL_BEGIN(k_add_co_addc_pair) REP8(OP_ADDCO) KERNEL_END

#define OP_LSHLADD64(i) { uint64_t k = ((uint64_t)b << 32) | a; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(k)); }
KERNEL_BEGIN(k_lshl_add_u64) REP8(OP_LSHLADD64) KERNEL_END

#define OP_MOV(i) { uint32_t lo = (uint32_t)acc[i], t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(lo)); acc[i] = t + 1; }
KERNEL_BEGIN(k_mov_plus_add) REP8(OP_MOV) KERNEL_END

#define OP_ALIGNBIT(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(lo) : "v"(a)); acc[i] = lo; }
KERNEL_BEGIN(k_alignbit_b32) REP8(OP_ALIGNBIT) KERNEL_END

#define OP_LSHR64(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
KERNEL_BEGIN(k_lshrrev_b64) REP8(OP_LSHR64) KERNEL_END

#define OP_AND_OR(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
KERNEL_BEGIN(k_and_or_b32) REP8(OP_AND_OR) KERNEL_END

#define OP_CNDMASK(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(a) : "vcc"); acc[i] = lo; }
KERNEL_BEGIN(k_cndmask_b32) REP8(OP_CNDMASK) KERNEL_END

typedef void (*kern_t)(uint64_t *, uint64_t *, uint32_t);
struct Entry { const char *name; kern_t k; int instr_per_chain; };

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    printf("device %s, CUs %d, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int ncu = prop.multiProcessorCount;
    uint64_t *d_out, *d_cyc;
    size_t maxthreads = (size_t)ncu * 2048;
    CHECK(hipMalloc(&d_out, maxthreads * 8));
    CHECK(hipMalloc(&d_cyc, maxthreads / 64 * 8));
    Entry entries[] = {
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
        {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_fma_f64", k_fma_f64, 1},
        {"v_fma_f32", k_fma_f32, 1}, {"v_add_u32", k_add_u32, 1}, {"v_add_co+v_addc (pair)", k_add_co_addc_pair, 2},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_mov_b32+v_add (pair)", k_mov_plus_add, 2}, {"v_alignbit_b32", k_alignbit_b32, 1},
        {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_and_or_b32", k_and_or_b32, 1}, {"v_cndmask_b32", k_cndmask_b32, 1},
    };
    printf("%-28s %12s %12s %12s %12s   (cycles per wave-instruction per SIMD; last col: aggregate Ginstr-lanes/s at 4 waves/SIMD)\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
    for (auto &e : entries) {
        printf("%-28s", e.name);
        double agg = 0;
        for (int wps : {1, 2, 4, 8}) {
            int threads = 256 * wps;  // 4 SIMDs x wps waves
            int blocks_per_cu = 1;
            if (threads > 1024) { blocks_per_cu = threads / 1024; threads = 1024; }
            int grid = ncu * blocks_per_cu;
            hipEvent_t ea, eb;
            CHECK(hipEventCreate(&ea)); CHECK(hipEventCreate(&eb));
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(threads), 0, 0, d_out, d_cyc, 1u);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(ea));
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(threads), 0, 0, d_out, d_cyc, 2u);
            CHECK(hipEventRecord(eb));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, ea, eb));
            size_t nw = (size_t)grid * threads / 64;
            std::vector<uint64_t> cyc(nw);
            CHECK(hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
            std::sort(cyc.begin(), cyc.end());
            double med = (double)cyc[nw / 2];
            double instr_per_wave = (double)ITERS * CHAINS * e.instr_per_chain;
            // a SIMD hosts wps waves concurrently: cycles per wave-instruction per SIMD = elapsed / (instr * wps)
            printf(" %12.2f", med / (instr_per_wave * wps));
            if (wps == 4) agg = instr_per_wave * 64.0 * nw / (ms * 1e-3) / 1e9;
            CHECK(hipEventDestroy(ea)); CHECK(hipEventDestroy(eb));
        }
        printf("   %10.1f\n", agg);
    }
    return 0;
}
