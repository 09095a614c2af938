// Round 4 re-base of the VALU issue roofline: WALL-CLOCK instruction rates of the integer instructions the
// field arithmetic is made of, at exactly w = 1 ... 8 waves per SIMD (one-wave-per-SIMD workgroups, dynamic LDS sized so that
// w of them fit a CU).  tools/microbench.hip (round 1) counted s_memtime ticks, which are NOT shader cycles on gfx950; this one
// reports ns per wave-instruction per SIMD and the tick rate next to it.  bench.py's issue fractions are priced with this table.
// Build: make -C tools microbench2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
#define ITERS 4096
#define CHAINS 8

#define KERNEL_BEGIN(name)                                                                                      \
    __global__ __launch_bounds__(256) void name(uint64_t *out, uint64_t *cycles, uint32_t seed) {                \
        extern __shared__ uint32_t lds_dummy[];                                                                   \
        uint32_t a = seed * 2654435761u + threadIdx.x, b = a ^ 0x9e3779b9u;                                       \
        uint64_t acc[CHAINS];                                                                                     \
        for (int i = 0; i < CHAINS; ++i) acc[i] = ((uint64_t)(a + i) << 32) | (b + i);                             \
        uint64_t t0, t1;                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");        \
        for (int it = 0; it < ITERS; ++it) {
#define KERNEL_END                                                                                               \
        }                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");        \
        uint64_t s = 0;                                                                                           \
        for (int i = 0; i < CHAINS; ++i) s ^= acc[i];                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                           \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;              \
        if (seed == 0x7fffffffu) lds_dummy[threadIdx.x] = 1;                                                      \
    }
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define OP_MADU(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
KERNEL_BEGIN(k_mad_u64_u32) REP8(OP_MADU) KERNEL_END
#define OP_MADI(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
KERNEL_BEGIN(k_mad_i64_i32) REP8(OP_MADI) KERNEL_END
#define OP_MADS(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(seed) : "vcc");
KERNEL_BEGIN(k_mad_u64_u32_sgpr) REP8(OP_MADS) KERNEL_END
#define OP_MULLO(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
KERNEL_BEGIN(k_mul_lo_u32) REP8(OP_MULLO) KERNEL_END
#define OP_ADD32(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[i] = lo; }
KERNEL_BEGIN(k_add_u32_vop2) REP8(OP_ADD32) KERNEL_END
#define OP_AND32(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(lo)); acc[i] = lo + a; }
KERNEL_BEGIN(k_and_lit_plus_add) REP8(OP_AND32) KERNEL_END
#define OP_LSHR64(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
KERNEL_BEGIN(k_lshrrev_b64) REP8(OP_LSHR64) KERNEL_END
#define OP_ASHR64(i) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(acc[i]));
KERNEL_BEGIN(k_ashrrev_i64) REP8(OP_ASHR64) KERNEL_END
#define OP_BFEI(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_bfe_i32 %0, %0, 0, 30" : "+v"(lo)); acc[i] = lo; }
KERNEL_BEGIN(k_bfe_i32) REP8(OP_BFEI) KERNEL_END
#define OP_LSHLADD64(i) { uint64_t k = ((uint64_t)b << 32) | a; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(k)); }
KERNEL_BEGIN(k_lshl_add_u64) REP8(OP_LSHLADD64) KERNEL_END
#define OP_ADD3(i) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
KERNEL_BEGIN(k_add3_u32) REP8(OP_ADD3) KERNEL_END
// the accumulation kernel's mix: 12 multiply-adds : 1 64-bit shift : 1 64-bit add : 2 VOP2
#define OP_MIX(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
#define OP_MIXT(i) asm volatile("v_lshrrev_b64 %0, 1, %0\n\tv_and_b32 %1, 0x1fffffff, %1" : "+v"(acc[i]), "+v"(a));
KERNEL_BEGIN(k_mix_24mad_8shift_8and) REP8(OP_MIX) REP8(OP_MIXT) KERNEL_END

typedef void (*kern_t)(uint64_t *, uint64_t *, uint32_t);
struct Entry { const char *name; kern_t k; int instr_per_iter; };

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, CUs %d, clockRate %d kHz\n", prop.name, ncu, prop.clockRate);
    uint64_t *d_out, *d_cyc;
    CK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 8)); CK(hipMalloc(&d_cyc, (size_t)ncu * 8 * 4 * 8));
    Entry entries[] = {
        {"v_mad_u64_u32", k_mad_u64_u32, 8}, {"v_mad_i64_i32", k_mad_i64_i32, 8}, {"v_mad_u64_u32 (sgpr src)", k_mad_u64_u32_sgpr, 8},
        {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_add_u32 (VOP2)", k_add_u32_vop2, 8}, {"v_and_b32 lit + v_add (2 VOP2)", k_and_lit_plus_add, 16},
        {"v_lshrrev_b64", k_lshrrev_b64, 8}, {"v_ashrrev_i64", k_ashrrev_i64, 8}, {"v_bfe_i32", k_bfe_i32, 8},
        {"v_lshl_add_u64", k_lshl_add_u64, 8}, {"v_add3_u32", k_add3_u32, 8}, {"mix 24 mad + 8 shr64 + 8 and", k_mix_24mad_8shift_8and, 40},
    };
    printf("ns per wave-instruction per SIMD at w waves per SIMD (wall clock); last column: s_memtime tick rate, GHz\n%-34s", "instruction");
    for (int w = 1; w <= 8; ++w) printf("   w=%d  ", w);
    printf("  tick GHz\n");
    for (auto &e : entries) {
        CK(hipFuncSetAttribute((const void *)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        printf("%-34s", e.name);
        double tick = 0;
        for (int w = 1; w <= 8; ++w) {
            size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)1023; if (w == 1) lds = 96 * 1024;
            hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
            hipLaunchKernelGGL(e.k, dim3(ncu * w), dim3(256), lds, 0, d_out, d_cyc, 1u);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ea));
            hipLaunchKernelGGL(e.k, dim3(ncu * w), dim3(256), lds, 0, d_out, d_cyc, 2u);
            CK(hipEventRecord(eb));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, ea, eb));
            std::vector<uint64_t> cyc((size_t)ncu * w * 4);
            CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
            std::sort(cyc.begin(), cyc.end());
            const double instr = (double)ITERS * e.instr_per_iter;
            printf(" %7.3f ", ms * 1e6 / (instr * w));
            tick = (double)cyc[cyc.size() / 2] / (ms * 1e6);
            CK(hipEventDestroy(ea)); CK(hipEventDestroy(eb));
        }
        printf("  %.3f\n", tick);
    }
    return 0;
}
