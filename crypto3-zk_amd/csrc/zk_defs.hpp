// Common definitions for the zkhip device code (gfx950 / CDNA4).
//
// The arithmetic headers (fp.hpp, fu.hpp, curve.hpp) are __host__ __device__: hipcc builds them into the
// kernels, and the header-only C++ shim (include/nil/crypto3/zk/hip/) reuses the very same code on the host
// for the few serial group operations the reference's prover also does on the CPU (prover.hpp:141-155).
// When a host compiler without HIP includes them, the decorations vanish and a plain uint4 stands in.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#define ZK_NOINLINE_HD __host__ __device__ __noinline__
#define ZK_NOINLINE_D __device__ __noinline__
#define ZK_UNROLL _Pragma("unroll")
#else
#define ZK_HD inline
#define ZK_D inline
#define ZK_NOINLINE_HD __attribute__((noinline))
#define ZK_NOINLINE_D __attribute__((noinline))
// the limb loops must be flat for the host compiler too (gcc does not know `#pragma unroll`: a Montgomery product with its
// loops left rolled is 3x slower, and the prover's host-side scalar multiplications are on a proof's critical path)
#if defined(__clang__)
#define ZK_UNROLL _Pragma("unroll")
#elif defined(__GNUC__)
#define ZK_UNROLL _Pragma("GCC unroll 32")
#else
#define ZK_UNROLL
#endif
struct alignas(16) uint4 {
    uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
#endif

namespace zkhip {

// ids shared with include/zkhip.h
enum : int { CURVE_BLS12_381 = 0, CURVE_BN254 = 1 };
enum : int { GROUP_G1 = 1, GROUP_G2 = 2 };

}  // namespace zkhip
