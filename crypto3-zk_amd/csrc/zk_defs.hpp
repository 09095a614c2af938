// Common definitions for the zkhip device code (gfx950 / CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__

namespace zkhip {

// ids shared with include/zkhip.h
enum : int { CURVE_BLS12_381 = 0, CURVE_BN254 = 1 };
enum : int { GROUP_G1 = 1, GROUP_G2 = 2 };

}  // namespace zkhip
