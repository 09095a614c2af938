// Scalar -> signed-digit recoding shared by msm.hip and the CPU test shim.
#pragma once
#include "zk_defs.hpp"

namespace zkhip {

static constexpr uint32_t DIG_NONE = 0xFFFFFFFFu;

// bits [lo, lo+c) of a 256-bit little-endian integer held as 8 u32 in global memory (c <= 16)
ZK_HD uint32_t scalar_bits(const uint32_t *s, int lo, int c) {
    int limb = lo >> 5, off = lo & 31;
    if (limb >= 8) return 0;
    uint64_t v = s[limb];
    if (limb + 1 < 8) v |= (uint64_t)s[limb + 1] << 32;
    return (uint32_t)(v >> off) & ((1u << c) - 1);
}

// Signed c-bit digit of window w: returns (|d| - 1) | sign << 31, or DIG_NONE for d = 0; `carry` threads
// through the windows from w = 0 upwards.  d = bits + carry in [0, 2^c]; values above B = 2^(c-1) are
// replaced by d - 2^c (carry 1), so every non-zero digit has magnitude in [1, B].
ZK_HD uint32_t msm_recode(const uint32_t *s, int w, int c, uint32_t &carry) {
    const uint32_t B = 1u << (c - 1);
    uint32_t d = scalar_bits(s, w * c, c) + carry;
    uint32_t neg = 0;
    carry = 0;
    if (d > B) {
        d = (1u << c) - d;
        neg = d != 0 ? 1u : 0u;
        carry = 1;
    }
    return d == 0 ? DIG_NONE : ((d - 1) | (neg << 31));
}


}  // namespace zkhip
