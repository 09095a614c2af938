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

// Window geometry: the tb = bitlen(r) scalar bits are cut into W = ceil(tb / c) windows of NEARLY EQUAL width
// (floor or ceil of tb / W, never more than c): window w covers bits [off(w), off(w + 1)).  Cutting at multiples of c
// instead leaves a top window with whatever remains -- 2 bits for c = 12 or 14 -- whose handful of buckets each
// receive a quarter of all points.
struct MsmWindows {
    int tb, W;
    uint8_t offs[64];  // offs[w] = floor(w * tb / W), w = 0..W (tb <= 255, W <= 64 for c >= 4; smaller c: computed)
    ZK_HD int off(int w) const { return W < 64 ? (int)offs[w] : (int)(((long long)w * tb) / W); }
    ZK_HD int width(int w) const { return off(w + 1) - off(w); }
};
inline MsmWindows msm_make_windows(int tb, int W) {
    MsmWindows m;
    m.tb = tb;
    m.W = W;
    for (int w = 0; w < 64; ++w) m.offs[w] = (uint8_t)(w <= W ? ((long long)w * tb) / W : 0);
    return m;
}

// Signed digit of a window of `width` bits at bit offset `off`: returns (|d| - 1) | sign << 31, or DIG_NONE for
// d = 0; `carry` threads through the windows from the lowest upwards.  d = bits + carry in [0, 2^width]; values
// above 2^(width-1) are replaced by d - 2^width (carry 1), so every non-zero digit has magnitude in [1, 2^(width-1)].
ZK_HD uint32_t msm_recode(const uint32_t *s, int off, int width, uint32_t &carry) {
    const uint32_t B = 1u << (width - 1);
    uint32_t d = scalar_bits(s, off, width) + carry;
    uint32_t neg = 0;
    carry = 0;
    if (d > B) {
        d = (1u << width) - d;
        neg = d != 0 ? 1u : 0u;
        carry = 1;
    }
    return d == 0 ? DIG_NONE : ((d - 1) | (neg << 31));
}

// Fold a scalar s (reduced mod r first when it is not canonical) into the symmetric range: when 2 s > r it is replaced by r - s and the caller
// flips the sign of every digit (s P = (r - s)(-P)).  |s| <= (r - 1) / 2 < 2^(bitlen(r) - 1), so the recoding over
// MsmWindows{bitlen(r), msm_windows(bitlen(r), c)} never carries out of the top window -- without the fold a window size that
// divides bitlen(r) (c = 15 for the 255-bit BLS12-381 r) leaves a carry-only top window whose single bucket
// receives ~45 % of all points.  FR: saturated scalar-field constants (mod(i), 8 u32 limbs).
template <class FR>
ZK_HD bool msm_fold_scalar(const uint32_t *s, uint32_t out[8]) {
    uint32_t v[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = s[i];
    for (int it = 0; it < 6; ++it) {  // input that is not canonical is reduced first (2^256 < 6 r for both fields)
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t d = (uint64_t)v[i] - FR::mod(i) - borrow;
            t[i] = (uint32_t)d;
            borrow = (d >> 32) & 1;
        }
        if (borrow) break;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = t[i];
    }
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)FR::mod(i) - v[i] - borrow;
        t[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    bool fold = false, decided = false;  // fold iff v > r - v
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        if (!decided && v[i] != t[i]) {
            fold = v[i] > t[i];
            decided = true;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = fold ? t[i] : v[i];
    return fold;
}

// windows needed for folded scalars of a field with `rbits`-bit modulus
ZK_HD int msm_windows(int rbits, int c) { return (rbits + c - 1) / c; }


}  // namespace zkhip
