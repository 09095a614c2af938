// Pippenger MSM instantiated for one (curve, group): see msm_core.hpp (kernels + per-call logic) and msm.hip (dispatch).
// The pair-split Fq2 products INLINE: with the Montgomery products as single asm blocks (mont_asm.hpp) the inlined bucket kernel needs
// 199 VGPRs and no scratch, and measures 6.54 against 7.35 ms per 2^20-point accumulation with the products out of line (round 4;
// round 1 had it the other way round, when the C++ products overflowed the instruction cache).
#define ZK_PAIR_INLINE 1
#include "msm_core.hpp"

const MsmOps *zk_msm_ops_bn_g2() { return msm_make_ops<zkhip::CurveTraits<zkhip::CURVE_BN254, zkhip::GROUP_G2>::F>(); }
