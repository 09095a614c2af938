// Pippenger MSM instantiated for one (curve, group): see msm_core.hpp (kernels + per-call logic) and msm.hip (dispatch).
#include "msm_core.hpp"

const MsmOps *zk_msm_ops_bn_g1() { return msm_make_ops<zkhip::CurveTraits<zkhip::CURVE_BN254, zkhip::GROUP_G1>::F>(); }
