// Pippenger MSM instantiated for one (curve, group): see msm_core.hpp (kernels + per-call logic) and msm.hip (dispatch).
#include "msm_core.hpp"

const MsmOps *zk_msm_ops_bls_g2() { return msm_make_ops<zkhip::CurveTraits<zkhip::CURVE_BLS12_381, zkhip::GROUP_G2>::F>(); }
