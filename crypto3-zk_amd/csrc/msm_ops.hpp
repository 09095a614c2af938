// The per-(curve, group) entry points of the MSM family: one table per translation unit (msm_bls_g1.hip, ...), looked
// up by msm.hip's curve-independent host side.
#pragma once
#include "ctx.hpp"

struct MsmOps {
    int (*run)(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint32_t *d_scalars, uint32_t *d_out_jac);
    int (*batch)(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                 const uint32_t *const *d_scalars, uint32_t *const *d_outs);
    int (*precompute)(zkhip_ctx *ctx, zkhip_bases *b);
    int (*to_mont)(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_canonical, const uint8_t *d_inf);
    int (*from_mont)(zkhip_ctx *ctx, const zkhip_bases *b, size_t offset, size_t n, uint32_t *d_out, uint8_t *d_inf);
    int (*mul)(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_base_canonical, const uint32_t *d_scalars);
    int (*jac_to_affine)(zkhip_ctx *ctx, const uint32_t *d_jac, uint32_t *d_aff, uint8_t *d_inf);
    int (*jac_sum)(zkhip_ctx *ctx, const uint32_t *d_pts, size_t count, uint32_t *d_out);
    int (*write_infinity)(zkhip_ctx *ctx, uint32_t *d_out_jac);
    size_t point_words;  // u32 words per affine point in device buffers
};

const MsmOps *zk_msm_ops_bls_g1();
const MsmOps *zk_msm_ops_bls_g2();
const MsmOps *zk_msm_ops_bn_g1();
const MsmOps *zk_msm_ops_bn_g2();
const MsmOps *zk_msm_ops(int curve, int group);  // nullptr for an unknown pair

size_t zk_msm_target_lanes();  // buckets (= lanes) the accumulation kernel wants at least: decides the number of bucket sets
