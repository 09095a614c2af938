// Internal interface of domain.hip: evaluation domains of the radix-2 family (basic, extended, step) over split vectors.
#pragma once
#include <cstddef>
#include <cstdint>

#include "ctx.hpp"

struct ZkDomain {
    int kind = ZKHIP_DOMAIN_BASIC_RADIX2;
    size_t m = 0;
    size_t n0 = 0, n1 = 0;  // points in part 0 / part 1: basic (m, 0), extended (m/2, m/2), step (big, small)
    uint64_t omega[4] = {0, 0, 0, 0}, shift[4] = {0, 0, 0, 0};
};

int zk_dom_two_adicity(int curve);
// make_evaluation_domain(min_size)'s choice over a field of two-adicity s; false: none of the radix-2 family fits
bool zk_dom_choice(size_t min_size, size_t s, int *kind, size_t *m);
int zk_dom_parse(int curve, const zkhip_domain *d, ZkDomain *out);
// 32-byte elements of scratch zk_dom_fft_split needs for a batch
size_t zk_dom_scratch_elems(const ZkDomain &d, size_t batch);
// in-place transform of `batch` vectors held split: p0 = batch x n0, p1 = batch x n1 (canonical Fr, 8 words each).
// forward: coefficients (low n0, high n1) -> evaluations (part 0, part 1 of the point set), on the coset when given;
// inverse: the other way, times coset^-i.
int zk_dom_fft_split(zkhip_ctx *ctx, int curve, const ZkDomain &d, uint32_t *p0, uint32_t *p1, size_t batch, int inverse, const uint64_t *coset,
                     uint32_t *scratch);
// 1 / Z(coset x_i): *d_zinv = nz Montgomery entries for part 0 (entry i mod nz) followed by the one of part 1
int zk_dom_zinv(zkhip_ctx *ctx, int curve, const ZkDomain &d, const uint64_t *coset, const uint32_t **d_zinv, size_t *nz);
