// Device side of the R1CS -> QAP witness map that feeds the Groth16 H-query MSM.
//
// Replaces reductions::r1cs_to_qap<F>::witness_map (zk/snark/reductions/r1cs_to_qap.hpp:219-325, with
// d1 = d2 = d3 = 0 as r1cs_gg_ppzksnark_prover::process calls it, prover.hpp:79-83):
//     aA[i] = <A_i, z>, aB[i] = <B_i, z>, aC[i] = <C_i, z>          (:245-248, :288-291)  sparse mat-vec
//     aA[M + i] = z_i for i <= n (input-consistency rows)             (:240-243)
//     inverse_fft; multiply_by_coset(g); fft    on aA, aB, aC         (:250-276, :293-299)  6 NTTs, batched x3
//     H_tmp = (aA o aB - aC) / Z on the coset (divide_by_z_on_coset)  (:283-308)            one pointwise pass
//     inverse_fft; multiply_by_coset(g^-1)                            (:310-315)            1 NTT
//     coefficients_for_H = H_tmp | 0                                  (m + 1 entries)
// z = (1, primary, auxiliary) and all vectors stay in HBM; the result is consumed in place by zkhip_msm_dev.
// The transforms run over the domain make_evaluation_domain(M + n + 1) picks (:229-230) -- basic, extended or step radix-2,
// domain.hip -- with every vector held split in (part 0, part 1) of that domain.
#include <algorithm>
#include <cstring>

#include "ctx.hpp"
#include "domain.hpp"
#include "fu.hpp"

using namespace zkhip;

struct zkhip_r1cs {
    int curve;
    size_t M, n, N, m;
    int dkind = ZKHIP_DOMAIN_BASIC_RADIX2;  // the evaluation domain's kind and split (domain.hpp): m = n0 + n1
    size_t n0 = 0, n1 = 0;
    // CSR, three matrices; coefficients in Montgomery form of the lazy Fr type (SL words each)
    uint32_t *rowptr[3] = {nullptr, nullptr, nullptr};
    uint32_t *col[3] = {nullptr, nullptr, nullptr};
    uint32_t *coeff[3] = {nullptr, nullptr, nullptr};
    uint32_t *long_rows[3] = {nullptr, nullptr, nullptr};  // rows with more than LONG_ROW terms
    uint32_t n_long[3] = {0, 0, 0};
    size_t nnz[3] = {0, 0, 0};
    size_t long_terms[3] = {0, 0, 0};  // total terms in long rows
};

static constexpr uint32_t LONG_ROW = 64;

// one vector of m = n0 + n1 elements held split: element r lives at p0 + 8 r (r < n0) or p1 + 8 (r - n0)
struct SplitVec {
    uint32_t *p0, *p1;
    uint32_t n0;
    ZK_D uint32_t *at(uint32_t r) const { return r < n0 ? p0 + (size_t)r * 8 : p1 + (size_t)(r - n0) * 8; }
};

// value < 2p (product output) -> [0, p)
template <class U>
ZK_D Fu<U> fr_reduce(const Fu<U> &a) { return fu_cond_sub_p(a); }
// a, b in [0, p) -> a + b in [0, p)
template <class U>
ZK_D Fu<U> fr_add_mod(const Fu<U> &a, const Fu<U> &b) { return fu_cond_sub_p(fu_add(a, b)); }

template <class U>
__global__ __launch_bounds__(256) void r1cs_coeff_to_mont(const uint32_t *__restrict__ canon, size_t nnz, uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    fu_store<U>(out + i * U::SL, fu_cond_sub_p(fu_from_canonical<U>(canon + i * U::NL)));
}

// out[row] = <row, z> for short rows (one lane per row); long rows are left to r1cs_eval_long.
// z is canonical (NL words per element); coefficients are Montgomery, so mul(unpack(z), c) = z*c canonical.
template <class U>
__global__ __launch_bounds__(256) void r1cs_eval_rows(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                      const uint32_t *__restrict__ coeff, const uint32_t *__restrict__ z, uint32_t M,
                                                      SplitVec out) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    uint32_t lo = rowptr[row], hi = rowptr[row + 1];
    if (hi - lo > LONG_ROW) return;
    Fu<U> acc = Fu<U>::zero();
    for (uint32_t k = lo; k < hi; ++k) {
        Fu<U> t = fu_mul(fu_unpack<U>(z + (size_t)col[k] * U::NL), fu_load<U>(coeff + (size_t)k * U::SL));
        acc = fr_add_mod(acc, fr_reduce(t));
    }
    fu_pack<U>(out.at(row), acc);
}

// one workgroup per long row: strided partial sums, LDS tree
template <class U>
__global__ __launch_bounds__(256) void r1cs_eval_long(const uint32_t *__restrict__ long_rows, const uint32_t *__restrict__ rowptr,
                                                      const uint32_t *__restrict__ col, const uint32_t *__restrict__ coeff,
                                                      const uint32_t *__restrict__ z, uint32_t *__restrict__ out) {
    __shared__ uint32_t part[256 * U::L];
    const uint32_t row = long_rows[blockIdx.x], t = threadIdx.x;
    // blockIdx.y selects one of gridDim.y equal slices of the row; `out` then holds per-slice partial sums
    const uint32_t rlo = rowptr[row], rhi = rowptr[row + 1];
    const uint32_t per = (rhi - rlo + gridDim.y - 1) / gridDim.y;
    const uint32_t lo = min(rhi, rlo + blockIdx.y * per), hi = min(rhi, lo + per);
    Fu<U> acc = Fu<U>::zero();
    for (uint32_t k = lo + t; k < hi; k += 256) {
        Fu<U> x = fu_mul(fu_unpack<U>(z + (size_t)col[k] * U::NL), fu_load<U>(coeff + (size_t)k * U::SL));
        acc = fr_add_mod(acc, fr_reduce(x));
    }
#pragma unroll
    for (int i = 0; i < U::L; ++i) part[i * 256 + t] = acc.v[i];
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        if (t < d) {
            Fu<U> o;
#pragma unroll
            for (int i = 0; i < U::L; ++i) o.v[i] = part[i * 256 + t + d];
            acc = fr_add_mod(acc, o);
#pragma unroll
            for (int i = 0; i < U::L; ++i) part[i * 256 + t] = acc.v[i];
        }
        __syncthreads();
    }
    if (t == 0) fu_store<U>(out + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * U::SL, acc);
}

// out[row] = sum of the row's `nslice` partial sums
template <class U>
__global__ __launch_bounds__(64) void r1cs_long_combine(const uint32_t *__restrict__ long_rows, uint32_t n_long, uint32_t nslice,
                                                        const uint32_t *__restrict__ partial, SplitVec out) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_long) return;
    Fu<U> acc = Fu<U>::zero();
    for (uint32_t s = 0; s < nslice; ++s) acc = fr_add_mod(acc, fu_load<U>(partial + ((size_t)j * nslice + s) * U::SL));
    fu_pack<U>(out.at(long_rows[j]), acc);
}

// rows M .. m-1 of the three vectors: aA[M + i] = z_i (i <= n), everything else zero
template <class U>
__global__ __launch_bounds__(256) void r1cs_fill_tail(uint32_t *__restrict__ p0, uint32_t *__restrict__ p1, uint32_t n0, uint32_t n1,
                                                      const uint32_t *__restrict__ z, uint32_t M, uint32_t n, uint32_t m) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // index in [M, m) x 3
    uint32_t tail = m - M;
    if (i >= 3 * tail) return;
    uint32_t mat = i / tail, r = M + i % tail;
    const SplitVec v{p0 + (size_t)mat * n0 * 8, p1 + (size_t)mat * n1 * 8, n0};
    uint4 *dst = reinterpret_cast<uint4 *>(v.at(r));
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (mat == 0 && r - M <= n) {
        const uint4 *src = reinterpret_cast<const uint4 *>(z + (size_t)(r - M) * U::NL);
        a = src[0];
        b = src[1];
    }
    dst[0] = a;
    dst[1] = b;
}

// h[i] = (a[i] b[i] - c[i]) / Z(g x_i) (divide_by_z_on_coset), canonical in and out.  The three vectors are split like the
// domain (part 0: 3 x n0, part 1: 3 x n1); zinv: nz Montgomery entries for part 0 (entry i mod nz), then the one of part 1; h is
// contiguous (part 1 follows part 0).
template <class U>
__global__ __launch_bounds__(256) void groth16_h_pointwise(const uint32_t *__restrict__ p0, const uint32_t *__restrict__ p1, uint32_t n0, uint32_t n1,
                                                           const uint32_t *__restrict__ zinv, uint32_t nz, uint32_t *__restrict__ h) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n0 + n1) return;
    const bool lo = e < n0;
    const uint32_t *base = lo ? p0 : p1;
    const size_t i = lo ? e : e - n0, stride = lo ? n0 : n1;
    Fu<U> a = fu_unpack<U>(base + i * U::NL), b = fu_unpack<U>(base + (stride + i) * U::NL), c = fu_unpack<U>(base + (2 * stride + i) * U::NL);
    Fu<U> k = fu_unpack<U>(zinv + (size_t)(lo ? e & (nz - 1) : nz) * 8);
    Fu<U> ab = fu_mul(fu_mul(a, Fu<U>::r2()), b);  // a b as an integer mod p (< 2p)
    Fu<U> r = fu_mul(fu_sub<2>(ab, c), k);         // (a b - c) zinv       (< 2p)
    fu_pack<U>(h + (size_t)e * U::NL, fu_cond_sub_p(r));
}

// dst[j] = src[idx[j]] on 32-byte elements: the scalar side of a sparse query (B_query.indices)
// An index >= src_count (a malformed key) yields the zero scalar and raises ZK_STATUS_GATHER_RANGE.
__global__ __launch_bounds__(256) void fr_gather(const uint4 *__restrict__ src, size_t src_count, const uint32_t *__restrict__ idx, size_t count,
                                                 uint4 *__restrict__ dst, uint32_t *__restrict__ status) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    size_t s = idx[j];
    if (s >= src_count) {
        dst[2 * j] = dst[2 * j + 1] = make_uint4(0, 0, 0, 0);
        atomicOr(status, ZK_STATUS_GATHER_RANGE);
        return;
    }
    dst[2 * j] = src[2 * s];
    dst[2 * j + 1] = src[2 * s + 1];
}

template <class U>
__global__ void fr_zero_one(uint32_t *__restrict__ p) {
    if (blockIdx.x != 0 || threadIdx.x >= U::NL) return;
    p[threadIdx.x] = 0;
}

// ---- host side ------------------------------------------------------------------------------------
template <class U>
static int r1cs_upload_t(zkhip_ctx *ctx, zkhip_r1cs *r, const uint32_t *const rowptr[3], const uint32_t *const col[3],
                         const uint64_t *const coeff[3]) {
    for (int k = 0; k < 3; ++k) {
        size_t nnz = rowptr[k][r->M];
        r->nnz[k] = nnz;
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&r->rowptr[k], (r->M + 1) * 4));
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&r->col[k], std::max<size_t>(1, nnz) * 4));
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&r->coeff[k], std::max<size_t>(1, nnz) * U::SL * 4));
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(r->rowptr[k], rowptr[k], (r->M + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        std::vector<uint32_t> lr;
        for (size_t i = 0; i < r->M; ++i) {
            if (rowptr[k][i + 1] - rowptr[k][i] > LONG_ROW) {
                lr.push_back((uint32_t)i);
                r->long_terms[k] += rowptr[k][i + 1] - rowptr[k][i];
            }
            if (rowptr[k][i + 1] < rowptr[k][i]) return ZKHIP_ERR_INVALID;
        }
        for (size_t j = 0; j < nnz; ++j)
            if (col[k][j] > r->N) return ZKHIP_ERR_RANGE;
        r->n_long[k] = (uint32_t)lr.size();
        if (!lr.empty()) {
            ZK_HIP_CHECK(ctx, hipMalloc((void **)&r->long_rows[k], lr.size() * 4));
            ZK_HIP_CHECK(ctx, hipMemcpyAsync(r->long_rows[k], lr.data(), lr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        if (nnz) {
            uint32_t *d_c = nullptr;
            ZK_HIP_CHECK(ctx, hipMalloc((void **)&d_c, nnz * 32));
            ZK_HIP_CHECK(ctx, hipMemcpyAsync(r->col[k], col[k], nnz * 4, hipMemcpyHostToDevice, ctx->stream));
            ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_c, coeff[k], nnz * 32, hipMemcpyHostToDevice, ctx->stream));
            ZK_LAUNCH(ctx, "r1cs_coeff_to_mont", r1cs_coeff_to_mont<U>, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, d_c, nnz, r->coeff[k]);
            ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(d_c);
        }
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

template <class U>
static int witness_h_t(zkhip_ctx *ctx, const zkhip_r1cs *r, const uint32_t *d_z, const ZkDomain &dom, const uint64_t *coset, uint32_t *d_h,
                       uint32_t *d_scratch) {
    const uint32_t M = (uint32_t)r->M, m = (uint32_t)r->m, n0 = (uint32_t)dom.n0, n1 = (uint32_t)dom.n1;
    // scratch: the three vectors split like the domain (3 x n0, then 3 x n1), then the domain's own scratch
    uint32_t *p0 = d_scratch, *p1 = p0 + (size_t)3 * n0 * 8, *dscr = p1 + (size_t)3 * n1 * 8;
    for (int k = 0; k < 3; ++k) {
        const SplitVec out{p0 + (size_t)k * n0 * 8, p1 + (size_t)k * n1 * 8, n0};
        ZK_LAUNCH(ctx, "r1cs_eval_rows", r1cs_eval_rows<U>, dim3((M + 255) / 256), dim3(256), 0, r->rowptr[k], r->col[k], r->coeff[k], d_z, M, out);
        if (r->n_long[k]) {
            // slices of ~4096 terms, at most 64 per row; per-slice partial sums live in the context workspace
            uint32_t nslice = (uint32_t)std::max<size_t>(1, std::min<size_t>(64, r->long_terms[k] / ((size_t)r->n_long[k] * 4096)));
            ZK_TRY(ctx->ws_reserve((size_t)r->n_long[k] * nslice * U::SL * 4 + 256));
            ctx->ws_reset();
            uint32_t *partial = ctx->ws_take<uint32_t>((size_t)r->n_long[k] * nslice * U::SL);
            ZK_LAUNCH(ctx, "r1cs_eval_long", r1cs_eval_long<U>, dim3(r->n_long[k], nslice), dim3(256), 0, r->long_rows[k], r->rowptr[k], r->col[k],
                      r->coeff[k], d_z, partial);
            ZK_LAUNCH(ctx, "r1cs_eval_long", r1cs_long_combine<U>, dim3((r->n_long[k] + 63) / 64), dim3(64), 0, r->long_rows[k], r->n_long[k], nslice,
                      partial, out);
        }
    }
    if (m > M) ZK_LAUNCH(ctx, "r1cs_fill_tail", r1cs_fill_tail<U>, dim3((3 * (m - M) + 255) / 256), dim3(256), 0, p0, p1, n0, n1, d_z, M, (uint32_t)r->n, m);
    // coefficients, then evaluations on the coset g * domain
    ZK_TRY(zk_dom_fft_split(ctx, r->curve, dom, p0, p1, 3, 1, nullptr, dscr));
    ZK_TRY(zk_dom_fft_split(ctx, r->curve, dom, p0, p1, 3, 0, coset, dscr));
    // 1 / Z on the coset: a table the domain caches per coset generator
    const uint32_t *zinv = nullptr;
    size_t nz = 1;
    ZK_TRY(zk_dom_zinv(ctx, r->curve, dom, coset, &zinv, &nz));
    ZK_LAUNCH(ctx, "groth16_h_pointwise", groth16_h_pointwise<U>, dim3((m + 255) / 256), dim3(256), 0, p0, p1, n0, n1, zinv, (uint32_t)nz, d_h);
    ZK_TRY(zk_dom_fft_split(ctx, r->curve, dom, d_h, d_h + (size_t)n0 * 8, 1, 1, coset, dscr));
    ZK_LAUNCH(ctx, "fr_zero_one", fr_zero_one<U>, dim3(1), dim3(64), 0, d_h + (size_t)m * U::NL);
    return 0;
}

static int r1cs_shape(zkhip_r1cs *r, int kind, size_t m) {
    zkhip_domain d;
    memset(&d, 0, sizeof(d));
    d.kind = kind;
    d.m = m;
    ZkDomain z;
    ZK_TRY(zk_dom_parse(r->curve, &d, &z));
    if (m < r->M + r->n + 1 || m >= ((size_t)1 << 32)) return ZKHIP_ERR_RANGE;
    r->dkind = kind;
    r->m = m;
    r->n0 = z.n0;
    r->n1 = z.n1;
    return ZKHIP_OK;
}

extern "C" {

int zkhip_r1cs_upload(zkhip_ctx *ctx, int curve, size_t num_constraints, size_t num_inputs, size_t num_variables, const uint32_t *rowptr_a,
                      const uint32_t *col_a, const uint64_t *coeff_a, const uint32_t *rowptr_b, const uint32_t *col_b, const uint64_t *coeff_b,
                      const uint32_t *rowptr_c, const uint32_t *col_c, const uint64_t *coeff_c, zkhip_r1cs **out) {
    if (!ctx || !out || !rowptr_a || !rowptr_b || !rowptr_c) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (num_inputs > num_variables || num_constraints == 0 || num_constraints >= (1ull << 31)) return ZKHIP_ERR_RANGE;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    zkhip_r1cs *r = new zkhip_r1cs();
    r->curve = curve;
    r->M = num_constraints;
    r->n = num_inputs;
    r->N = num_variables;
    // make_evaluation_domain(num_constraints + num_inputs + 1), r1cs_to_qap.hpp:229-230: the radix-2 family's selection
    // (basic, extended, step at that size, then at big + rounded_small); zkhip_r1cs_set_domain overrides it
    {
        int kind = 0;
        size_t m = 0;
        if (!zk_dom_choice(r->M + r->n + 1, (size_t)zk_dom_two_adicity(curve), &kind, &m) || r1cs_shape(r, kind, m) != ZKHIP_OK) {
            delete r;
            return ZKHIP_ERR_RANGE;
        }
    }
    const uint32_t *rp[3] = {rowptr_a, rowptr_b, rowptr_c}, *cl[3] = {col_a, col_b, col_c};
    const uint64_t *cf[3] = {coeff_a, coeff_b, coeff_c};
    int rc = curve == CURVE_BLS12_381 ? r1cs_upload_t<BlsFrU>(ctx, r, rp, cl, cf) : r1cs_upload_t<BnFrU>(ctx, r, rp, cl, cf);
    if (rc) {
        zkhip_r1cs_free(ctx, r);
        return rc;
    }
    *out = r;
    return ZKHIP_OK;
}

void zkhip_r1cs_free(zkhip_ctx *ctx, zkhip_r1cs *r) {
    if (!r) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (int k = 0; k < 3; ++k) {
        (void)hipFree(r->rowptr[k]);
        (void)hipFree(r->col[k]);
        (void)hipFree(r->coeff[k]);
        (void)hipFree(r->long_rows[k]);
    }
    delete r;
}

int zkhip_fr_gather_dev(zkhip_ctx *ctx, const void *d_src, size_t src_count, const void *d_indices, size_t count, void *d_dst) {
    if (!ctx || (count && (!d_src || !d_indices || !d_dst))) return ZKHIP_ERR_INVALID;
    if (count == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ZK_LAUNCH(ctx, "fr_gather", fr_gather, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (const uint4 *)d_src, src_count,
              (const uint32_t *)d_indices, count, (uint4 *)d_dst, ctx->d_status);
    return ZKHIP_OK;
}

size_t zkhip_r1cs_domain_size(const zkhip_r1cs *r) { return r ? r->m : 0; }
int zkhip_r1cs_domain_kind(const zkhip_r1cs *r) { return r ? r->dkind : ZKHIP_ERR_INVALID; }
int zkhip_r1cs_set_domain(zkhip_r1cs *r, int kind, size_t m) {
    if (!r) return ZKHIP_ERR_INVALID;
    return r1cs_shape(r, kind, m);
}

size_t zkhip_groth16_scratch_bytes(const zkhip_r1cs *r) {
    if (!r) return 0;
    ZkDomain d;
    d.kind = r->dkind;
    d.m = r->m;
    d.n0 = r->n0;
    d.n1 = r->n1;
    return ((size_t)3 * r->m + zk_dom_scratch_elems(d, 3)) * 32 + 4096;
}

int zkhip_groth16_witness_h_domain_dev(zkhip_ctx *ctx, const zkhip_r1cs *r, const void *d_assignment, const zkhip_domain *dom,
                                       const uint64_t *coset_gen, void *d_h, void *d_scratch) {
    if (!ctx || !r || !d_assignment || !dom || !coset_gen || !d_h || !d_scratch) return ZKHIP_ERR_INVALID;
    ZkDomain d;
    ZK_TRY(zk_dom_parse(r->curve, dom, &d));
    if (d.kind != r->dkind || d.m != r->m) {
        ctx->last_error = "the domain passed differs from the constraint system's (zkhip_r1cs_set_domain)";
        return ZKHIP_ERR_INVALID;
    }
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (r->curve == CURVE_BLS12_381)
        return witness_h_t<BlsFrU>(ctx, r, (const uint32_t *)d_assignment, d, coset_gen, (uint32_t *)d_h, (uint32_t *)d_scratch);
    return witness_h_t<BnFrU>(ctx, r, (const uint32_t *)d_assignment, d, coset_gen, (uint32_t *)d_h, (uint32_t *)d_scratch);
}

int zkhip_groth16_witness_h_dev(zkhip_ctx *ctx, const zkhip_r1cs *r, const void *d_assignment, const uint64_t *omega, const uint64_t *coset_gen,
                                void *d_h, void *d_scratch) {
    if (!ctx || !r || !omega) return ZKHIP_ERR_INVALID;
    if (r->dkind == ZKHIP_DOMAIN_EXTENDED_RADIX2) {
        ctx->last_error = "an extended radix-2 domain needs its shift: use zkhip_groth16_witness_h_domain_dev";
        return ZKHIP_ERR_INVALID;
    }
    zkhip_domain d;
    memset(&d, 0, sizeof(d));
    d.kind = r->dkind;
    d.m = r->m;
    memcpy(d.omega, omega, 32);
    return zkhip_groth16_witness_h_domain_dev(ctx, r, d_assignment, &d, coset_gen, d_h, d_scratch);
}

}  // extern "C"
