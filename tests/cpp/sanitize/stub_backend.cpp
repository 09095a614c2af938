// TEST INFRASTRUCTURE: a stand-in for libzkhip.so that computes NOTHING -- "device" memory is host memory, transfers are memcpy,
// every kernel entry point is a no-op that leaves zeros (the point at infinity) behind.  It exists so that the HOST side of the
// header-only shim (threads, staging buffers, parsers, ownership) can run under AddressSanitizer / UBSan / ThreadSanitizer on a box
// without a GPU (GPU sanitizers are not available on this pool).  It is NOT a CPU fallback: nothing links it outside
// tests/cpp/sanitize, and no result it produces is ever compared with anything.
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../../include/zkhip.h"

struct zkhip_ctx {
    int device;
};
struct zkhip_bases {
    int curve, group;
    size_t n;
};
struct zkhip_r1cs {
    int curve, kind;
    size_t M, n, N, m;
};

static size_t coord_limbs(int curve, int group) { return (curve == ZKHIP_BLS12_381 ? 6 : 4) * (group == ZKHIP_G2 ? 2 : 1); }
static size_t ceil_log2(size_t n) {
    size_t r = 0;
    while (((size_t)1 << r) < n) ++r;
    return r;
}

extern "C" {
int zkhip_init(int device, zkhip_ctx **out) {
    *out = new zkhip_ctx {device};
    return ZKHIP_OK;
}
void zkhip_destroy(zkhip_ctx *ctx) { delete ctx; }
const char *zkhip_strerror(int status) { return status == ZKHIP_OK ? "ok" : "stub backend error"; }
const char *zkhip_last_error(const zkhip_ctx *) { return ""; }
int zkhip_set_stream(zkhip_ctx *, void *) { return ZKHIP_OK; }
int zkhip_stream_wait(zkhip_ctx *, zkhip_ctx *) { return ZKHIP_OK; }
int zkhip_device(const zkhip_ctx *ctx) { return ctx->device; }
int zkhip_sync(zkhip_ctx *) { return ZKHIP_OK; }
int zkhip_device_status(zkhip_ctx *, uint32_t *flags) {
    if (flags) *flags = 0;
    return ZKHIP_OK;
}
int zkhip_set_option(zkhip_ctx *, const char *, int64_t) { return ZKHIP_OK; }
int zkhip_get_option(const zkhip_ctx *, const char *, int64_t *value) {
    *value = 14;
    return ZKHIP_OK;
}
int zkhip_malloc(zkhip_ctx *, size_t bytes, void **dptr) {
    *dptr = calloc(bytes ? bytes : 1, 1);
    return *dptr ? ZKHIP_OK : ZKHIP_ERR_OOM;
}
int zkhip_free(zkhip_ctx *, void *dptr) {
    free(dptr);
    return ZKHIP_OK;
}
int zkhip_memcpy_h2d(zkhip_ctx *, void *dst, const void *src, size_t bytes) {
    memcpy(dst, src, bytes);
    return ZKHIP_OK;
}
int zkhip_memcpy_d2h(zkhip_ctx *, void *dst, const void *src, size_t bytes) {
    memcpy(dst, src, bytes);
    return ZKHIP_OK;
}
int zkhip_memcpy_h2d_async(zkhip_ctx *, void *dst, const void *src, size_t bytes) {
    memcpy(dst, src, bytes);
    return ZKHIP_OK;
}
int zkhip_memcpy_d2h_async(zkhip_ctx *, void *dst, const void *src, size_t bytes) {
    memcpy(dst, src, bytes);
    return ZKHIP_OK;
}
int zkhip_memcpy_d2d_async(zkhip_ctx *, void *dst, const void *src, size_t bytes) {
    memmove(dst, src, bytes);
    return ZKHIP_OK;
}
int zkhip_memcpy_2d_d2d_async(zkhip_ctx *, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width, size_t rows) {
    if (width > dst_pitch || width > src_pitch) return ZKHIP_ERR_INVALID;
    for (size_t r = 0; r < rows; ++r) memcpy(static_cast<char *>(dst) + r * dst_pitch, static_cast<const char *>(src) + r * src_pitch, width);
    return ZKHIP_OK;
}
int zkhip_host_alloc(zkhip_ctx *, size_t bytes, void **hptr) {
    *hptr = malloc(bytes ? bytes : 1);
    return *hptr ? ZKHIP_OK : ZKHIP_ERR_OOM;
}
int zkhip_host_free(zkhip_ctx *, void *hptr) {
    free(hptr);
    return ZKHIP_OK;
}
static int new_bases(int curve, int group, size_t n, zkhip_bases **out) {
    *out = new zkhip_bases {curve, group, n};
    return ZKHIP_OK;
}
int zkhip_bases_upload(zkhip_ctx *, int curve, int group, const uint64_t *xy, const uint8_t *inf, size_t n, zkhip_bases **out) {
    // touch every input byte: the sanitizers see what the real upload would read
    volatile uint64_t acc = 0;
    for (size_t i = 0; i < n * 2 * coord_limbs(curve, group); ++i) acc += xy[i];
    if (inf)
        for (size_t i = 0; i < n; ++i) acc += inf[i];
    return new_bases(curve, group, n, out);
}
int zkhip_bases_upload_compressed(zkhip_ctx *, int curve, int group, const uint8_t *octets, size_t n, zkhip_bases **out) {
    volatile uint64_t acc = 0;
    for (size_t i = 0; i < n * (group == ZKHIP_G1 ? 48 : 96); ++i) acc += octets[i];
    return new_bases(curve, group, n, out);
}
int zkhip_bases_from_scalars(zkhip_ctx *, int curve, int group, const uint64_t *, const uint64_t *scalars, size_t n, zkhip_bases **out) {
    volatile uint64_t acc = 0;
    for (size_t i = 0; i < 4 * n; ++i) acc += scalars[i];
    return new_bases(curve, group, n, out);
}
int zkhip_bases_spread(zkhip_ctx *, const zkhip_bases *src, const uint32_t *, size_t first, size_t n_total, zkhip_bases **out) {
    if (first > n_total) return ZKHIP_ERR_RANGE;
    return new_bases(src->curve, src->group, n_total, out);
}
int zkhip_bases_download(zkhip_ctx *, const zkhip_bases *b, size_t offset, size_t n, uint64_t *xy, uint8_t *inf) {
    if (offset + n > b->n) return ZKHIP_ERR_RANGE;
    memset(xy, 0, n * 2 * coord_limbs(b->curve, b->group) * 8);
    memset(inf, 1, n);
    return ZKHIP_OK;
}
size_t zkhip_bases_size(const zkhip_bases *b) { return b->n; }
void zkhip_bases_free(zkhip_ctx *, zkhip_bases *b) { delete b; }
int zkhip_msm(zkhip_ctx *, const zkhip_bases *b, size_t offset, size_t n, const uint64_t *scalars, uint64_t *out) {
    if (offset + n > b->n) return ZKHIP_ERR_RANGE;
    volatile uint64_t acc = 0;
    for (size_t i = 0; i < 4 * n; ++i) acc += scalars[i];
    memset(out, 0, 3 * coord_limbs(b->curve, b->group) * 8);
    return ZKHIP_OK;
}
int zkhip_msm_dev(zkhip_ctx *ctx, const zkhip_bases *b, size_t offset, size_t n, const void *d_scalars, void *d_out) {
    return zkhip_msm(ctx, b, offset, n, (const uint64_t *)d_scalars, (uint64_t *)d_out);
}
int zkhip_msm_batch_dev(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns, const void *const *d_scalars,
                        void *const *d_out) {
    for (size_t i = 0; i < count; ++i) {
        int rc = zkhip_msm_dev(ctx, bases[i], offsets[i], ns[i], d_scalars[i], d_out[i]);
        if (rc) return rc;
    }
    return ZKHIP_OK;
}
int zkhip_jacobian_sum_dev(zkhip_ctx *, int curve, int group, const void *, size_t, void *d_out) {
    memset(d_out, 0, 3 * coord_limbs(curve, group) * 8);
    return ZKHIP_OK;
}
int zkhip_jacobian_to_affine(zkhip_ctx *, int curve, int group, const uint64_t *, uint64_t *xy, uint8_t *inf) {
    memset(xy, 0, 2 * coord_limbs(curve, group) * 8);
    *inf = 1;
    return ZKHIP_OK;
}
static void touch(void *d, size_t bytes) { memset(d, 0, bytes); }
int zkhip_ntt(zkhip_ctx *, int, uint64_t *data, size_t log_m, size_t batch, const uint64_t *, int, const uint64_t *) {
    touch(data, (batch << log_m) * 32);
    return ZKHIP_OK;
}
int zkhip_ntt_dev(zkhip_ctx *, int, void *d, size_t log_m, size_t batch, const uint64_t *, int, const uint64_t *) {
    touch(d, (batch << log_m) * 32);
    return ZKHIP_OK;
}
int zkhip_domain_choice(int curve, size_t min_size, int *kind, size_t *m) {
    // the radix-2 family's selection, restated once more (host arithmetic only)
    const size_t s = curve == ZKHIP_BLS12_381 ? 32 : 28;
    auto basic_ok = [&](size_t n) { return n > 1 && n == (size_t)1 << ceil_log2(n) && ceil_log2(n) <= s; };
    auto step_ok = [&](size_t n) {
        if (n <= 1) return false;
        const size_t small = n - ((size_t)1 << (ceil_log2(n) - 1));
        return small == (size_t)1 << ceil_log2(small) && ceil_log2(n) <= s;
    };
    if (min_size <= 1) return ZKHIP_ERR_RANGE;
    const size_t big = (size_t)1 << (ceil_log2(min_size) - 1), small = min_size - big, rounded = big + ((size_t)1 << ceil_log2(small));
    for (size_t n : {min_size, rounded}) {
        if (basic_ok(n)) return *kind = ZKHIP_DOMAIN_BASIC_RADIX2, *m = n, ZKHIP_OK;
        if (step_ok(n)) return *kind = ZKHIP_DOMAIN_STEP_RADIX2, *m = n, ZKHIP_OK;
    }
    return ZKHIP_ERR_RANGE;
}
int zkhip_domain_fft_dev(zkhip_ctx *, int, const zkhip_domain *dom, void *d, size_t batch, int, const uint64_t *) {
    touch(d, batch * dom->m * 32);
    return ZKHIP_OK;
}
int zkhip_domain_lagrange_dev(zkhip_ctx *, int, const zkhip_domain *dom, const uint64_t *t, void *d_out) {
    volatile uint64_t x = t[0];
    (void)x;
    uint64_t *o = static_cast<uint64_t *>(d_out);    // "L_i(t)" = i + 1: distinct non-zero values for the host code downstream
    for (size_t i = 0; i < dom->m; ++i) o[4 * i] = i + 1, o[4 * i + 1] = o[4 * i + 2] = o[4 * i + 3] = 0;
    return ZKHIP_OK;
}
int zkhip_r1cs_upload(zkhip_ctx *, int curve, size_t M, size_t n, size_t N, const uint32_t *rpa, const uint32_t *cla, const uint64_t *cfa, const uint32_t *rpb,
                      const uint32_t *clb, const uint64_t *cfb, const uint32_t *rpc, const uint32_t *clc, const uint64_t *cfc, zkhip_r1cs **out) {
    const uint32_t *rp[3] = {rpa, rpb, rpc}, *cl[3] = {cla, clb, clc};
    const uint64_t *cf[3] = {cfa, cfb, cfc};
    volatile uint64_t acc = 0;
    for (int k = 0; k < 3; ++k) {
        for (size_t i = 0; i <= M; ++i) acc += rp[k][i];
        for (size_t j = 0; j < rp[k][M]; ++j) {
            if (cl[k][j] > N) return ZKHIP_ERR_RANGE;
            acc += cf[k][4 * j] + cf[k][4 * j + 3];
        }
    }
    int kind = 0;
    size_t m = 0;
    if (zkhip_domain_choice(curve, M + n + 1, &kind, &m)) return ZKHIP_ERR_RANGE;
    *out = new zkhip_r1cs {curve, kind, M, n, N, m};
    return ZKHIP_OK;
}
void zkhip_r1cs_free(zkhip_ctx *, zkhip_r1cs *r) { delete r; }
int zkhip_r1cs_set_domain(zkhip_r1cs *r, int kind, size_t m) {
    if (m < r->M + r->n + 1) return ZKHIP_ERR_RANGE;
    r->kind = kind;
    r->m = m;
    return ZKHIP_OK;
}
size_t zkhip_r1cs_domain_size(const zkhip_r1cs *r) { return r->m; }
int zkhip_r1cs_domain_kind(const zkhip_r1cs *r) { return r->kind; }
size_t zkhip_groth16_scratch_bytes(const zkhip_r1cs *r) { return 3 * r->m * 32 + 4096; }
int zkhip_groth16_witness_h_dev(zkhip_ctx *, const zkhip_r1cs *r, const void *, const uint64_t *, const uint64_t *, void *d_h, void *d_scratch) {
    touch(d_h, (r->m + 1) * 32);
    touch(d_scratch, zkhip_groth16_scratch_bytes(r));
    return ZKHIP_OK;
}
int zkhip_groth16_witness_h_domain_dev(zkhip_ctx *ctx, const zkhip_r1cs *r, const void *a, const zkhip_domain *dom, const uint64_t *g, void *d_h,
                                       void *d_scratch) {
    if (dom->m != r->m || dom->kind != r->kind) return ZKHIP_ERR_INVALID;
    return zkhip_groth16_witness_h_dev(ctx, r, a, dom->omega, g, d_h, d_scratch);
}
int zkhip_fr_gather_dev(zkhip_ctx *, const void *d_src, size_t src_count, const void *d_indices, size_t count, void *d_dst) {
    const uint32_t *idx = (const uint32_t *)d_indices;
    for (size_t j = 0; j < count; ++j) {
        if (idx[j] >= src_count) memset((char *)d_dst + 32 * j, 0, 32);
        else memcpy((char *)d_dst + 32 * j, (const char *)d_src + 32 * (size_t)idx[j], 32);
    }
    return ZKHIP_OK;
}
int zkhip_poly_resize_dev(zkhip_ctx *, int, void *d_in, size_t log_n, size_t batch, const uint64_t *, void *d_out, size_t log_out, const uint64_t *) {
    touch(d_in, (batch << log_n) * 32);
    touch(d_out, (batch << log_out) * 32);
    return ZKHIP_OK;
}
int zkhip_poly_shift_dev(zkhip_ctx *, const void *, size_t log_size, int64_t, void *d_out) {
    touch(d_out, ((size_t)1 << log_size) * 32);
    return ZKHIP_OK;
}
int zkhip_fr_vec_prod_dev(zkhip_ctx *, int, size_t, const void *const *, void *d_out, size_t n) {
    touch(d_out, n * 32);
    return ZKHIP_OK;
}
int zkhip_fri_fold_dev(zkhip_ctx *, int, const void *, size_t log_size, const uint64_t *, const uint64_t *, void *d_out) {
    touch(d_out, ((size_t)1 << log_size) / 2 * 32);
    return ZKHIP_OK;
}
int zkhip_fri_leaves_dev(zkhip_ctx *, const void *d_polys, size_t log_domain, size_t batch, size_t, void *d_out) {
    memcpy(d_out, d_polys, (batch << log_domain) * 32);
    return ZKHIP_OK;
}
int zkhip_ec_ntt_dev(zkhip_ctx *, int, int, void *, size_t, const uint64_t *, int) { return ZKHIP_OK; }
int zkhip_fr_vec_op_dev(zkhip_ctx *, int, int, const void *d_a, const void *d_b, void *d_out, size_t count) {
    volatile uint64_t acc = 0;
    for (size_t i = 0; i < 4 * count; ++i) acc += ((const uint64_t *)d_a)[i] + ((const uint64_t *)d_b)[i];
    touch(d_out, count * 32);
    return ZKHIP_OK;
}
int zkhip_poly_eval_dev(zkhip_ctx *, int, const void *, size_t, size_t, size_t batch, const uint64_t *, size_t npoints, uint64_t *out) {
    touch(out, batch * npoints * 32);
    return ZKHIP_OK;
}
int zkhip_poly_div_linear_dev(zkhip_ctx *, int, const void *, size_t n, const uint64_t *, void *d_out, uint64_t *remainder) {
    touch(d_out, n * 32);
    if (remainder) memset(remainder, 0, 32);
    return ZKHIP_OK;
}
int zkhip_poly_div_vanishing_dev(zkhip_ctx *, int, const void *, size_t len, size_t n, void *d_quot, uint64_t *nonzero_remainders) {
    if (len > n) touch(d_quot, (len - n) * 32);
    if (nonzero_remainders) *nonzero_remainders = 0;
    return ZKHIP_OK;
}
int zkhip_perm_grand_product_dev(zkhip_ctx *, int, size_t k, const void *const *, const void *const *, const void *const *, size_t n, const uint64_t *,
                                 const uint64_t *, void *d_g, void *d_h, void *d_vp) {
    if (d_g) touch(d_g, k * n * 32);
    if (d_h) touch(d_h, k * n * 32);
    touch(d_vp, n * 32);
    return ZKHIP_OK;
}
int zkhip_perm_factor_products_dev(zkhip_ctx *, int, size_t, const void *const *, const void *const *, const void *const *, size_t n, const uint64_t *,
                                   const uint64_t *, void *d_g, void *d_h) {
    touch(d_g, n * 32);
    touch(d_h, n * 32);
    return ZKHIP_OK;
}
int zkhip_fr_vec_affine_dev(zkhip_ctx *, int, const void *, const void *, const uint64_t *, const uint64_t *, const uint64_t *, void *d_out, size_t count) {
    touch(d_out, count * 32);
    return ZKHIP_OK;
}
int zkhip_fr_vec_mul_div_dev(zkhip_ctx *, int, const void *, const void *, const void *, void *d_out, size_t count) {
    touch(d_out, count * 32);
    return ZKHIP_OK;
}
int zkhip_lookup_grand_product_dev(zkhip_ctx *, int, size_t, const void *const *, size_t, const void *const *, size_t, const void *const *, size_t n, size_t,
                                   const uint64_t *, const uint64_t *, void *d_vl) {
    touch(d_vl, n * 32);
    return ZKHIP_OK;
}
int zkhip_lookup_sort_dev(zkhip_ctx *, size_t k_in, const void *const *, size_t k_val, const void *const *, size_t n, size_t, void *const *d_sorted) {
    for (size_t i = 0; i < k_in + k_val; ++i) touch(d_sorted[i], n * 32);
    return ZKHIP_OK;
}
int zkhip_poly_lincomb_dev(zkhip_ctx *, int, size_t count, const void *const *d_polys, const size_t *lens, const uint64_t *coeffs, size_t taps, void *d_acc,
                           size_t acc_len, int) {
    volatile uint64_t acc = 0;
    for (size_t i = 0; i < count; ++i) {
        if (lens[i]) acc += ((const uint64_t *)d_polys[i])[4 * lens[i] - 1];
        for (size_t t = 0; t < 4 * taps; ++t) acc += coeffs[4 * i * taps + t];
    }
    touch(d_acc, acc_len * 32);
    return ZKHIP_OK;
}
int zkhip_gate_eval_dev(zkhip_ctx *, int, const zkhip_gate_program *prog, const void *const *d_slots, size_t log_size, const void *d_mask, int, void *d_out) {
    // walk the program the way the kernel would (every table entry and one element of every slot it names is read: the sanitizers see a
    // program whose ranges or slots run past their arrays), then leave zeros behind
    const size_t size = (size_t)1 << log_size;
    volatile uint64_t acc = 0;
    if (prog->n_gates && prog->gate_terms[prog->n_gates] != prog->n_terms) return ZKHIP_ERR_INVALID;
    if (prog->n_terms && prog->term_factors[prog->n_terms] != prog->n_factors) return ZKHIP_ERR_INVALID;
    for (uint32_t g = 0; g < prog->n_gates; ++g) {
        if (prog->gate_selector[g] != ZKHIP_GATE_NO_SELECTOR) {
            if (prog->gate_selector[g] >= prog->n_slots) return ZKHIP_ERR_RANGE;
            const size_t at = ((size_t)(int64_t)prog->gate_selector_rot[g]) & (size - 1);
            acc += static_cast<const uint64_t *>(d_slots[prog->gate_selector[g]])[4 * at];
        }
        for (uint32_t t = prog->gate_terms[g]; t < prog->gate_terms[g + 1]; ++t) {
            acc += prog->term_coeff[4 * t] + prog->term_coeff[4 * t + 3];
            for (uint32_t f = prog->term_factors[t]; f < prog->term_factors[t + 1]; ++f) {
                if (prog->factor_slot[f] >= prog->n_slots) return ZKHIP_ERR_RANGE;
                const size_t at = ((size_t)(int64_t)prog->factor_rot[f]) & (size - 1);
                acc += static_cast<const uint64_t *>(d_slots[prog->factor_slot[f]])[4 * at + 3];
            }
        }
    }
    if (d_mask) acc += static_cast<const uint64_t *>(d_mask)[4 * (size - 1)];
    touch(d_out, size * 32);
    return ZKHIP_OK;
}
// ---- device group: the members are stub contexts, the exchange is memcpy (what is exercised is the shim's bookkeeping around it)
struct zkhip_device_group_stub {
    int n;
    zkhip_ctx **members;
    int transport;
};
int zkhip_group_init(const int *device_ids, int n_dev, zkhip_device_group **out) {
    if (!device_ids || n_dev < 1) return ZKHIP_ERR_INVALID;
    auto *g = new zkhip_device_group_stub {n_dev, new zkhip_ctx *[n_dev], ZKHIP_GROUP_AUTO};
    for (int k = 0; k < n_dev; ++k) g->members[k] = new zkhip_ctx {device_ids[k]};
    *out = reinterpret_cast<zkhip_device_group *>(g);
    return ZKHIP_OK;
}
static zkhip_device_group_stub *stub(const zkhip_device_group *g) { return reinterpret_cast<zkhip_device_group_stub *>(const_cast<zkhip_device_group *>(g)); }
void zkhip_group_destroy(zkhip_device_group *g) {
    if (!g) return;
    for (int k = 0; k < stub(g)->n; ++k) delete stub(g)->members[k];
    delete[] stub(g)->members;
    delete stub(g);
}
int zkhip_group_size(const zkhip_device_group *g) { return stub(g)->n; }
zkhip_ctx *zkhip_group_ctx(const zkhip_device_group *g, int member) { return member >= 0 && member < stub(g)->n ? stub(g)->members[member] : nullptr; }
const char *zkhip_group_last_error(const zkhip_device_group *) { return ""; }
int zkhip_group_set_transport(zkhip_device_group *g, int transport) {
    stub(g)->transport = transport;
    return ZKHIP_OK;
}
int zkhip_group_transport(const zkhip_device_group *g) { return stub(g)->transport == ZKHIP_GROUP_AUTO ? ZKHIP_GROUP_PEER : stub(g)->transport; }
int zkhip_group_all_gather(zkhip_device_group *g, const void *const *d_send, void *const *d_recv, size_t bytes) {
    const int n = stub(g)->n;
    for (int j = 0; j < n; ++j)
        if (d_recv[j])
            for (int k = 0; k < n; ++k) memcpy(static_cast<char *>(d_recv[j]) + (size_t)k * bytes, d_send[k], bytes);
    return ZKHIP_OK;
}
int zkhip_group_copy(zkhip_device_group *g, int dst_member, void *d_dst, int src_member, const void *d_src, size_t bytes) {
    if (dst_member < 0 || src_member < 0 || dst_member >= stub(g)->n || src_member >= stub(g)->n) return ZKHIP_ERR_INVALID;
    memmove(d_dst, d_src, bytes);
    return ZKHIP_OK;
}
int zkhip_group_sync(zkhip_device_group *) { return ZKHIP_OK; }
struct zkhip_group_bases {
    int curve, group;
    size_t n;
    int world;
    zkhip_bases **member;
};
static size_t part_lo(size_t n, size_t k, size_t world) { return k * (n / world) + (k < n % world ? k : n % world); }
int zkhip_group_bases_upload(zkhip_device_group *g, int curve, int group, const uint64_t *xy, const uint8_t *inf, size_t n, zkhip_group_bases **out) {
    const int world = stub(g)->n;
    auto *b = new zkhip_group_bases {curve, group, n, world, new zkhip_bases *[world]};
    for (int k = 0; k < world; ++k) {
        const size_t lo = part_lo(n, k, world), cnt = part_lo(n, k + 1, world) - lo;
        zkhip_bases_upload(stub(g)->members[k], curve, group, xy + lo * 2 * coord_limbs(curve, group), inf ? inf + lo : nullptr, cnt, &b->member[k]);
    }
    *out = b;
    return ZKHIP_OK;
}
int zkhip_group_bases_from_scalars(zkhip_device_group *g, int curve, int group, const uint64_t *base, const uint64_t *scalars, size_t n, zkhip_group_bases **out) {
    const int world = stub(g)->n;
    auto *b = new zkhip_group_bases {curve, group, n, world, new zkhip_bases *[world]};
    for (int k = 0; k < world; ++k) {
        const size_t lo = part_lo(n, k, world), cnt = part_lo(n, k + 1, world) - lo;
        zkhip_bases_from_scalars(stub(g)->members[k], curve, group, base, scalars + 4 * lo, cnt, &b->member[k]);
    }
    *out = b;
    return ZKHIP_OK;
}
void zkhip_group_bases_free(zkhip_device_group *, zkhip_group_bases *b) {
    if (!b) return;
    for (int k = 0; k < b->world; ++k) delete b->member[k];
    delete[] b->member;
    delete b;
}
size_t zkhip_group_bases_size(const zkhip_group_bases *b) { return b->n; }
const zkhip_bases *zkhip_group_bases_member(const zkhip_group_bases *b, int member, size_t *first) {
    if (member < 0 || member >= b->world) return nullptr;
    if (first) *first = part_lo(b->n, member, b->world);
    return b->member[member];
}
int zkhip_group_msm(zkhip_device_group *, const zkhip_group_bases *b, size_t offset, size_t n, const uint64_t *scalars, uint64_t *out) {
    if (offset + n > b->n) return ZKHIP_ERR_RANGE;
    volatile uint64_t acc = 0;
    for (size_t i = 0; i < 4 * n; ++i) acc += scalars[i];
    memset(out, 0, 3 * coord_limbs(b->curve, b->group) * 8);
    return ZKHIP_OK;
}
int zkhip_group_ntt(zkhip_device_group *, int, uint64_t *data, size_t log_m, size_t batch, const uint64_t *, int, const uint64_t *) {
    touch(data, (batch << log_m) * 32);
    return ZKHIP_OK;
}
int zkhip_profile_enable(zkhip_ctx *, int) { return ZKHIP_OK; }
int zkhip_profile_reset(zkhip_ctx *) { return ZKHIP_OK; }
int zkhip_profile_filter(zkhip_ctx *, const char *) { return ZKHIP_OK; }
int zkhip_profile_get(zkhip_ctx *, const char *, double *ms, uint64_t *n) {
    *ms = 0;
    *n = 0;
    return ZKHIP_OK;
}
size_t zkhip_profile_dump(zkhip_ctx *, char *, size_t) { return 0; }
}
