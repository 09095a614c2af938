// The grand products of placeholder's permutation and lookup arguments, and the pointwise a b / c behind their multi-part forms.
// The permutation argument's (zk/snark/systems/plonk/placeholder/permutation_argument.hpp:103-136):
//   g_i = column_i + beta S_id_i + gamma,   h_i = column_i + beta S_sigma_i + gamma          (pointwise over the n rows, i < k)
//   V_P[0] = 1,   V_P[j] = V_P[j - 1] * prod_i g_i[j - 1] / prod_i h_i[j - 1]
// The lookup argument's V_L (lookup_argument.hpp:375-409) is the same recurrence over other rows (LookupRows below), zero behind usable_rows.
// The reference walks the rows one after the other with one field inversion per row.  Here: every lane takes PERM_CHUNK consecutive
// rows, forms their numerators and denominators, inverts the denominators with ONE inversion (Montgomery's trick), and the exclusive
// prefix PRODUCT over all rows is a three-level scan (lane chunk -> workgroup in LDS -> one workgroup over the workgroup totals).
// Everything between the loads and the stores is in Montgomery form; HBM holds canonical Fr (4 x u64) as everywhere.
#include <algorithm>

#include "ctx.hpp"
#include "fu.hpp"

using namespace zkhip;

namespace {

constexpr uint32_t PERM_CHUNK = 8, PERM_THREADS = 256;

template <class U>
ZK_D Fu<U> p_load_mont(const uint32_t *p, size_t i) {  // canonical in memory -> Montgomery
    const uint4 *q = reinterpret_cast<const uint4 *>(p) + 2 * i;
    const uint4 a = q[0], b = q[1];
    const uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fu_mul(fu_unpack<U>(s), Fu<U>::r2());
}
template <class U>
ZK_D void p_store_raw(uint32_t *p, size_t i, const Fu<U> &x) {  // x normalised, < 2^256: stored as it is
    uint32_t s[8];
    fu_pack<U>(s, x);
    uint4 *q = reinterpret_cast<uint4 *>(p) + 2 * i;
    q[0] = make_uint4(s[0], s[1], s[2], s[3]);
    q[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
template <class U>
ZK_D Fu<U> p_load_raw(const uint32_t *p, size_t i) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p) + 2 * i;
    const uint4 a = q[0], b = q[1];
    const uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fu_unpack<U>(s);
}
template <class U>
ZK_D Fu<U> from_mont(const Fu<U> &x) {  // Montgomery -> canonical representative
    Fu<U> one = Fu<U>::zero();
    one.v[0] = 1;
    return fu_cond_sub_p(fu_mul(x, one));
}
template <class U>
ZK_D Fu<U> mmul(const Fu<U> &a, const Fu<U> &b) { return fu_cond_sub_p(fu_mul(a, b)); }

// inclusive prefix product of one value per lane over the workgroup, in LDS (Hillis-Steele, log2(PERM_THREADS) rounds)
template <class U>
ZK_D Fu<U> block_scan_mul(uint32_t *lds, Fu<U> x, uint32_t t) {
    constexpr int L = U::L;
    auto put = [&](uint32_t i, const Fu<U> &v) {
#pragma unroll
        for (int l = 0; l < L; ++l) lds[l * PERM_THREADS + i] = v.v[l];
    };
    auto get = [&](uint32_t i) {
        Fu<U> v;
#pragma unroll
        for (int l = 0; l < L; ++l) v.v[l] = lds[l * PERM_THREADS + i];
        return v;
    };
    put(t, x);
    __syncthreads();
    for (uint32_t d = 1; d < PERM_THREADS; d <<= 1) {
        Fu<U> other;
        const bool take = t >= d;
        if (take) other = get(t - d);
        __syncthreads();
        if (take) {
            x = mmul(x, other);
            put(t, x);
        }
        __syncthreads();
    }
    return x;
}

// the rows of the permutation argument: numerator prod_i (column_i + beta S_id_i + gamma), denominator the same over S_sigma
template <class U>
struct PermRows {
    const uint32_t *const *cols, *const *sid, *const *ssig;
    uint32_t k;
    size_t n;
    uint32_t *gv, *hv;  // the g_v / h_v vectors (canonical, k x n), or null
    ZK_D void operator()(size_t row, const uint32_t *consts, Fu<U> &nm, Fu<U> &dn) const {
        const Fu<U> beta = p_load_raw<U>(consts, 0), gamma = p_load_raw<U>(consts, 1);  // Montgomery
        nm = Fu<U>::one();
        dn = Fu<U>::one();
        for (uint32_t i = 0; i < k; ++i) {
            const Fu<U> c = p_load_mont<U>(cols[i], row);
            const Fu<U> cg = fu_cond_sub_p(fu_add(c, gamma));
            const Fu<U> g = fu_cond_sub_p(fu_add(mmul(beta, p_load_mont<U>(sid[i], row)), cg));
            const Fu<U> h = fu_cond_sub_p(fu_add(mmul(beta, p_load_mont<U>(ssig[i], row)), cg));
            if (gv) p_store_raw<U>(gv, (size_t)i * n + row, from_mont(g));
            if (hv) p_store_raw<U>(hv, (size_t)i * n + row, from_mont(h));
            nm = mmul(nm, g);
            dn = mmul(dn, h);
        }
    }
};

// the rows of the lookup argument (lookup_argument.hpp:387-407), row = k - 1 of the reference's loop:
//   numerator (1 + beta)^k_in prod_i (gamma + input_i[row]) prod_i ((1 + beta) gamma + value_i[row] + beta value_i[row + 1]),
//   denominator prod_i ((1 + beta) gamma + sorted_i[row] + beta sorted_i[row + 1]);   consts[2] = (1 + beta)^k_in, consts[3] = (1 + beta) gamma
template <class U>
struct LookupRows {
    const uint32_t *const *in, *const *val, *const *sorted;
    uint32_t k_in, k_val, k_sorted;
    size_t n;
    ZK_D void operator()(size_t row, const uint32_t *consts, Fu<U> &nm, Fu<U> &dn) const {
        const Fu<U> beta = p_load_raw<U>(consts, 0), gamma = p_load_raw<U>(consts, 1), part1 = p_load_raw<U>(consts, 3);
        const size_t next = row + 1 == n ? 0 : row + 1;
        nm = p_load_raw<U>(consts, 2);
        dn = Fu<U>::one();
        for (uint32_t i = 0; i < k_in; ++i) nm = mmul(nm, fu_cond_sub_p(fu_add(gamma, p_load_mont<U>(in[i], row))));
        for (uint32_t i = 0; i < k_val; ++i) {
            const Fu<U> a = fu_cond_sub_p(fu_add(part1, p_load_mont<U>(val[i], row)));
            nm = mmul(nm, fu_cond_sub_p(fu_add(a, mmul(beta, p_load_mont<U>(val[i], next)))));
        }
        for (uint32_t i = 0; i < k_sorted; ++i) {
            const Fu<U> a = fu_cond_sub_p(fu_add(part1, p_load_mont<U>(sorted[i], row)));
            dn = mmul(dn, fu_cond_sub_p(fu_add(a, mmul(beta, p_load_mont<U>(sorted[i], next)))));
        }
    }
};

// pass 1: ratios of the lane's rows (stored in Montgomery form; rows >= `rows` count as ratio 1), the lane's exclusive prefix inside its
// workgroup and the workgroup's total
template <class U, class Rows>
__global__ __launch_bounds__(PERM_THREADS) void perm_scan_local(Rows rows_of, size_t rows, const uint32_t *__restrict__ consts, uint32_t *__restrict__ ratio,
                                                                uint32_t *__restrict__ lane_prefix, uint32_t *__restrict__ block_tot) {
    __shared__ uint32_t lds[U::L * PERM_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t lane = (size_t)blockIdx.x * PERM_THREADS + t, lo = lane * PERM_CHUNK;
    Fu<U> local = Fu<U>::one();
    if (lo < rows) {
        const uint32_t cnt = (uint32_t)(rows - lo < PERM_CHUNK ? rows - lo : PERM_CHUNK);
        Fu<U> nom[PERM_CHUNK], pre[PERM_CHUNK], den[PERM_CHUNK];
        Fu<U> acc = Fu<U>::one();
        for (uint32_t r = 0; r < cnt; ++r) {
            rows_of(lo + r, consts, nom[r], den[r]);
            pre[r] = acc;
            acc = mmul(acc, den[r]);
        }
        Fu<U> inv = fu_cond_sub_p(fu_inv(acc));  // a zero denominator (probability ~ k n / r) leaves zeros behind, as 1 / 0 "=" 0 in the reference's field type
        for (uint32_t r = cnt; r-- > 0;) {
            const Fu<U> q = mmul(nom[r], mmul(inv, pre[r]));
            inv = mmul(inv, den[r]);
            nom[r] = q;
        }
        for (uint32_t r = 0; r < cnt; ++r) {
            p_store_raw<U>(ratio, lo + r, nom[r]);
            local = mmul(local, nom[r]);
        }
    }
    const Fu<U> incl = block_scan_mul<U>(lds, local, t);
    // exclusive prefix of this lane inside the workgroup = inclusive value of the lane before it
    __syncthreads();
#pragma unroll
    for (int l = 0; l < U::L; ++l) lds[l * PERM_THREADS + t] = incl.v[l];
    __syncthreads();
    Fu<U> excl = Fu<U>::one();
    if (t > 0) {
#pragma unroll
        for (int l = 0; l < U::L; ++l) excl.v[l] = lds[l * PERM_THREADS + t - 1];
    }
    p_store_raw<U>(lane_prefix, lane, excl);
    if (t == PERM_THREADS - 1) p_store_raw<U>(block_tot, blockIdx.x, incl);
}

// pass 2: exclusive prefix products of the workgroup totals (one workgroup; every lane takes `per` consecutive totals)
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void perm_scan_top(uint32_t *__restrict__ block_tot, uint32_t nblk, uint32_t per) {
    __shared__ uint32_t lds[U::L * PERM_THREADS];
    const uint32_t t = threadIdx.x, lo = t * per;
    Fu<U> local = Fu<U>::one();
    for (uint32_t i = lo; i < lo + per && i < nblk; ++i) local = mmul(local, p_load_raw<U>(block_tot, i));
    const Fu<U> incl = block_scan_mul<U>(lds, local, t);
    __syncthreads();
#pragma unroll
    for (int l = 0; l < U::L; ++l) lds[l * PERM_THREADS + t] = incl.v[l];
    __syncthreads();
    Fu<U> run = Fu<U>::one();
    if (t > 0) {
#pragma unroll
        for (int l = 0; l < U::L; ++l) run.v[l] = lds[l * PERM_THREADS + t - 1];
    }
    for (uint32_t i = lo; i < lo + per && i < nblk; ++i) {
        const Fu<U> mine = p_load_raw<U>(block_tot, i);
        p_store_raw<U>(block_tot, i, run);  // exclusive
        run = mmul(run, mine);
    }
}

// pass 3: V[row] = (prefix of the workgroup) (prefix of the lane) (product of the lane's earlier ratios), canonical, for row <= rows;
// the entries behind that are zero (the lookup argument's V_L, lookup_argument.hpp:382-383; the permutation argument has rows = n)
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void perm_scan_apply(const uint32_t *__restrict__ ratio, const uint32_t *__restrict__ lane_prefix,
                                                                const uint32_t *__restrict__ block_pre, size_t n, size_t rows, uint32_t *__restrict__ vp) {
    const size_t lane = (size_t)blockIdx.x * PERM_THREADS + threadIdx.x, lo = lane * PERM_CHUNK;
    if (lo >= n) return;
    const uint32_t cnt = (uint32_t)(n - lo < PERM_CHUNK ? n - lo : PERM_CHUNK);
    if (lo > rows) {
        for (uint32_t r = 0; r < cnt; ++r) p_store_raw<U>(vp, lo + r, Fu<U>::zero());
        return;
    }
    Fu<U> run = mmul(p_load_raw<U>(block_pre, blockIdx.x), p_load_raw<U>(lane_prefix, lane));
    for (uint32_t r = 0; r < cnt; ++r) {
        if (lo + r <= rows) {
            p_store_raw<U>(vp, lo + r, from_mont(run));
            if (lo + r < rows) run = mmul(run, p_load_raw<U>(ratio, lo + r));
        } else {
            p_store_raw<U>(vp, lo + r, Fu<U>::zero());
        }
    }
}

// consts[0] = beta, consts[1] = gamma: canonical in, Montgomery out; consts[2] = (1 + beta)^k_in, consts[3] = (1 + beta) gamma (the lookup argument's)
template <class U>
__global__ void perm_setup(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t k_in) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const Fu<U> beta = fu_cond_sub_p(fu_mul(p_load_raw<U>(in, 0), Fu<U>::r2())), gamma = fu_cond_sub_p(fu_mul(p_load_raw<U>(in, 1), Fu<U>::r2()));
    const Fu<U> opb = fu_cond_sub_p(fu_add(Fu<U>::one(), beta));
    Fu<U> pw = Fu<U>::one();
    for (uint32_t i = 0; i < k_in; ++i) pw = mmul(pw, opb);
    p_store_raw<U>(out, 0, beta);
    p_store_raw<U>(out, 1, gamma);
    p_store_raw<U>(out, 2, pw);
    p_store_raw<U>(out, 3, mmul(opb, gamma));
}

// the three passes over `n` entries of which the first `rows` carry ratios; ptrs: the device pointer table of the row functor (count entries)
template <class U, class MakeRows>
int scan_run(zkhip_ctx *ctx, size_t count, MakeRows make_rows, size_t n, size_t rows, uint32_t k_in, const uint64_t *beta, const uint64_t *gamma, uint32_t *d_vp) {
    const size_t lanes = (n + PERM_CHUNK - 1) / PERM_CHUNK, nblk = (lanes + PERM_THREADS - 1) / PERM_THREADS;
    if (nblk > (size_t)PERM_THREADS * 4096) return ZKHIP_ERR_RANGE;
    const uint32_t per = (uint32_t)((nblk + PERM_THREADS - 1) / PERM_THREADS);
    size_t need = zkhip_ctx::ws_round(count * sizeof(void *)) + zkhip_ctx::ws_round(6 * 32) + zkhip_ctx::ws_round(n * 32) +
                  zkhip_ctx::ws_round(nblk * PERM_THREADS * 32) + zkhip_ctx::ws_round(nblk * 32);
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    const uint32_t **d_ptrs = ctx->ws_take<const uint32_t *>(count);
    uint32_t *d_consts = ctx->ws_take<uint32_t>(6 * 8);
    uint32_t *d_ratio = ctx->ws_take<uint32_t>(n * 8);
    uint32_t *d_lane = ctx->ws_take<uint32_t>(nblk * PERM_THREADS * 8);
    uint32_t *d_blk = ctx->ws_take<uint32_t>(nblk * 8);
    ctx->lagrange_stage.assign(16, 0u);  // host copies alive until the asynchronous copies ran (synchronised below); batch_ptrs filled by the caller
    memcpy(ctx->lagrange_stage.data(), beta, 32);
    memcpy(ctx->lagrange_stage.data() + 8, gamma, 32);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_ptrs, ctx->batch_ptrs.data(), count * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_consts + 32, ctx->lagrange_stage.data(), 64, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "perm_grand_product", perm_setup<U>, dim3(1), dim3(64), 0, d_consts + 32, d_consts, k_in);
    auto rows_of = make_rows(d_ptrs);
    ZK_LAUNCH(ctx, "perm_grand_product", (perm_scan_local<U, decltype(rows_of)>), dim3((unsigned)nblk), dim3(PERM_THREADS), 0, rows_of, rows, d_consts, d_ratio, d_lane,
              d_blk);
    ZK_LAUNCH(ctx, "perm_grand_product", perm_scan_top<U>, dim3(1), dim3(PERM_THREADS), 0, d_blk, (uint32_t)nblk, per);
    ZK_LAUNCH(ctx, "perm_grand_product", perm_scan_apply<U>, dim3((unsigned)nblk), dim3(PERM_THREADS), 0, d_ratio, d_lane, d_blk, n, rows, d_vp);
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the staged pointers / constants may be reused after return
    return ZKHIP_OK;
}

template <class U>
int perm_run(zkhip_ctx *ctx, size_t k, const void *const *d_cols, const void *const *d_sid, const void *const *d_ssig, size_t n, const uint64_t *beta,
             const uint64_t *gamma, uint32_t *d_gv, uint32_t *d_hv, uint32_t *d_vp) {
    ctx->batch_ptrs.resize(3 * k);
    for (size_t i = 0; i < k; ++i) {
        ctx->batch_ptrs[i] = (uint32_t *)d_cols[i];
        ctx->batch_ptrs[k + i] = (uint32_t *)d_sid[i];
        ctx->batch_ptrs[2 * k + i] = (uint32_t *)d_ssig[i];
    }
    return scan_run<U>(
        ctx, 3 * k, [&](const uint32_t **p) { return PermRows<U>{p, p + k, p + 2 * k, (uint32_t)k, n, d_gv, d_hv}; }, n, n, 0, beta, gamma, d_vp);
}

template <class U>
int lookup_run(zkhip_ctx *ctx, size_t k_in, const void *const *d_in, size_t k_val, const void *const *d_val, size_t k_sorted, const void *const *d_sorted, size_t n,
               size_t usable_rows, const uint64_t *beta, const uint64_t *gamma, uint32_t *d_vl) {
    ctx->batch_ptrs.clear();
    for (size_t i = 0; i < k_in; ++i) ctx->batch_ptrs.push_back((uint32_t *)d_in[i]);
    for (size_t i = 0; i < k_val; ++i) ctx->batch_ptrs.push_back((uint32_t *)d_val[i]);
    for (size_t i = 0; i < k_sorted; ++i) ctx->batch_ptrs.push_back((uint32_t *)d_sorted[i]);
    return scan_run<U>(
        ctx, k_in + k_val + k_sorted,
        [&](const uint32_t **p) { return LookupRows<U>{p, p + k_in, p + k_in + k_val, (uint32_t)k_in, (uint32_t)k_val, (uint32_t)k_sorted, n}; }, n, usable_rows,
        (uint32_t)k_in, beta, gamma, d_vl);
}

// out[j] = a[j] b[j] / c[j] for j < count, PERM_CHUNK rows per lane sharing one inversion: the intermediate polynomials of the multi-part
// permutation / lookup arguments (permutation_argument.hpp:196-198, lookup_argument.hpp:264-266: one inversion per row in the reference)
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void fr_vec_mul_div(const uint32_t *a, const uint32_t *b, const uint32_t *c, size_t count, uint32_t *out) {  // out may alias an input
    const size_t lane = (size_t)blockIdx.x * PERM_THREADS + threadIdx.x, lo = lane * PERM_CHUNK;
    if (lo >= count) return;
    const uint32_t cnt = (uint32_t)(count - lo < PERM_CHUNK ? count - lo : PERM_CHUNK);
    Fu<U> nom[PERM_CHUNK], pre[PERM_CHUNK], den[PERM_CHUNK];
    Fu<U> acc = Fu<U>::one();
    for (uint32_t r = 0; r < cnt; ++r) {
        nom[r] = mmul(p_load_mont<U>(a, lo + r), p_load_mont<U>(b, lo + r));
        den[r] = p_load_mont<U>(c, lo + r);
        pre[r] = acc;
        acc = mmul(acc, den[r]);
    }
    Fu<U> inv = fu_cond_sub_p(fu_inv(acc));  // a zero among the chunk's denominators leaves zeros in the whole chunk
    for (uint32_t r = cnt; r-- > 0;) {
        const Fu<U> q = mmul(nom[r], mmul(inv, pre[r]));
        inv = mmul(inv, den[r]);
        nom[r] = q;
    }
    for (uint32_t r = 0; r < cnt; ++r) p_store_raw<U>(out, lo + r, from_mont(nom[r]));
}

}  // namespace

extern "C" int zkhip_fr_vec_mul_div_dev(zkhip_ctx *ctx, int curve, const void *d_a, const void *d_b, const void *d_c, void *d_out, size_t count) {
    if (!ctx || (count && (!d_a || !d_b || !d_c || !d_out))) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (count >= ((size_t)1 << 39)) return ZKHIP_ERR_RANGE;
    if (count == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t lanes = (count + PERM_CHUNK - 1) / PERM_CHUNK;
    const dim3 grid((unsigned)((lanes + PERM_THREADS - 1) / PERM_THREADS));
    if (curve == CURVE_BLS12_381)
        ZK_LAUNCH(ctx, "fr_vec_mul_div", fr_vec_mul_div<BlsFrU>, grid, dim3(PERM_THREADS), 0, (const uint32_t *)d_a, (const uint32_t *)d_b, (const uint32_t *)d_c, count,
                  (uint32_t *)d_out);
    else
        ZK_LAUNCH(ctx, "fr_vec_mul_div", fr_vec_mul_div<BnFrU>, grid, dim3(PERM_THREADS), 0, (const uint32_t *)d_a, (const uint32_t *)d_b, (const uint32_t *)d_c, count,
                  (uint32_t *)d_out);
    return ZKHIP_OK;
}

extern "C" int zkhip_perm_grand_product_dev(zkhip_ctx *ctx, int curve, size_t k, const void *const *d_cols, const void *const *d_sid, const void *const *d_ssigma,
                                            size_t n, const uint64_t *beta, const uint64_t *gamma, void *d_g, void *d_h, void *d_vp) {
    if (!ctx || k == 0 || !d_cols || !d_sid || !d_ssigma || !beta || !gamma || (n && !d_vp)) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (k >= 4096 || n >= ((size_t)1 << 32)) return ZKHIP_ERR_RANGE;
    for (size_t i = 0; i < k; ++i)
        if (n && (!d_cols[i] || !d_sid[i] || !d_ssigma[i])) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (curve == CURVE_BLS12_381) return perm_run<BlsFrU>(ctx, k, d_cols, d_sid, d_ssigma, n, beta, gamma, (uint32_t *)d_g, (uint32_t *)d_h, (uint32_t *)d_vp);
    return perm_run<BnFrU>(ctx, k, d_cols, d_sid, d_ssigma, n, beta, gamma, (uint32_t *)d_g, (uint32_t *)d_h, (uint32_t *)d_vp);
}

/* V_L of the lookup argument (lookup_argument.hpp:375-409): V_L[0] = 1, V_L[k] = V_L[k - 1] g(k - 1) / h(k - 1) for k <= usable_rows, zero behind */
extern "C" int zkhip_lookup_grand_product_dev(zkhip_ctx *ctx, int curve, size_t k_in, const void *const *d_input, size_t k_val, const void *const *d_value,
                                              size_t k_sorted, const void *const *d_sorted, size_t n, size_t usable_rows, const uint64_t *beta, const uint64_t *gamma,
                                              void *d_vl) {
    if (!ctx || !beta || !gamma || (n && !d_vl) || (k_in && !d_input) || (k_val && !d_value) || (k_sorted && !d_sorted)) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (k_in >= 4096 || k_val >= 4096 || k_sorted >= 4096 || n >= ((size_t)1 << 32)) return ZKHIP_ERR_RANGE;
    if (n && usable_rows >= n) return ZKHIP_ERR_RANGE;
    for (size_t i = 0; n && i < k_in; ++i)
        if (!d_input[i]) return ZKHIP_ERR_INVALID;
    for (size_t i = 0; n && i < k_val; ++i)
        if (!d_value[i]) return ZKHIP_ERR_INVALID;
    for (size_t i = 0; n && i < k_sorted; ++i)
        if (!d_sorted[i]) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (curve == CURVE_BLS12_381)
        return lookup_run<BlsFrU>(ctx, k_in, d_input, k_val, d_value, k_sorted, d_sorted, n, usable_rows, beta, gamma, (uint32_t *)d_vl);
    return lookup_run<BnFrU>(ctx, k_in, d_input, k_val, d_value, k_sorted, d_sorted, n, usable_rows, beta, gamma, (uint32_t *)d_vl);
}
