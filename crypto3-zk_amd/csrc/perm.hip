// The grand products of placeholder's permutation and lookup arguments, and the pointwise a b / c behind their multi-part forms.
// The permutation argument's (zk/snark/systems/plonk/placeholder/permutation_argument.hpp:103-136):
//   g_i = column_i + beta S_id_i + gamma,   h_i = column_i + beta S_sigma_i + gamma          (pointwise over the n rows, i < k)
//   V_P[0] = 1,   V_P[j] = V_P[j - 1] * prod_i g_i[j - 1] / prod_i h_i[j - 1]
// The lookup argument's V_L (lookup_argument.hpp:375-409) is the same recurrence over other rows (LookupRows below), zero behind usable_rows.
// The reference walks the rows one after the other with one field inversion PER ROW.  Rounds 3-4 gave every lane 8 rows sharing an
// inversion (Montgomery's trick): 37 of the 57 products per row were the lanes' 295-product inversions, and their dependent chain (0.24 ms)
// sat under every call.  Round 5: ONE inversion per call.  With N_j = nominator of row j, D_j = its denominator,
//   V[j] = prod_{i < j} N_i / prod_{i < j} D_i = (prod_{i < j} N_i) (prod_{i >= j} D_i) / (prod_i D_i):
// an exclusive PREFIX product of the nominators, an inclusive SUFFIX product of the denominators (both three-level scans: lane chunk ->
// workgroup in LDS -> one workgroup over the workgroup totals), and the inverse of the total, taken once by the top-level workgroup.
// A zero denominator (probability ~ k n / r for honest inputs) is handled as the reference's row-by-row loop handles it (0^-1 = 0 in its
// field type: the row's ratio is 0, V stays as it is up to that row and is zero behind it): the FIRST such row z is found by an atomic
// minimum, zero denominators count as 1 in the products, and rows > z are written as zero -- bit for bit the reference's vector.
// No conversion products: HBM holds canonical Fr; a factor g = column + beta S + gamma is formed CANONICALLY (beta S = one Montgomery
// product of the canonical S with the Montgomery beta) and multiplied into an accumulator that starts at R^(factors + 1), so that after
// the row's factors it holds the Montgomery form of their product: 4 products per column and row instead of 9.
#include <algorithm>

#include "ctx.hpp"
#include "fu_safegcd.hpp"

using namespace zkhip;

namespace {

constexpr uint32_t PERM_CHUNK = 8, PERM_THREADS = 256;
// Where row `row`'s nominator / denominator live in the scratch vectors: inside each block of PERM_THREADS * PERM_CHUNK rows (one workgroup
// of the scan kernels) the r-th row of lane t sits at r * PERM_THREADS + t, so that the scan kernels -- lane = PERM_CHUNK consecutive rows --
// read consecutive 32-byte elements across a wave, while gp_rows (one row per lane) still writes runs of 8 elements (256 B).
ZK_D size_t perm_slot(size_t row) {
    constexpr size_t BLK = (size_t)PERM_THREADS * PERM_CHUNK;
    const size_t in = row % BLK;
    return row - in + (in % PERM_CHUNK) * PERM_THREADS + in / PERM_CHUNK;
}
enum : uint32_t { C_BETA_M = 0, C_GAMMA = 1, C_ACC_NOM = 2, C_ACC_DEN = 3, C_PART1 = 4, C_SLOTS = 8 };  // the constants' slots (32 B each)

template <class U>
ZK_D Fu<U> p_load_raw(const uint32_t *p, size_t i) {  // the stored words as they are: canonical Fr, or a Montgomery value written by p_store_raw
    const uint4 *q = reinterpret_cast<const uint4 *>(p) + 2 * i;
    const uint4 a = q[0], b = q[1];
    const uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fu_unpack<U>(s);
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
ZK_D Fu<U> from_mont(const Fu<U> &x) {  // Montgomery -> canonical representative
    Fu<U> one = Fu<U>::zero();
    one.v[0] = 1;
    return fu_cond_sub_p(fu_mul(x, one));
}
template <class U>
ZK_D Fu<U> mmul(const Fu<U> &a, const Fu<U> &b) { return fu_cond_sub_p(fu_mul(a, b)); }
// a value < 3p (a sum of canonical terms) -> canonical
template <class U>
ZK_D Fu<U> reduce3(const Fu<U> &a) { return fu_cond_sub_p(fu_cond_sub_p(a)); }

// inclusive prefix product of x and inclusive suffix product of y over the workgroup's lanes, in LDS (Hillis-Steele, log2(PERM_THREADS) rounds)
template <class U>
ZK_D void block_scan_pair(uint32_t *lds, Fu<U> &x, Fu<U> &y, uint32_t t) {
    constexpr int L = U::L;
    uint32_t *lx = lds, *ly = lds + L * PERM_THREADS;
    auto put = [&](uint32_t *base, uint32_t i, const Fu<U> &v) {
ZK_UNROLL
        for (int l = 0; l < L; ++l) base[l * PERM_THREADS + i] = v.v[l];
    };
    auto get = [&](const uint32_t *base, uint32_t i) {
        Fu<U> v;
ZK_UNROLL
        for (int l = 0; l < L; ++l) v.v[l] = base[l * PERM_THREADS + i];
        return v;
    };
    put(lx, t, x);
    put(ly, t, y);
    __syncthreads();
    for (uint32_t d = 1; d < PERM_THREADS; d <<= 1) {
        Fu<U> ox, oy;
        const bool tx = t >= d, ty = t + d < PERM_THREADS;
        if (tx) ox = get(lx, t - d);
        if (ty) oy = get(ly, t + d);
        __syncthreads();
        if (tx) {
            x = mmul(x, ox);
            put(lx, t, x);
        }
        if (ty) {
            y = mmul(y, oy);
            put(ly, t, y);
        }
        __syncthreads();
    }
}
// the inclusive results of block_scan_pair (still in LDS) shifted by one lane: exclusive prefix of x, exclusive suffix of y
template <class U>
ZK_D void block_exclusive(const uint32_t *lds, uint32_t t, Fu<U> &ex, Fu<U> &ey) {
    constexpr int L = U::L;
    ex = Fu<U>::one();
    ey = Fu<U>::one();
    if (t > 0) {
ZK_UNROLL
        for (int l = 0; l < L; ++l) ex.v[l] = lds[l * PERM_THREADS + t - 1];
    }
    if (t + 1 < PERM_THREADS) {
ZK_UNROLL
        for (int l = 0; l < L; ++l) ey.v[l] = lds[(L + l) * PERM_THREADS + t + 1];
    }
}

// the rows of the permutation argument: nominator prod_i (column_i + beta S_id_i + gamma), denominator the same over S_sigma
template <class U>
struct PermRows {
    const uint32_t *const *cols, *const *sid, *const *ssig;
    uint32_t k;
    size_t n;
    uint32_t *gv, *hv;  // the g_v / h_v vectors (canonical, k x n), or null
    ZK_D void operator()(size_t row, const uint32_t *consts, Fu<U> &nm, Fu<U> &dn) const {
        const Fu<U> beta = p_load_raw<U>(consts, C_BETA_M), gamma = p_load_raw<U>(consts, C_GAMMA);
        nm = p_load_raw<U>(consts, C_ACC_NOM);
        dn = p_load_raw<U>(consts, C_ACC_DEN);
        for (uint32_t i = 0; i < k; ++i) {
            const Fu<U> cg = fu_add(p_load_raw<U>(cols[i], row), gamma);
            const Fu<U> g = reduce3(fu_add(mmul(p_load_raw<U>(sid[i], row), beta), cg));
            const Fu<U> h = reduce3(fu_add(mmul(p_load_raw<U>(ssig[i], row), beta), cg));
            if (gv) p_store_raw<U>(gv, (size_t)i * n + row, g);
            if (hv) p_store_raw<U>(hv, (size_t)i * n + row, h);
            nm = fu_mul(nm, g);
            dn = fu_mul(dn, h);
        }
    }
};

// the rows of the lookup argument (lookup_argument.hpp:387-407), row = k - 1 of the reference's loop:
//   nominator (1 + beta)^k_in prod_i (gamma + input_i[row]) prod_i ((1 + beta) gamma + value_i[row] + beta value_i[row + 1]),
//   denominator prod_i ((1 + beta) gamma + sorted_i[row] + beta sorted_i[row + 1]);   (1 + beta)^k_in rides in the nominator's start value
template <class U>
struct LookupRows {
    const uint32_t *const *in, *const *val, *const *sorted;
    uint32_t k_in, k_val, k_sorted;
    size_t n;
    ZK_D void operator()(size_t row, const uint32_t *consts, Fu<U> &nm, Fu<U> &dn) const {
        const Fu<U> beta = p_load_raw<U>(consts, C_BETA_M), gamma = p_load_raw<U>(consts, C_GAMMA), part1 = p_load_raw<U>(consts, C_PART1);
        const size_t next = row + 1 == n ? 0 : row + 1;
        nm = p_load_raw<U>(consts, C_ACC_NOM);
        dn = p_load_raw<U>(consts, C_ACC_DEN);
        for (uint32_t i = 0; i < k_in; ++i) nm = fu_mul(nm, fu_add(gamma, p_load_raw<U>(in[i], row)));
        for (uint32_t i = 0; i < k_val; ++i)
            nm = fu_mul(nm, fu_add(fu_add(part1, p_load_raw<U>(val[i], row)), mmul(p_load_raw<U>(val[i], next), beta)));
        for (uint32_t i = 0; i < k_sorted; ++i)
            dn = fu_mul(dn, fu_add(fu_add(part1, p_load_raw<U>(sorted[i], row)), mmul(p_load_raw<U>(sorted[i], next), beta)));
    }
};

// pass 0: one row per lane, fully coalesced (consecutive lanes read consecutive 32-byte elements of every vector: the row functor moves
// 3 k + 2 k vectors, 704 B per row at k = 4, and is the memory-bound part of a call): nominator and denominator of the row in Montgomery
// form; a zero denominator counts as 1 and reports its row
template <class U, class Rows>
__global__ __launch_bounds__(PERM_THREADS) void gp_rows(Rows rows_of, size_t rows, const uint32_t *__restrict__ consts, uint32_t *__restrict__ nom,
                                                        uint32_t *__restrict__ den, uint32_t *__restrict__ first_zero) {
    const size_t row = (size_t)blockIdx.x * PERM_THREADS + threadIdx.x;
    if (row >= rows) return;
    Fu<U> nm, dn;
    rows_of(row, consts, nm, dn);
    nm = fu_cond_sub_p(nm);
    dn = fu_cond_sub_p(dn);
    if (dn.limbs_zero()) {
        atomicMin(first_zero, (uint32_t)row);
        dn = Fu<U>::one();
    }
    p_store_raw<U>(nom, perm_slot(row), nm);
    p_store_raw<U>(den, perm_slot(row), dn);
}

// the row functor's two products as CANONICAL vectors in natural order: prod_i g_i and prod_i h_i of the permutation argument on whatever domain
// the caller's vectors live on (zkhip_perm_factor_products_dev: the argument's g and h polynomials on the extended domain in ONE pass over the
// extended columns and permutation polynomials, instead of 2 k linear passes and two k-way products)
template <class U, class Rows>
__global__ __launch_bounds__(PERM_THREADS) void gp_products(Rows rows_of, size_t rows, const uint32_t *__restrict__ consts, uint32_t *__restrict__ out_n,
                                                            uint32_t *__restrict__ out_d) {
    const size_t row = (size_t)blockIdx.x * PERM_THREADS + threadIdx.x;
    if (row >= rows) return;
    Fu<U> nm, dn;
    rows_of(row, consts, nm, dn);
    p_store_raw<U>(out_n, row, from_mont(fu_cond_sub_p(nm)));
    p_store_raw<U>(out_d, row, from_mont(fu_cond_sub_p(dn)));
}

// pass 1: every lane takes PERM_CHUNK consecutive rows (rows >= `rows` count as 1 / 1): the products of their nominators and denominators,
// the lane's exclusive prefix (nominators) and exclusive suffix (denominators) inside its workgroup, the workgroup's totals
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void gp_local(size_t rows, const uint32_t *__restrict__ nom, const uint32_t *__restrict__ den,
                                                         uint32_t *__restrict__ lane_pre, uint32_t *__restrict__ lane_suf, uint32_t *__restrict__ blk_nom,
                                                         uint32_t *__restrict__ blk_den) {
    __shared__ uint32_t lds[2 * U::L * PERM_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t lane = (size_t)blockIdx.x * PERM_THREADS + t, lo = lane * PERM_CHUNK;
    Fu<U> ln = Fu<U>::one(), ld = Fu<U>::one();
    if (lo < rows) {
        const uint32_t cnt = (uint32_t)(rows - lo < PERM_CHUNK ? rows - lo : PERM_CHUNK);
        for (uint32_t r = 0; r < cnt; ++r) {
            ln = mmul(ln, p_load_raw<U>(nom, perm_slot(lo + r)));
            ld = mmul(ld, p_load_raw<U>(den, perm_slot(lo + r)));
        }
    }
    Fu<U> x = ln, y = ld;
    block_scan_pair<U>(lds, x, y, t);
    Fu<U> ex, ey;
    block_exclusive<U>(lds, t, ex, ey);
    p_store_raw<U>(lane_pre, lane, ex);
    p_store_raw<U>(lane_suf, lane, ey);
    if (t == PERM_THREADS - 1) p_store_raw<U>(blk_nom, blockIdx.x, x);  // inclusive prefix of the last lane = the workgroup's total
    if (t == 0) p_store_raw<U>(blk_den, blockIdx.x, y);                 // inclusive suffix of the first lane likewise
}

// pass 2 (one workgroup; every lane takes `per` consecutive workgroup totals): exclusive prefix products of the nominator totals, exclusive
// suffix products of the denominator totals, and the inverse of the denominators' grand total (consts slot C_SLOTS - 1)
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void gp_top(uint32_t *__restrict__ blk_nom, uint32_t *__restrict__ blk_den, uint32_t nblk, uint32_t per,
                                                       uint32_t *__restrict__ consts) {
    __shared__ uint32_t lds[2 * U::L * PERM_THREADS];
    const uint32_t t = threadIdx.x, lo = t * per;
    Fu<U> ln = Fu<U>::one(), ld = Fu<U>::one();
    for (uint32_t i = lo; i < lo + per && i < nblk; ++i) {
        ln = mmul(ln, p_load_raw<U>(blk_nom, i));
        ld = mmul(ld, p_load_raw<U>(blk_den, i));
    }
    Fu<U> x = ln, y = ld;
    block_scan_pair<U>(lds, x, y, t);
    Fu<U> ex, ey;
    block_exclusive<U>(lds, t, ex, ey);
    // the lane's own totals: prefix forward, suffix backward
    Fu<U> run = ex;
    for (uint32_t i = lo; i < lo + per && i < nblk; ++i) {
        const Fu<U> mine = p_load_raw<U>(blk_nom, i);
        p_store_raw<U>(blk_nom, i, run);
        run = mmul(run, mine);
    }
    run = ey;
    const uint32_t hi = lo + per < nblk ? lo + per : nblk;
    for (uint32_t i = hi; i-- > lo;) {
        const Fu<U> mine = p_load_raw<U>(blk_den, i);
        p_store_raw<U>(blk_den, i, run);
        run = mmul(run, mine);
    }
    if (t < 64) {  // the first wave, every lane the same value: y of lane 0 is the grand total (no zero among its factors)
        Fu<U> total;
ZK_UNROLL
        for (int l = 0; l < U::L; ++l) total.v[l] = lds[(U::L + l) * PERM_THREADS];
        const Fu<U> inv = fu_inv_gcd<U>(total);  // safegcd: ~15 k instructions on the call's critical path (Fermat: 70 k)
        if (t == 0) p_store_raw<U>(consts, C_SLOTS - 1, inv);
    }
}

// pass 3: V[row] = (nominators before the row) (denominators from the row on) / (all denominators), canonical, for row <= min(rows, first
// zero denominator's row); zero behind that (the lookup argument's V_L is zero behind usable_rows, lookup_argument.hpp:382-383; the
// permutation argument has rows = n)
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void gp_apply(const uint32_t *__restrict__ nom, const uint32_t *__restrict__ den, const uint32_t *__restrict__ lane_pre,
                                                         const uint32_t *__restrict__ lane_suf, const uint32_t *__restrict__ blk_pre, const uint32_t *__restrict__ blk_suf,
                                                         const uint32_t *__restrict__ consts, const uint32_t *__restrict__ first_zero, size_t n, size_t rows,
                                                         uint32_t *__restrict__ vp) {
    const size_t lane = (size_t)blockIdx.x * PERM_THREADS + threadIdx.x, lo = lane * PERM_CHUNK;
    if (lo >= n) return;
    const uint32_t cnt = (uint32_t)(n - lo < PERM_CHUNK ? n - lo : PERM_CHUNK);
    const size_t z = *first_zero, last = rows < z ? rows : z;  // the last row that holds a value
    if (lo > last) {
        for (uint32_t r = 0; r < cnt; ++r) p_store_raw<U>(vp, lo + r, Fu<U>::zero());
        return;
    }
    // suffix products of the lane's own denominators (rows >= `rows` hold none: 1)
    Fu<U> suf[PERM_CHUNK];
    Fu<U> s = mmul(mmul(p_load_raw<U>(blk_suf, blockIdx.x), p_load_raw<U>(lane_suf, lane)), p_load_raw<U>(consts, C_SLOTS - 1));
    for (uint32_t r = cnt; r-- > 0;) {
        if (lo + r < rows) s = mmul(s, p_load_raw<U>(den, perm_slot(lo + r)));
        suf[r] = s;
    }
    Fu<U> run = mmul(p_load_raw<U>(blk_pre, blockIdx.x), p_load_raw<U>(lane_pre, lane));
    for (uint32_t r = 0; r < cnt; ++r) {
        if (lo + r <= last) {
            p_store_raw<U>(vp, lo + r, from_mont(mmul(run, suf[r])));
            if (lo + r < rows) run = mmul(run, p_load_raw<U>(nom, perm_slot(lo + r)));
        } else {
            p_store_raw<U>(vp, lo + r, Fu<U>::zero());
        }
    }
}

// the constants: in[0] = beta, in[1] = gamma (canonical) -> beta in Montgomery form, gamma as it is, the accumulators' start values
// R^(f_nom + 1) (1 + beta)^pow_opb and R^(f_den + 1) (f: factors per row), (1 + beta) gamma; the first-zero-row cell to "none"
template <class U>
__global__ void gp_setup(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t f_nom, uint32_t f_den, uint32_t pow_opb, uint32_t *__restrict__ first_zero) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    *first_zero = 0xffffffffu;
    const Fu<U> beta = mmul(p_load_raw<U>(in, 0), Fu<U>::r2()), gamma_m = mmul(p_load_raw<U>(in, 1), Fu<U>::r2());
    const Fu<U> opb = fu_cond_sub_p(fu_add(Fu<U>::one(), beta));
    auto r_pow = [](uint32_t m) {  // R^(m + 1): the Montgomery one times R, m times
        Fu<U> t = Fu<U>::one();
        for (uint32_t i = 0; i < m; ++i) t = mmul(t, Fu<U>::r2());
        return t;
    };
    Fu<U> an = r_pow(f_nom);
    for (uint32_t i = 0; i < pow_opb; ++i) an = mmul(an, opb);
    p_store_raw<U>(out, C_BETA_M, beta);
    p_store_raw<U>(out, C_GAMMA, p_load_raw<U>(in, 1));
    p_store_raw<U>(out, C_ACC_NOM, an);
    p_store_raw<U>(out, C_ACC_DEN, r_pow(f_den));
    p_store_raw<U>(out, C_PART1, from_mont(mmul(opb, gamma_m)));
}

// the passes over `n` entries of which the first `rows` carry ratios; ptrs: the device pointer table of the row functor (count entries)
template <class U, class MakeRows>
int scan_run(zkhip_ctx *ctx, size_t count, MakeRows make_rows, size_t n, size_t rows, uint32_t f_nom, uint32_t f_den, uint32_t pow_opb, const uint64_t *beta,
             const uint64_t *gamma, uint32_t *d_vp) {
    const size_t lanes = (n + PERM_CHUNK - 1) / PERM_CHUNK, nblk = (lanes + PERM_THREADS - 1) / PERM_THREADS;
    if (nblk > (size_t)PERM_THREADS * 4096) return ZKHIP_ERR_RANGE;
    const uint32_t per = (uint32_t)((nblk + PERM_THREADS - 1) / PERM_THREADS);
    const size_t n_slots = nblk * PERM_THREADS * PERM_CHUNK;  // perm_slot stays inside the row's block: whole blocks of scratch
    size_t need = zkhip_ctx::ws_round(count * sizeof(void *)) + zkhip_ctx::ws_round((C_SLOTS + 2) * 32 + 64) + 2 * zkhip_ctx::ws_round(n_slots * 32) +
                  2 * zkhip_ctx::ws_round(nblk * PERM_THREADS * 32) + 2 * zkhip_ctx::ws_round(nblk * 32);
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    const uint32_t **d_ptrs = ctx->ws_take<const uint32_t *>(count);
    uint32_t *d_consts = ctx->ws_take<uint32_t>((C_SLOTS + 2) * 8 + 16);  // C_SLOTS constants | beta, gamma as uploaded | the first-zero-row cell
    uint32_t *d_nom = ctx->ws_take<uint32_t>(n_slots * 8), *d_den = ctx->ws_take<uint32_t>(n_slots * 8);
    uint32_t *d_lpre = ctx->ws_take<uint32_t>(nblk * PERM_THREADS * 8), *d_lsuf = ctx->ws_take<uint32_t>(nblk * PERM_THREADS * 8);
    uint32_t *d_bn = ctx->ws_take<uint32_t>(nblk * 8), *d_bd = ctx->ws_take<uint32_t>(nblk * 8);
    uint32_t *d_in = d_consts + C_SLOTS * 8, *d_z = d_consts + (C_SLOTS + 2) * 8;
    ctx->lagrange_stage.assign(16, 0u);  // host copies alive until the asynchronous copies ran (synchronised below); batch_ptrs filled by the caller
    memcpy(ctx->lagrange_stage.data(), beta, 32);
    memcpy(ctx->lagrange_stage.data() + 8, gamma, 32);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_ptrs, ctx->batch_ptrs.data(), count * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in, ctx->lagrange_stage.data(), 64, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "perm_grand_product", gp_setup<U>, dim3(1), dim3(64), 0, d_in, d_consts, f_nom, f_den, pow_opb, d_z);
    auto rows_of = make_rows(d_ptrs);
    if (rows)
        ZK_LAUNCH(ctx, "perm_grand_product", (gp_rows<U, decltype(rows_of)>), dim3((unsigned)((rows + PERM_THREADS - 1) / PERM_THREADS)), dim3(PERM_THREADS), 0, rows_of,
                  rows, d_consts, d_nom, d_den, d_z);
    ZK_LAUNCH(ctx, "perm_grand_product", gp_local<U>, dim3((unsigned)nblk), dim3(PERM_THREADS), 0, rows, d_nom, d_den, d_lpre, d_lsuf, d_bn, d_bd);
    ZK_LAUNCH(ctx, "perm_grand_product", gp_top<U>, dim3(1), dim3(PERM_THREADS), 0, d_bn, d_bd, (uint32_t)nblk, per, d_consts);
    ZK_LAUNCH(ctx, "perm_grand_product", gp_apply<U>, dim3((unsigned)nblk), dim3(PERM_THREADS), 0, d_nom, d_den, d_lpre, d_lsuf, d_bn, d_bd, d_consts, d_z, n, rows, d_vp);
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
        ctx, 3 * k, [&](const uint32_t **p) { return PermRows<U>{p, p + k, p + 2 * k, (uint32_t)k, n, d_gv, d_hv}; }, n, n, (uint32_t)k, (uint32_t)k, 0, beta, gamma, d_vp);
}

template <class U>
int perm_products_run(zkhip_ctx *ctx, size_t k, const void *const *d_cols, const void *const *d_sid, const void *const *d_ssig, size_t n, const uint64_t *beta,
                      const uint64_t *gamma, uint32_t *d_g, uint32_t *d_h) {
    ZK_TRY(ctx->ws_reserve(zkhip_ctx::ws_round(3 * k * sizeof(void *)) + zkhip_ctx::ws_round((C_SLOTS + 2) * 32 + 64)));
    ctx->ws_reset();
    const uint32_t **d_ptrs = ctx->ws_take<const uint32_t *>(3 * k);
    uint32_t *d_consts = ctx->ws_take<uint32_t>((C_SLOTS + 2) * 8 + 16);
    uint32_t *d_in = d_consts + C_SLOTS * 8, *d_z = d_consts + (C_SLOTS + 2) * 8;
    ctx->batch_ptrs.resize(3 * k);
    for (size_t i = 0; i < k; ++i) {
        ctx->batch_ptrs[i] = (uint32_t *)d_cols[i];
        ctx->batch_ptrs[k + i] = (uint32_t *)d_sid[i];
        ctx->batch_ptrs[2 * k + i] = (uint32_t *)d_ssig[i];
    }
    ctx->lagrange_stage.assign(16, 0u);
    memcpy(ctx->lagrange_stage.data(), beta, 32);
    memcpy(ctx->lagrange_stage.data() + 8, gamma, 32);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_ptrs, ctx->batch_ptrs.data(), 3 * k * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in, ctx->lagrange_stage.data(), 64, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "perm_factor_products", gp_setup<U>, dim3(1), dim3(64), 0, d_in, d_consts, (uint32_t)k, (uint32_t)k, 0u, d_z);
    PermRows<U> rows_of{d_ptrs, d_ptrs + k, d_ptrs + 2 * k, (uint32_t)k, n, nullptr, nullptr};
    ZK_LAUNCH(ctx, "perm_factor_products", (gp_products<U, PermRows<U>>), dim3((unsigned)((n + PERM_THREADS - 1) / PERM_THREADS)), dim3(PERM_THREADS), 0, rows_of, n,
              d_consts, d_g, d_h);
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the staged pointers / constants may be reused after return
    return ZKHIP_OK;
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
        (uint32_t)(k_in + k_val), (uint32_t)k_sorted, (uint32_t)k_in, beta, gamma, d_vl);
}

// out[j] = a[j] b[j] / c[j] for j < count, with ONE inversion per call: 1 / c_j = (prod_{i < j} c_i) (prod_{i > j} c_i) / prod_i c_i -- the
// intermediate polynomials of the multi-part permutation / lookup arguments (permutation_argument.hpp:196-198, lookup_argument.hpp:264-266: one
// inversion per row in the reference).  A zero c_j gives out[j] = 0 (0^-1 = 0 in the reference's field type) and counts as 1 in the products.
// Pass 1: c in Montgomery form (zeros as 1), the lane's exclusive prefix / suffix inside its workgroup, the workgroup totals (gp_top finishes
// them as for the grand products); pass 2: out.
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void md_local(const uint32_t *__restrict__ c, size_t count, uint32_t *__restrict__ cm, uint32_t *__restrict__ lane_pre,
                                                         uint32_t *__restrict__ lane_suf, uint32_t *__restrict__ blk_pre, uint32_t *__restrict__ blk_suf) {
    __shared__ uint32_t lds[2 * U::L * PERM_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t lane = (size_t)blockIdx.x * PERM_THREADS + t, lo = lane * PERM_CHUNK;
    Fu<U> l = Fu<U>::one();
    if (lo < count) {
        const uint32_t cnt = (uint32_t)(count - lo < PERM_CHUNK ? count - lo : PERM_CHUNK);
        for (uint32_t r = 0; r < cnt; ++r) {
            Fu<U> x = mmul(p_load_raw<U>(c, lo + r), Fu<U>::r2());
            if (x.limbs_zero()) x = Fu<U>::zero();  // marked: md_apply writes 0 for the row and skips it in the products
            p_store_raw<U>(cm, lo + r, x);
            if (!x.limbs_zero()) l = mmul(l, x);
        }
    }
    Fu<U> x = l, y = l;
    block_scan_pair<U>(lds, x, y, t);
    Fu<U> ex, ey;
    block_exclusive<U>(lds, t, ex, ey);
    p_store_raw<U>(lane_pre, lane, ex);
    p_store_raw<U>(lane_suf, lane, ey);
    if (t == PERM_THREADS - 1) p_store_raw<U>(blk_pre, blockIdx.x, x);
    if (t == 0) p_store_raw<U>(blk_suf, blockIdx.x, y);
}
template <class U>
__global__ __launch_bounds__(PERM_THREADS) void md_apply(const uint32_t *a, const uint32_t *b, const uint32_t *__restrict__ cm, const uint32_t *__restrict__ lane_pre,
                                                         const uint32_t *__restrict__ lane_suf, const uint32_t *__restrict__ blk_pre, const uint32_t *__restrict__ blk_suf,
                                                         const uint32_t *__restrict__ consts, size_t count, uint32_t *out) {  // out may alias a or b
    const size_t lane = (size_t)blockIdx.x * PERM_THREADS + threadIdx.x, lo = lane * PERM_CHUNK;
    if (lo >= count) return;
    const uint32_t cnt = (uint32_t)(count - lo < PERM_CHUNK ? count - lo : PERM_CHUNK);
    // suf[r] = product of the lane's c behind row r, times everything behind the lane, times 1 / total, times R (so that the last product below
    // leaves the canonical value: a b are multiplied in as they are)
    Fu<U> suf[PERM_CHUNK], cs[PERM_CHUNK];
    Fu<U> s = mmul(mmul(mmul(p_load_raw<U>(blk_suf, blockIdx.x), p_load_raw<U>(lane_suf, lane)), p_load_raw<U>(consts, C_SLOTS - 1)), Fu<U>::r2());
    for (uint32_t r = cnt; r-- > 0;) {
        cs[r] = p_load_raw<U>(cm, lo + r);
        suf[r] = s;
        if (!cs[r].limbs_zero()) s = mmul(s, cs[r]);
    }
    Fu<U> run = mmul(p_load_raw<U>(blk_pre, blockIdx.x), p_load_raw<U>(lane_pre, lane));
    for (uint32_t r = 0; r < cnt; ++r) {
        if (cs[r].limbs_zero()) {
            p_store_raw<U>(out, lo + r, Fu<U>::zero());
            continue;
        }
        // a b / R (canonical operands) -> times prefix (Montgomery) -> a b prefix / R -> times suf (Montgomery value times R^2) -> canonical a b / c
        const Fu<U> ab = fu_mul(p_load_raw<U>(a, lo + r), p_load_raw<U>(b, lo + r));
        p_store_raw<U>(out, lo + r, mmul(fu_mul(ab, run), suf[r]));
        run = mmul(run, cs[r]);
    }
}

template <class U>
int mul_div_run(zkhip_ctx *ctx, const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out, size_t count) {
    const size_t lanes = (count + PERM_CHUNK - 1) / PERM_CHUNK, nblk = (lanes + PERM_THREADS - 1) / PERM_THREADS;
    if (nblk > (size_t)PERM_THREADS * 4096) return ZKHIP_ERR_RANGE;
    const uint32_t per = (uint32_t)((nblk + PERM_THREADS - 1) / PERM_THREADS);
    size_t need = zkhip_ctx::ws_round(C_SLOTS * 32) + zkhip_ctx::ws_round(count * 32) + 2 * zkhip_ctx::ws_round(nblk * PERM_THREADS * 32) + 2 * zkhip_ctx::ws_round(nblk * 32);
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    uint32_t *d_consts = ctx->ws_take<uint32_t>(C_SLOTS * 8);
    uint32_t *d_cm = ctx->ws_take<uint32_t>(count * 8);
    uint32_t *d_lpre = ctx->ws_take<uint32_t>(nblk * PERM_THREADS * 8), *d_lsuf = ctx->ws_take<uint32_t>(nblk * PERM_THREADS * 8);
    uint32_t *d_bp = ctx->ws_take<uint32_t>(nblk * 8), *d_bs = ctx->ws_take<uint32_t>(nblk * 8);
    ZK_LAUNCH(ctx, "fr_vec_mul_div", md_local<U>, dim3((unsigned)nblk), dim3(PERM_THREADS), 0, c, count, d_cm, d_lpre, d_lsuf, d_bp, d_bs);
    ZK_LAUNCH(ctx, "fr_vec_mul_div", gp_top<U>, dim3(1), dim3(PERM_THREADS), 0, d_bp, d_bs, (uint32_t)nblk, per, d_consts);
    ZK_LAUNCH(ctx, "fr_vec_mul_div", md_apply<U>, dim3((unsigned)nblk), dim3(PERM_THREADS), 0, a, b, d_cm, d_lpre, d_lsuf, d_bp, d_bs, d_consts, count, out);
    return ZKHIP_OK;
}

}  // namespace

extern "C" int zkhip_fr_vec_mul_div_dev(zkhip_ctx *ctx, int curve, const void *d_a, const void *d_b, const void *d_c, void *d_out, size_t count) {
    if (!ctx || (count && (!d_a || !d_b || !d_c || !d_out))) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (count >= ((size_t)1 << 39)) return ZKHIP_ERR_RANGE;
    if (count == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (curve == CURVE_BLS12_381) return mul_div_run<BlsFrU>(ctx, (const uint32_t *)d_a, (const uint32_t *)d_b, (const uint32_t *)d_c, (uint32_t *)d_out, count);
    return mul_div_run<BnFrU>(ctx, (const uint32_t *)d_a, (const uint32_t *)d_b, (const uint32_t *)d_c, (uint32_t *)d_out, count);
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

/* prod_i (column_i + beta S_id_i + gamma) and prod_i (column_i + beta S_sigma_i + gamma), pointwise over n entries of vectors on ANY domain: the g and
 * h polynomials of the permutation argument (permutation_argument.hpp:140-160) on the extended domain, from the extended columns in one pass */
extern "C" int zkhip_perm_factor_products_dev(zkhip_ctx *ctx, int curve, size_t k, const void *const *d_cols, const void *const *d_sid, const void *const *d_ssigma,
                                              size_t n, const uint64_t *beta, const uint64_t *gamma, void *d_g, void *d_h) {
    if (!ctx || k == 0 || !d_cols || !d_sid || !d_ssigma || !beta || !gamma || (n && (!d_g || !d_h))) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (k >= 4096 || n >= ((size_t)1 << 38)) return ZKHIP_ERR_RANGE;
    for (size_t i = 0; i < k; ++i)
        if (n && (!d_cols[i] || !d_sid[i] || !d_ssigma[i])) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (curve == CURVE_BLS12_381) return perm_products_run<BlsFrU>(ctx, k, d_cols, d_sid, d_ssigma, n, beta, gamma, (uint32_t *)d_g, (uint32_t *)d_h);
    return perm_products_run<BnFrU>(ctx, k, d_cols, d_sid, d_ssigma, n, beta, gamma, (uint32_t *)d_g, (uint32_t *)d_h);
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
