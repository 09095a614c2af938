// Radix-2 number-theoretic transform over the scalar field (BLS12-381 Fr / BN254 Fr) for gfx950.
//
// Replaces evaluation_domain<F>::fft / inverse_fft and math::multiply_by_coset as called at
//   zk/snark/reductions/r1cs_to_qap.hpp:250-315 (7 transforms per Groth16 proof),
//   zk/snark/arithmetization/plonk/detail/column_polynomial.hpp:53,
//   polynomial_dfs::coefficients()/resize() inside kzg.hpp:431 and basic_fri.hpp:452-455.
// Natural order in, natural order out, out[i] = sum_j in[j] omega^(ij): the result is the DFT itself,
// so it is bit-identical to any correct radix-2 implementation.
//
// Algorithm: multi-pass Stockham autosort.  m = R_1 R_2 ... R_p, R_i = 2^(s_i), s_i <= 8.  Pass i with
// running sub-transform size Ns = R_1...R_(i-1) does, for every j in [0, m/R):
//     k = j mod Ns;   u[t] = x[j + t m/R] * omega_(Ns R)^(k t);   v = DFT_R(u);   y[(j-k) R + k + t' Ns] = v[t']
// The factor omega_(Ns R)^(k t) of pass i+1 is applied by pass i while it STORES (each stored element knows
// its next-pass (k, t)), so every pass is: load, s_i radix-2 stages, one product, store.
// (tests/test_ntt_model.py replays this index arithmetic line by line on the CPU against the O(n^2) DFT.)
//
// One workgroup owns a tile of T consecutive j's x all R values of t, staged in LDS as 9 x 29-bit lazy limbs
// (fu.hpp; three planes: 16 B + 16 B + 4 B per element, so every LDS access is conflict-free).  Rows are
// written bit-reversed, the stages are decimation-in-time, (a, b) -> (a + w b, a - w b), so values grow by at
// most 4p per stage (33p after 8 stages, inside the 6-7 bits of slack of R = 2^261) and are only carry-
// normalised; the product at the store brings them back below 2p, one conditional subtraction makes them
// canonical.  Global reads are runs of T*32 B, writes runs of >= T*32 B.
//
// Data stays CANONICAL in HBM between passes; twiddles are Montgomery, so mul(x, w) = x*w is canonical and no
// conversion pass exists.  omega^e for arbitrary e comes from a two-level table (lo[e mod 2^h] * hi[e >> h],
// 2 * 2^h entries, L2-resident) instead of an m/2-entry table in HBM.  The coset pre-scale (forward), the 1/m
// and the coset^-1 post-scale (inverse) ride on the first load / the last store.
#include <algorithm>

#include "ctx.hpp"
#include "fu.hpp"

using namespace zkhip;

struct NttTables {
    int curve;
    size_t log_m;
    int inverse;
    bool has_coset;
    uint64_t omega[4], coset[4];
    int lo_bits;
    uint32_t *d_lo = nullptr, *d_hi = nullptr;    // omega^i, omega^(i << lo_bits)          (Montgomery, SL words each)
    uint32_t *d_clo = nullptr, *d_chi = nullptr;  // g^i, g^(i << lo_bits), g = coset or coset^-1
    uint32_t *d_scale = nullptr;                  // [0] = 1/m (inverse) or 1 (forward), Montgomery
    uint32_t *d_base = nullptr;                   // [omega_eff, g_eff] (Montgomery)
};

struct NttPass {
    const uint32_t *in;
    uint32_t *out;
    uint32_t log_m, s, log_ns, log_t;
    uint32_t tiles_per_poly;
    uint32_t next_s;            // radix bits of the following pass, 0 on the last pass
    const uint32_t *lo, *hi;
    uint32_t lo_bits;
    const uint32_t *clo, *chi;  // coset power tables (g or g^-1), or null
    const uint32_t *scale;      // last pass: 1/m (inverse) or the Montgomery one (forward)
    uint32_t pre_coset;         // multiply by g^index while loading   (first pass, forward coset)
    uint32_t post_coset;        // multiply by g^index while storing   (last pass, inverse coset)
};

// out[i] = base^(i << shift), i < count  (Montgomery in/out, canonical representatives)
template <class U>
__global__ __launch_bounds__(256) void ntt_pow_table(const uint32_t *__restrict__ base, uint32_t count, uint32_t shift,
                                                     uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fu<U> b = fu_load<U>(base), r = Fu<U>::one();
    uint64_t e = (uint64_t)i << shift;
    while (e) {
        if (e & 1) r = fu_mul(r, b);
        b = fu_mul(b, b);
        e >>= 1;
    }
    fu_store<U>(out + (size_t)i * U::SL, fu_cond_sub_p(r));
}

// base[0] = omega or omega^-1, base[1] = g or g^-1 (Montgomery); scale = 1/m (inverse) or 1 (forward)
template <class U>
__global__ void ntt_setup(const uint32_t *__restrict__ omega_c, const uint32_t *__restrict__ coset_c, int inverse, uint32_t log_m,
                          uint32_t *__restrict__ base, uint32_t *__restrict__ scale) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Fu<U> w = fu_from_canonical<U>(omega_c);
    if (inverse) w = fu_inv(w);
    fu_store<U>(base, fu_cond_sub_p(w));
    if (coset_c != nullptr) {
        Fu<U> g = fu_from_canonical<U>(coset_c);
        if (inverse) g = fu_inv(g);
        fu_store<U>(base + U::SL, fu_cond_sub_p(g));
    }
    Fu<U> sc = Fu<U>::one();
    if (inverse) {
        Fu<U> two = fu_add(Fu<U>::one(), Fu<U>::one()), m = Fu<U>::one();
        for (uint32_t i = 0; i < log_m; ++i) m = fu_mul_call(m, two);
        sc = fu_inv(m);
    }
    fu_store<U>(scale, fu_cond_sub_p(sc));
}

template <class U>
ZK_D Fu<U> tw_lookup(const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi, uint32_t lo_bits, uint64_t e) {
    uint32_t el = (uint32_t)(e & ((1u << lo_bits) - 1));
    uint32_t eh = (uint32_t)(e >> lo_bits);
    Fu<U> a = fu_load<U>(lo + (size_t)el * U::SL);
    if (eh == 0) return a;
    return fu_cond_sub_p(fu_mul(a, fu_load<U>(hi + (size_t)eh * U::SL)));  // canonical, like the table entries
}

ZK_D uint32_t bitrev(uint32_t v, uint32_t bits) { return bits == 0 ? 0 : (__brev(v) >> (32 - bits)); }

// LDS element e of `nelem`: limbs 0-3 at plane 0, 4-7 at plane 1 (uint4 each), limb 8 at plane 2 (u32)
template <class U>
ZK_D Fu<U> lds_get(const uint4 *lds, uint32_t nelem, uint32_t e) {
    static_assert(U::L == 9, "Fr compute form is 9 x 29-bit limbs");
    Fu<U> r;
    uint4 a = lds[e], b = lds[nelem + e];
    r.v[0] = a.x, r.v[1] = a.y, r.v[2] = a.z, r.v[3] = a.w;
    r.v[4] = b.x, r.v[5] = b.y, r.v[6] = b.z, r.v[7] = b.w;
    r.v[8] = reinterpret_cast<const uint32_t *>(lds + 2 * nelem)[e];
    return r;
}
template <class U>
ZK_D void lds_put(uint4 *lds, uint32_t nelem, uint32_t e, const Fu<U> &x) {
    lds[e] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    lds[nelem + e] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    reinterpret_cast<uint32_t *>(lds + 2 * nelem)[e] = x.v[8];
}

// canonical 8 x u32 in global memory <-> lazy limbs
template <class U>
ZK_D Fu<U> g_load(const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fu_unpack<U>(s);
}
template <class U>
ZK_D void g_store(uint32_t *p, const Fu<U> &x) {
    uint32_t s[8];
    fu_pack<U>(s, x);
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(s[0], s[1], s[2], s[3]);
    q[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

template <class U>
__global__ __launch_bounds__(256) void ntt_pass(NttPass p) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const uint32_t R = 1u << p.s, T = 1u << p.log_t, nelem = R * T;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    // tile: 2 * nelem uint4 + nelem u32 (rounded to uint4); stage twiddles behind it, same three-plane shape
    const uint32_t tile_u4 = 2 * nelem + (nelem + 3) / 4;
    uint4 *twr = lds + tile_u4;

    const uint32_t poly = blockIdx.x / p.tiles_per_poly;
    const uint32_t tile = blockIdx.x % p.tiles_per_poly;
    const uint32_t log_stride = p.log_m - p.s;  // m / R
    const uint64_t j0 = (uint64_t)tile << p.log_t;
    const uint32_t *src = p.in + ((size_t)poly << p.log_m) * 8;
    uint32_t *dst = p.out + ((size_t)poly << p.log_m) * 8;
    const uint32_t ns_mask = (1u << p.log_ns) - 1;  // log_ns < 32 always (m <= 2^32)
    const uint32_t nhalf = R / 2 > 0 ? R / 2 : 1;

    // stage twiddles omega_R^q = omega^(q m/R), q < R/2
    for (uint32_t q = tid; q < R / 2; q += nth) lds_put(twr, nhalf, q, tw_lookup<U>(p.lo, p.hi, p.lo_bits, (uint64_t)q << log_stride));
    // load tile: row bitrev(t), column c  <-  x[j0 + c + t m/R]   [* g^index on the first pass of a coset transform]
    for (uint32_t e = tid; e < nelem; e += nth) {
        uint32_t t = e >> p.log_t, c = e & (T - 1);
        uint64_t gi = j0 + c + ((uint64_t)t << log_stride);
        Fu<U> x = g_load<U>(src + gi * 8);
        if (p.pre_coset) x = fu_mul(x, tw_lookup<U>(p.clo, p.chi, p.lo_bits, gi));
        lds_put(lds, nelem, (bitrev(t, p.s) << p.log_t) + c, x);
    }
    __syncthreads();
    // s radix-2 DIT stages over the rows (bit-reversed in, natural out): (a, b) -> (a + w b, a + 4p - w b)
    const uint32_t nbf = nelem >> 1;
    for (uint32_t st = 0; st < p.s; ++st) {
        const uint32_t h = 1u << st;
        for (uint32_t bf = tid; bf < nbf; bf += nth) {
            uint32_t c = bf & (T - 1), q = bf >> p.log_t;
            uint32_t qq = q & (h - 1);
            uint32_t i0 = ((q - qq) << 1) + qq;
            uint32_t e0 = (i0 << p.log_t) + c, e1 = ((i0 + h) << p.log_t) + c;
            Fu<U> a = lds_get<U>(lds, nelem, e0), b = lds_get<U>(lds, nelem, e1);
            if (st != 0) b = fu_mul(b, lds_get<U>(twr, nhalf, qq << (p.s - 1 - st)));  // st = 0: w = 1
            else b = fu_cond_sub_p(b);  // loaded values are < 2p; keep the subtrahend below 3p
            lds_put(lds, nelem, e0, fu_add(a, b));
            lds_put(lds, nelem, e1, fu_sub<4>(a, b));
        }
        __syncthreads();
    }
    // store: y[(j - k) R + k + t' Ns] = v[t'] * (next pass's twiddle | final scale)
    const uint32_t last = p.next_s == 0;
    const uint32_t n_log_ns = p.log_ns + p.s;                  // next pass: Ns' = Ns R
    const uint32_t n_log_stride = p.log_m - p.next_s;          // m / R'
    const uint32_t n_ns_mask = (1u << n_log_ns) - 1;
    const uint32_t n_tw_shift = p.log_m - n_log_ns - p.next_s;  // omega_(Ns' R') = omega^(m / (Ns' R'))
    Fu<U> scale = Fu<U>::one();
    if (last) scale = fu_load<U>(p.scale);
    for (uint32_t e = tid; e < nelem; e += nth) {
        uint32_t tp, c;
        if (p.log_ns >= p.log_t) {
            tp = e >> p.log_t;
            c = e & (T - 1);
        } else {  // Ns < T: (c_hi, t', k_lo) with k_lo fastest makes each c_hi a contiguous run of Ns*R
            uint32_t k_lo = e & ns_mask;
            tp = (e >> p.log_ns) & (R - 1);
            uint32_t c_hi = e >> (p.log_ns + p.s);
            c = (c_hi << p.log_ns) + k_lo;
        }
        uint64_t j = j0 + c;
        uint64_t k = j & ns_mask;
        uint64_t oi = ((j - k) << p.s) + k + ((uint64_t)tp << p.log_ns);
        Fu<U> x = lds_get<U>(lds, nelem, (tp << p.log_t) + c);
        Fu<U> f;
        if (!last) {
            uint64_t jn = oi & ((1ull << n_log_stride) - 1), tn = oi >> n_log_stride;
            uint64_t ex = ((jn & n_ns_mask) * tn) << n_tw_shift;
            f = tw_lookup<U>(p.lo, p.hi, p.lo_bits, ex);
        } else {
            f = scale;
            if (p.post_coset) f = fu_cond_sub_p(fu_mul(f, tw_lookup<U>(p.clo, p.chi, p.lo_bits, oi)));
        }
        g_store<U>(dst + oi * 8, fu_cond_sub_p(fu_mul(x, f)));
    }
}

// ---- host side ------------------------------------------------------------------------------------
void zk_ntt_free_tables(zkhip_ctx *ctx) {
    for (NttTables *t : ctx->ntt_tables) {
        (void)hipFree(t->d_lo);
        (void)hipFree(t->d_hi);
        (void)hipFree(t->d_clo);
        (void)hipFree(t->d_chi);
        (void)hipFree(t->d_scale);
        (void)hipFree(t->d_base);
        delete t;
    }
    ctx->ntt_tables.clear();
}

template <class U>
static int ntt_get_tables(zkhip_ctx *ctx, int curve, size_t log_m, const uint64_t *omega, int inverse, const uint64_t *coset,
                          NttTables **out) {
    for (NttTables *t : ctx->ntt_tables) {
        if (t->curve == curve && t->log_m == log_m && t->inverse == inverse && t->has_coset == (coset != nullptr) &&
            memcmp(t->omega, omega, 32) == 0 && (coset == nullptr || memcmp(t->coset, coset, 32) == 0)) {
            *out = t;
            return 0;
        }
    }
    NttTables *t = new NttTables();
    t->curve = curve;
    t->log_m = log_m;
    t->inverse = inverse;
    t->has_coset = coset != nullptr;
    memcpy(t->omega, omega, 32);
    if (coset) memcpy(t->coset, coset, 32);
    t->lo_bits = (int)((log_m + 1) / 2);
    if (t->lo_bits < 1) t->lo_bits = 1;
    const uint32_t nlo = 1u << t->lo_bits;
    const uint32_t nhi = (uint32_t)(((size_t)1 << log_m) >> t->lo_bits) + 1;
    const size_t eb = U::SL * 4;  // bytes per table entry
    ctx->ntt_tables.push_back(t);  // owned by the context from here on (freed in zk_ntt_free_tables)
    uint32_t *d_in = nullptr;
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&d_in, 64));
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_base, 2 * eb));
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_scale, eb));
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_lo, (size_t)nlo * eb));
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_hi, (size_t)nhi * eb));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in, omega, 32, hipMemcpyHostToDevice, ctx->stream));
    if (coset) ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in + 8, coset, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "ntt_setup", ntt_setup<U>, dim3(1), dim3(64), 0, d_in, coset ? d_in + 8 : (const uint32_t *)nullptr, inverse,
              (uint32_t)log_m, t->d_base, t->d_scale);
    ZK_LAUNCH(ctx, "ntt_pow_table", ntt_pow_table<U>, dim3((nlo + 255) / 256), dim3(256), 0, t->d_base, nlo, 0u, t->d_lo);
    ZK_LAUNCH(ctx, "ntt_pow_table", ntt_pow_table<U>, dim3((nhi + 255) / 256), dim3(256), 0, t->d_base, nhi, (uint32_t)t->lo_bits, t->d_hi);
    if (coset) {
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_clo, (size_t)nlo * eb));
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_chi, (size_t)nhi * eb));
        ZK_LAUNCH(ctx, "ntt_pow_table", ntt_pow_table<U>, dim3((nlo + 255) / 256), dim3(256), 0, t->d_base + U::SL, nlo, 0u, t->d_clo);
        ZK_LAUNCH(ctx, "ntt_pow_table", ntt_pow_table<U>, dim3((nhi + 255) / 256), dim3(256), 0, t->d_base + U::SL, nhi, (uint32_t)t->lo_bits,
                  t->d_chi);
    }
    // omega must be a PRIMITIVE m-th root of unity (the caller's evaluation domain is the basic radix-2 one): omega^m = 1
    // and omega^(m/2) != 1, read back from the tables just built.  Anything else would transform over the wrong domain.
    const size_t half = ((size_t)1 << log_m) >> 1;
    uint32_t w_m[U::L], w_half[U::L];
    const uint32_t *d_half = half < nlo ? t->d_lo + half * U::SL : t->d_hi + (half >> t->lo_bits) * U::SL;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(w_m, t->d_hi + (size_t)(nhi - 1) * U::SL, sizeof(w_m), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(w_half, d_half, sizeof(w_half), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_in);
    bool m_is_one = true, half_is_one = true;
    for (int i = 0; i < U::L; ++i) {
        m_is_one = m_is_one && w_m[i] == U::r1(i);
        half_is_one = half_is_one && w_half[i] == U::r1(i);
    }
    if (!m_is_one || half_is_one) {
        ctx->ntt_tables.pop_back();
        (void)hipFree(t->d_lo), (void)hipFree(t->d_hi), (void)hipFree(t->d_clo), (void)hipFree(t->d_chi), (void)hipFree(t->d_scale), (void)hipFree(t->d_base);
        delete t;
        ctx->last_error = "omega is not a primitive 2^" + std::to_string(log_m) + "-th root of unity";
        return ZKHIP_ERR_INVALID;
    }
    *out = t;
    return 0;
}

template <class U>
static int ntt_run_t(zkhip_ctx *ctx, int curve, uint32_t *d_data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
                     const uint64_t *coset) {
    if (batch == 0 || log_m == 0) return 0;  // a 1-point transform is the identity (also with coset: g^0 = 1, 1/1 = 1)
    NttTables *tb = nullptr;
    ZK_TRY(ntt_get_tables<U>(ctx, curve, log_m, omega, inverse, coset, &tb));
    const int smax = std::max(1, std::min(10, ctx->opt_ntt_radix_log));
    const int np = (int)((log_m + smax - 1) / smax);
    // split log_m into np nearly equal radices, larger ones first
    int sv[64];
    for (int i = 0; i < np; ++i) sv[i] = (int)(log_m / np) + (i < (int)(log_m % np) ? 1 : 0);
    const size_t m = (size_t)1 << log_m;
    const size_t bytes = batch * m * 32;
    uint32_t *wsA = nullptr, *wsB = nullptr;
    if (np > 1) {
        size_t need = zkhip_ctx::ws_round(bytes) * ((np & 1) ? 2 : 1);
        ZK_TRY(ctx->ws_reserve(need));
        ctx->ws_reset();
        wsA = ctx->ws_take<uint32_t>(bytes / 4);
        if (np & 1) wsB = ctx->ws_take<uint32_t>(bytes / 4);
    }
    uint32_t log_ns = 0;
    const uint32_t *src = d_data;
    for (int i = 0; i < np; ++i) {
        uint32_t *dst;
        if (np == 1) dst = d_data;  // the single tile is the whole polynomial: read fully into LDS before any store
        else {
            dst = ((np - 1 - i) % 2 == 0) ? d_data : wsA;
            if (i == 0 && dst == d_data) dst = wsB;  // odd pass count: never write the buffer being read
        }
        NttPass p;
        p.in = src;
        p.out = dst;
        p.log_m = (uint32_t)log_m;
        p.s = (uint32_t)sv[i];
        p.next_s = i + 1 < np ? (uint32_t)sv[i + 1] : 0u;
        p.log_ns = log_ns;
        uint32_t log_cols = (uint32_t)log_m - p.s;  // log2(m / R)
        p.log_t = std::min<uint32_t>((uint32_t)std::max(0, ctx->opt_ntt_tile_log), log_cols);
        // LDS budget: R * T * 36 B <= 144 KiB
        while (p.s + p.log_t > 12 && p.log_t > 0) --p.log_t;
        p.tiles_per_poly = 1u << (log_cols - p.log_t);
        p.lo = tb->d_lo;
        p.hi = tb->d_hi;
        p.lo_bits = (uint32_t)tb->lo_bits;
        p.clo = tb->d_clo;
        p.chi = tb->d_chi;
        p.scale = tb->d_scale;
        p.pre_coset = (!inverse && coset != nullptr && i == 0) ? 1u : 0u;
        p.post_coset = (inverse && coset != nullptr && i == np - 1) ? 1u : 0u;
        size_t nelem = (size_t)1 << (p.s + p.log_t), nhalf = std::max<size_t>(1, ((size_t)1 << p.s) / 2);
        size_t lds = (2 * nelem + (nelem + 3) / 4 + 2 * nhalf + (nhalf + 3) / 4) * 16;
        ZK_MAX_LDS(ctx, ntt_pass<U>, 160 * 1024);
        size_t grid = batch * p.tiles_per_poly;
        if (grid >= (1ull << 31)) return ZKHIP_ERR_RANGE;
        unsigned threads = (unsigned)std::min<size_t>(256, std::max<size_t>(64, nelem / 2));
        ZK_LAUNCH(ctx, "ntt_pass", ntt_pass<U>, dim3((unsigned)grid), dim3(threads), lds, p);
        src = dst;
        log_ns += p.s;
    }
    return 0;
}

int zk_ntt_run(zkhip_ctx *ctx, int curve, uint32_t *d_data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
               const uint64_t *coset) {
    if (log_m > 32) return ZKHIP_ERR_RANGE;
    if (curve == CURVE_BLS12_381) return ntt_run_t<BlsFrU>(ctx, curve, d_data, log_m, batch, omega, inverse, coset);
    if (curve == CURVE_BN254) return ntt_run_t<BnFrU>(ctx, curve, d_data, log_m, batch, omega, inverse, coset);
    return ZKHIP_ERR_INVALID;
}
