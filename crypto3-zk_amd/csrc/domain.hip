// Evaluation domains beyond the basic radix-2 one: what math::make_evaluation_domain(min_size) returns for a size that is
// no power of two (reductions/r1cs_to_qap.hpp:138-139, 229-230 call it with num_constraints + num_inputs + 1).
//
// crypto3-math is not vendored in the reference tree; its domain family and selection order are libfqfft's
// (get_evaluation_domain: basic, extended, step radix-2 at min_size, then at big + rounded_small); the tests hold every
// transform to its definition over the domain's point set.  Here the two
// composite domains are COMPOSITIONS of the radix-2 kernels of ntt.hip plus a few element-wise passes:
//
//   step_radix2(m = big + small), omega = primitive (2 big)-th root; points <omega^2> then omega <omega^(2 big/small)>.
//     forward (optionally on the coset g):   G = g^big, T[i] = (omega g)^i
//       c[i] = a[i] + G a[big + i] (i < small), a[i] otherwise         -> radix-2 transform of `big` points, coset g
//       e[j] = sum_{i = j mod small} T[i] (a[i] - G a[big + i] | a[i]) -> radix-2 transform of `small` points
//     inverse:  V0 = inverse transform of the first `big` values (coset g: times g^-i), V1 = of the last `small`
//       s[j] = sum_{i = j mod small, i >= small} T[i] V0[i]
//       u = (V1[j] - s[j]) T[j]^-1;   a[j] = (V0[j] + u) / 2,  a[big + j] = (V0[j] - u) / (2 G),  a[i] = V0[i] otherwise
//   extended_radix2(m = 2 n), omega = primitive n-th root, points <omega> then shift <omega>   (S = shift^n, G = g^n)
//     forward:  p0 = lo + G hi -> transform on the coset g;  p1 = lo + G S hi -> transform on the coset g shift
//     inverse:  W0, W1 the two inverse coset transforms;  lo = (W1 - S W0) / (1 - S),  hi = (W0 - W1) / (G (1 - S))
//
// Vectors are kept SPLIT (part 0: the first big / n values of every vector of a batch, contiguous; part 1: the rest) so
// that the radix-2 kernels run over whole batches in place; zkhip_domain_fft_dev converts from / to the contiguous layout.
// Everything is canonical Fr in HBM (4 x u64), tables are Montgomery: a product of the two is canonical again.
#include <algorithm>

#include "ctx.hpp"
#include "domain.hpp"
#include "fu.hpp"

using namespace zkhip;

struct DomTables {
    int curve, kind;
    size_t m, n0, n1;
    uint64_t omega[4], shift[4], coset[4];
    bool has_coset;
    uint32_t *d_T = nullptr;      // step: (omega g)^i, i < big        (Montgomery, 8 words each)
    uint32_t *d_Tinv = nullptr;   // step: (omega g)^-i, i < small
    uint32_t *d_consts = nullptr; // DC_* entries, Montgomery, 8 words each
    uint32_t *d_zinv = nullptr;   // has_coset: 1 / Z(g x_i) for part 0 by i mod nz, then one entry for part 1
    size_t nz = 1;
    uint64_t w0[4], w1[4];        // roots of the two sub-transforms
    uint64_t coset1[4];           // extended: g shift (the coset of the second sub-transform)
};
enum { DC_BASE = 0, DC_BASEINV, DC_G, DC_HALF, DC_HALF_GINV, DC_S, DC_GS, DC_K, DC_K_GINV, DC_Z_STEP, DC_Z_A, DC_Z_B, DC_Z1, DC_COUNT };

// ---- host-side field helpers (the same __host__ __device__ arithmetic the kernels use) --------------------------------
template <class U>
static void h_pow(const uint64_t *base, uint64_t e, uint64_t *out) {
    Fu<U> b = fu_from_canonical<U>(reinterpret_cast<const uint32_t *>(base)), r = Fu<U>::one();
    for (; e; e >>= 1) {
        if (e & 1) r = fu_mul(r, b);
        b = fu_mul(b, b);
    }
    fu_to_canonical<U>(reinterpret_cast<uint32_t *>(out), r);
}
template <class U>
static void h_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) {
    fu_to_canonical<U>(reinterpret_cast<uint32_t *>(out),
                       fu_mul(fu_from_canonical<U>(reinterpret_cast<const uint32_t *>(a)), fu_from_canonical<U>(reinterpret_cast<const uint32_t *>(b))));
}
static bool is_one(const uint64_t *a) { return a[0] == 1 && a[1] == 0 && a[2] == 0 && a[3] == 0; }
static size_t ceil_log2(size_t n) {
    size_t r = 0;
    while (((size_t)1 << r) < n) ++r;
    return r;
}

int zk_dom_two_adicity(int curve) { return curve == CURVE_BLS12_381 ? 32 : curve == CURVE_BN254 ? 28 : -1; }

bool zk_dom_choice(size_t min_size, size_t s, int *kind, size_t *m) {
    auto basic_ok = [&](size_t n) { return n > 1 && n == (size_t)1 << ceil_log2(n) && ceil_log2(n) <= s; };
    auto ext_ok = [&](size_t n) { return n > 1 && ceil_log2(n) == s + 1 && n == (size_t)1 << (s + 1); };
    auto step_ok = [&](size_t n) {
        if (n <= 1) return false;
        const size_t small = n - ((size_t)1 << (ceil_log2(n) - 1));
        return small == (size_t)1 << ceil_log2(small) && ceil_log2(n) <= s;
    };
    if (min_size <= 1 || ceil_log2(min_size) > 62) return false;
    const size_t big = (size_t)1 << (ceil_log2(min_size) - 1), small = min_size - big, rounded = big + ((size_t)1 << ceil_log2(small));
    for (size_t n : {min_size, rounded}) {
        if (basic_ok(n)) return *kind = ZKHIP_DOMAIN_BASIC_RADIX2, *m = n, true;
        if (ext_ok(n)) return *kind = ZKHIP_DOMAIN_EXTENDED_RADIX2, *m = n, true;
        if (step_ok(n)) return *kind = ZKHIP_DOMAIN_STEP_RADIX2, *m = n, true;
    }
    return false;
}

int zk_dom_parse(int curve, const zkhip_domain *d, ZkDomain *out) {
    if (!d || !out) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    ZkDomain z;
    z.kind = d->kind;
    z.m = d->m;
    memcpy(z.omega, d->omega, 32);
    memcpy(z.shift, d->shift, 32);
    if (d->m <= 1 || d->m > ((uint64_t)1 << 33)) return ZKHIP_ERR_RANGE;
    switch (d->kind) {
        case ZKHIP_DOMAIN_BASIC_RADIX2:
            if (d->m != (uint64_t)1 << ceil_log2(d->m)) return ZKHIP_ERR_INVALID;
            z.n0 = d->m;
            z.n1 = 0;
            break;
        case ZKHIP_DOMAIN_EXTENDED_RADIX2:
            if (d->m != (uint64_t)1 << ceil_log2(d->m)) return ZKHIP_ERR_INVALID;
            z.n0 = z.n1 = d->m / 2;
            break;
        case ZKHIP_DOMAIN_STEP_RADIX2:
            z.n0 = (size_t)1 << (ceil_log2(d->m) - 1);
            z.n1 = d->m - z.n0;
            if (z.n1 != (size_t)1 << ceil_log2(z.n1)) return ZKHIP_ERR_INVALID;  // step_radix2(): expected small_m == 1ul<<log2(small_m)
            break;
        default: return ZKHIP_ERR_INVALID;
    }
    if (ceil_log2(z.n0) > 32) return ZKHIP_ERR_RANGE;
    *out = z;
    return ZKHIP_OK;
}

// 32-byte elements in global memory <-> lazy limbs (value unchanged: canonical stays canonical, Montgomery stays Montgomery)
template <class U>
ZK_D Fu<U> e_load(const uint32_t *p, size_t i) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p) + 2 * i;
    const uint4 a = q[0], b = q[1];
    const uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fu_unpack<U>(s);
}
template <class U>
ZK_D void e_store(uint32_t *p, size_t i, const Fu<U> &x) {  // x normalised, < 2^256
    uint32_t s[8];
    fu_pack<U>(s, x);
    uint4 *q = reinterpret_cast<uint4 *>(p) + 2 * i;
    q[0] = make_uint4(s[0], s[1], s[2], s[3]);
    q[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
template <class U>
ZK_D Fu<U> addm(const Fu<U> &a, const Fu<U> &b) { return fu_cond_sub_p(fu_add(a, b)); }  // a, b < p -> a + b mod p
template <class U>
ZK_D Fu<U> mulm(const Fu<U> &a, const Fu<U> &b) { return fu_cond_sub_p(fu_mul(a, b)); }  // canonical x Montgomery -> canonical < p

// ---- table construction -----------------------------------------------------------------------------------------------
// in: omega, shift, coset (canonical; coset = 1 when the transform has none)
template <class U>
__global__ void dom_setup(int kind, const uint32_t *__restrict__ in, uint64_t n0, uint64_t n1, uint32_t *__restrict__ consts) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    auto powu = [](Fu<U> b, uint64_t e) {
        Fu<U> r = Fu<U>::one();
        for (; e; e >>= 1) {
            if (e & 1) r = fu_mul_call(r, b);
            b = fu_mul_call(b, b);
        }
        return r;
    };
    auto put = [&](int slot, const Fu<U> &x) { e_store<U>(consts, slot, fu_cond_sub_p(x)); };
    const Fu<U> w = fu_from_canonical<U>(in), sh = fu_from_canonical<U>(in + 8), g = fu_from_canonical<U>(in + 16), one = Fu<U>::one();
    const Fu<U> half = fu_inv(fu_add(one, one));
    const Fu<U> G = powu(g, n0), Ginv = fu_inv(G);
    put(DC_G, G);
    put(DC_HALF, half);
    put(DC_HALF_GINV, fu_mul_call(half, Ginv));
    if (kind == ZKHIP_DOMAIN_STEP_RADIX2) {
        const Fu<U> base = fu_mul_call(w, g);
        put(DC_BASE, base);
        put(DC_BASEINV, fu_inv(base));
        // divide_by_z_on_coset: Z(g omega^(2i)) = (g^big - 1)(g^small omega^(2 small i) - omega^small), compr = big / small values
        const Fu<U> Z0 = fu_sub<4>(G, one), w_sm = powu(w, n1);
        put(DC_Z_STEP, powu(w, 2 * n1));
        put(DC_Z_A, fu_mul_call(powu(g, n1), Z0));
        put(DC_Z_B, fu_mul_call(w_sm, Z0));
        // Z(g omega x), x in <omega_small>: ((g omega)^big - 1)((g omega)^small - omega^small)
        const Fu<U> gw = fu_mul_call(g, w);
        put(DC_Z1, fu_inv(fu_mul_call(fu_sub<4>(powu(gw, n0), one), fu_sub<4>(powu(gw, n1), w_sm))));
    } else {  // extended
        const Fu<U> S = powu(sh, n0), k = fu_inv(fu_sub<4>(one, S));
        put(DC_S, S);
        put(DC_GS, fu_mul_call(G, S));
        put(DC_K, k);
        put(DC_K_GINV, fu_mul_call(k, Ginv));
        // Z(x) = (x^n - 1)(x^n - S): constant on each half of the coset
        const Fu<U> GS = fu_mul_call(G, S);
        put(DC_Z_A, fu_inv(fu_mul_call(fu_sub<4>(G, one), fu_sub<4>(G, S))));      // part 0: x^n = G
        put(DC_Z1, fu_inv(fu_mul_call(fu_sub<4>(GS, one), fu_sub<4>(GS, S))));     // part 1: x^n = G S
    }
}
// out[i] = base^i, i < count (Montgomery, canonical representative)
template <class U>
__global__ __launch_bounds__(256) void dom_pow_table(const uint32_t *__restrict__ consts, int slot, uint64_t count, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fu<U> b = e_load<U>(consts, slot), r = Fu<U>::one();
    for (uint64_t e = i; e; e >>= 1) {
        if (e & 1) r = fu_mul(r, b);
        b = fu_mul(b, b);
    }
    e_store<U>(out, i, fu_cond_sub_p(r));
}
// step: zinv[j] = 1 / (A step^j - B), j < nz;  zinv[nz] = Z1 (already inverted).  extended: zinv[0] = DC_Z_A, zinv[1] = DC_Z1
template <class U>
__global__ __launch_bounds__(64) void dom_zinv_table(int kind, const uint32_t *__restrict__ consts, uint64_t nz, uint32_t *__restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nz) return;
    if (j == nz) {
        e_store<U>(out, nz, e_load<U>(consts, DC_Z1));
        return;
    }
    if (kind != ZKHIP_DOMAIN_STEP_RADIX2) {
        e_store<U>(out, 0, e_load<U>(consts, DC_Z_A));
        return;
    }
    Fu<U> b = e_load<U>(consts, DC_Z_STEP), r = Fu<U>::one();
    for (uint64_t e = j; e; e >>= 1) {
        if (e & 1) r = fu_mul_call(r, b);
        b = fu_mul_call(b, b);
    }
    const Fu<U> z = fu_sub<2>(fu_mul_call(e_load<U>(consts, DC_Z_A), r), e_load<U>(consts, DC_Z_B));
    e_store<U>(out, j, fu_cond_sub_p(fu_inv(z)));
}

void zk_dom_free_tables(zkhip_ctx *ctx) {
    for (DomTables *t : ctx->dom_tables) {
        (void)hipFree(t->d_T);
        (void)hipFree(t->d_Tinv);
        (void)hipFree(t->d_consts);
        (void)hipFree(t->d_zinv);
        delete t;
    }
    ctx->dom_tables.clear();
    (void)hipFree(ctx->dom_ws);
    ctx->dom_ws = nullptr;
    ctx->dom_ws_cap = 0;
}

template <class U>
static int dom_get_tables(zkhip_ctx *ctx, int curve, const ZkDomain &d, const uint64_t *coset, DomTables **out) {
    for (DomTables *t : ctx->dom_tables)
        if (t->curve == curve && t->kind == d.kind && t->m == d.m && t->has_coset == (coset != nullptr) && memcmp(t->omega, d.omega, 32) == 0 &&
            (d.kind != ZKHIP_DOMAIN_EXTENDED_RADIX2 || memcmp(t->shift, d.shift, 32) == 0) && (!coset || memcmp(t->coset, coset, 32) == 0)) {
            *out = t;
            return 0;
        }
    // the roots must be what the domain says they are: anything else transforms over another point set
    uint64_t t1[4], t2[4];
    if (d.kind == ZKHIP_DOMAIN_STEP_RADIX2) {
        h_pow<U>(d.omega, d.n0, t1);  // omega^big = -1 <=> primitive (2 big)-th root
        h_mul<U>(t1, t1, t2);
        if (is_one(t1) || !is_one(t2)) {
            ctx->last_error = "step_radix2 domain: omega is not a primitive 2^" + std::to_string(ceil_log2(d.n0) + 1) + "-th root of unity";
            return ZKHIP_ERR_INVALID;
        }
    } else {
        h_pow<U>(d.shift, d.n0, t1);  // the two halves are disjoint iff shift^n != 1 (the sub-transforms check omega)
        if (is_one(t1)) {
            ctx->last_error = "extended_radix2 domain: the shift lies in the subgroup";
            return ZKHIP_ERR_INVALID;
        }
    }
    if (ctx->dom_tables.size() >= 8) {
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        DomTables *old = ctx->dom_tables.front();
        (void)hipFree(old->d_T);
        (void)hipFree(old->d_Tinv);
        (void)hipFree(old->d_consts);
        (void)hipFree(old->d_zinv);
        delete old;
        ctx->dom_tables.erase(ctx->dom_tables.begin());
    }
    // Built in a local owner and PUBLISHED only after the last launch has completed: an allocation, copy or launch that fails on
    // the way (out of memory on d_T, say) must not leave a half-built entry in the cache for the next call to hit.
    struct Building {
        DomTables *t = new DomTables();
        uint32_t *d_in = nullptr;
        ~Building() {
            (void)hipFree(d_in);
            if (t) {
                (void)hipFree(t->d_T);
                (void)hipFree(t->d_Tinv);
                (void)hipFree(t->d_consts);
                (void)hipFree(t->d_zinv);
                delete t;
            }
        }
    } building;
    DomTables *t = building.t;
    uint32_t *&d_in = building.d_in;
    t->curve = curve;
    t->kind = d.kind;
    t->m = d.m;
    t->n0 = d.n0;
    t->n1 = d.n1;
    memcpy(t->omega, d.omega, 32);
    memcpy(t->shift, d.shift, 32);
    t->has_coset = coset != nullptr;
    if (coset) memcpy(t->coset, coset, 32);
    const uint64_t one[4] = {1, 0, 0, 0};
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&d_in, 96));
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_consts, DC_COUNT * 32));
    ZK_HIP_CHECK(ctx, hipMemsetAsync(t->d_consts, 0, DC_COUNT * 32, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in, d.omega, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in + 8, d.kind == ZKHIP_DOMAIN_EXTENDED_RADIX2 ? d.shift : one, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in + 16, coset ? coset : one, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "dom_setup", dom_setup<U>, dim3(1), dim3(64), 0, d.kind, d_in, (uint64_t)d.n0, (uint64_t)d.n1, t->d_consts);
    if (d.kind == ZKHIP_DOMAIN_STEP_RADIX2) {
        const size_t compr = d.n0 / d.n1;
        h_mul<U>(d.omega, d.omega, t->w0);
        h_pow<U>(d.omega, 2 * compr, t->w1);
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_T, d.n0 * 32));
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_Tinv, d.n1 * 32));
        ZK_LAUNCH(ctx, "dom_setup", dom_pow_table<U>, dim3((unsigned)((d.n0 + 255) / 256)), dim3(256), 0, t->d_consts, (int)DC_BASE, (uint64_t)d.n0, t->d_T);
        ZK_LAUNCH(ctx, "dom_setup", dom_pow_table<U>, dim3((unsigned)((d.n1 + 255) / 256)), dim3(256), 0, t->d_consts, (int)DC_BASEINV, (uint64_t)d.n1, t->d_Tinv);
        t->nz = compr;
    } else {
        memcpy(t->w0, d.omega, 32);
        memcpy(t->w1, d.omega, 32);
        if (coset) h_mul<U>(coset, d.shift, t->coset1);
        else memcpy(t->coset1, d.shift, 32);
        t->nz = 1;
    }
    if (coset) {
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_zinv, (t->nz + 1) * 32));
        ZK_LAUNCH(ctx, "dom_setup", dom_zinv_table<U>, dim3((unsigned)((t->nz + 1 + 63) / 64)), dim3(64), 0, d.kind, t->d_consts, (uint64_t)t->nz, t->d_zinv);
    }
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dom_tables.push_back(t);
    building.t = nullptr;  // the cache owns it now; d_in goes with `building`
    *out = t;
    return 0;
}

// ---- step domain: column sums ------------------------------------------------------------------------------------------
struct StepArgs {
    uint32_t *p0;          // batch x big
    const uint32_t *p1;    // batch x small: the high coefficients (forward), null (inverse: plain T[i] V0[i] sums over i >= small)
    uint64_t big, small;
    const uint32_t *T, *consts;
    uint32_t *partial;     // batch x P x small
    uint32_t P, K, R;      // regime A: tiles of 256 K indices; regime B: R rows per block
};
// d_i = T[i] x_i with x_i = a[i] - G a[big + i] (i < small; also rewrites a[i] += G a[big + i]) or a[i]; inverse: 0 for i < small
template <class U>
ZK_D Fu<U> step_term(const StepArgs &s, size_t b, uint64_t i, const Fu<U> &G) {
    Fu<U> x = e_load<U>(s.p0, b * s.big + i);
    if (i < s.small) {
        if (s.p1 == nullptr) return Fu<U>::zero();
        const Fu<U> gh = mulm(e_load<U>(s.p1, b * s.small + i), G);
        e_store<U>(s.p0, b * s.big + i, addm(x, gh));
        x = fu_sub<2>(x, gh);
    }
    return mulm(x, e_load<U>(s.T, i));
}
// regime A (small <= 256): a block sums a tile of 256 K consecutive indices by column (= thread index mod small)
template <class U>
__global__ __launch_bounds__(256) void step_colsum_a(StepArgs s) {
    __shared__ uint32_t part[256 * U::L];
    const uint32_t t = threadIdx.x;
    const size_t b = blockIdx.y;
    const Fu<U> G = e_load<U>(s.consts, DC_G);
    Fu<U> acc = Fu<U>::zero();
    const uint64_t base = (uint64_t)blockIdx.x * 256 * s.K;
    for (uint32_t k = 0; k < s.K; ++k) {
        const uint64_t i = base + t + 256ull * k;
        if (i < s.big) acc = addm(acc, step_term<U>(s, b, i, G));
    }
    if (s.small < 256) {
#pragma unroll
        for (int i = 0; i < U::L; ++i) part[i * 256 + t] = acc.v[i];
        __syncthreads();
        for (uint32_t d = 128; d >= s.small; d >>= 1) {
            if (t < d) {
                Fu<U> o;
#pragma unroll
                for (int i = 0; i < U::L; ++i) o.v[i] = part[i * 256 + t + d];
                acc = addm(acc, o);
#pragma unroll
                for (int i = 0; i < U::L; ++i) part[i * 256 + t] = acc.v[i];
            }
            __syncthreads();
        }
    }
    if (t < s.small) e_store<U>(s.partial, (b * s.P + blockIdx.x) * s.small + t, acc);
}
// regime B (small >= 512): a block owns 256 columns and R rows of the (big / small) x small matrix
template <class U>
__global__ __launch_bounds__(256) void step_colsum_b(StepArgs s) {
    const size_t b = blockIdx.z;
    const uint64_t col = (uint64_t)blockIdx.x * 256 + threadIdx.x, rows = s.big / s.small;
    const Fu<U> G = e_load<U>(s.consts, DC_G);
    Fu<U> acc = Fu<U>::zero();
    const uint64_t r0 = (uint64_t)blockIdx.y * s.R, r1 = min(rows, r0 + s.R);
    for (uint64_t r = r0; r < r1; ++r) acc = addm(acc, step_term<U>(s, b, r * s.small + col, G));
    e_store<U>(s.partial, (b * s.P + blockIdx.y) * s.small + col, acc);
}
// out[b][col] = sum_p partial[b][p][col]
template <class U>
__global__ __launch_bounds__(256) void step_colsum_finish(const uint32_t *__restrict__ partial, uint32_t P, uint64_t small, uint32_t *__restrict__ out) {
    __shared__ uint32_t part[256 * U::L];
    const uint32_t t = threadIdx.x;
    const size_t b = blockIdx.y;
    Fu<U> acc = Fu<U>::zero();
    if (small >= 256) {
        const uint64_t col = (uint64_t)blockIdx.x * 256 + t;
        for (uint32_t p = 0; p < P; ++p) acc = addm(acc, e_load<U>(partial, (b * P + p) * small + col));
        e_store<U>(out, b * small + col, acc);
        return;
    }
    const uint32_t col = t & ((uint32_t)small - 1), lanes = 256 / (uint32_t)small;
    for (uint32_t p = t / (uint32_t)small; p < P; p += lanes) acc = addm(acc, e_load<U>(partial, (b * P + p) * small + col));
#pragma unroll
    for (int i = 0; i < U::L; ++i) part[i * 256 + t] = acc.v[i];
    __syncthreads();
    for (uint32_t d = 128; d >= small; d >>= 1) {
        if (t < d) {
            Fu<U> o;
#pragma unroll
            for (int i = 0; i < U::L; ++i) o.v[i] = part[i * 256 + t + d];
            acc = addm(acc, o);
#pragma unroll
            for (int i = 0; i < U::L; ++i) part[i * 256 + t] = acc.v[i];
        }
        __syncthreads();
    }
    if (t < small) e_store<U>(out, b * small + t, acc);
}
// the last step of the inverse transform, j < small (see the file header)
template <class U>
__global__ __launch_bounds__(256) void step_inv_post(uint32_t *__restrict__ p0, uint32_t *__restrict__ p1, uint64_t big, uint64_t small,
                                                     const uint32_t *__restrict__ colsum, const uint32_t *__restrict__ Tinv,
                                                     const uint32_t *__restrict__ consts) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (j >= small) return;
    const Fu<U> v0 = e_load<U>(p0, b * big + j), v1 = e_load<U>(p1, b * small + j), s = e_load<U>(colsum, b * small + j);
    const Fu<U> u = mulm(fu_sub<2>(v1, s), e_load<U>(Tinv, j));
    e_store<U>(p0, b * big + j, mulm(fu_add(v0, u), e_load<U>(consts, DC_HALF)));
    e_store<U>(p1, b * small + j, mulm(fu_sub<2>(v0, u), e_load<U>(consts, DC_HALF_GINV)));
}
// ---- extended domain ---------------------------------------------------------------------------------------------------
// forward: (lo, hi) -> (lo + G hi, lo + G S hi);  inverse: (W0, W1) -> ((W1 - S W0) k, (W0 - W1) k / G)
template <class U>
__global__ __launch_bounds__(256) void ext_mix(uint32_t *__restrict__ p0, uint32_t *__restrict__ p1, uint64_t total, int inverse,
                                               const uint32_t *__restrict__ consts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Fu<U> a = e_load<U>(p0, i), c = e_load<U>(p1, i);
    if (!inverse) {
        e_store<U>(p0, i, addm(a, mulm(c, e_load<U>(consts, DC_G))));
        e_store<U>(p1, i, addm(a, mulm(c, e_load<U>(consts, DC_GS))));
    } else {
        const Fu<U> sw0 = mulm(a, e_load<U>(consts, DC_S));
        e_store<U>(p0, i, mulm(fu_sub<2>(c, sw0), e_load<U>(consts, DC_K)));
        e_store<U>(p1, i, mulm(fu_sub<2>(a, c), e_load<U>(consts, DC_K_GINV)));
    }
}

// contiguous <-> split: vec[b][i] (m each) <-> p0[b][i] (i < n0), p1[b][i - n0]
__global__ __launch_bounds__(256) void dom_split_copy(uint4 *__restrict__ vec, uint4 *__restrict__ p0, uint4 *__restrict__ p1, uint64_t m, uint64_t n0,
                                                      uint64_t total, int to_split) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const uint64_t b = e / m, i = e % m;
    uint4 *part = i < n0 ? p0 + 2 * (b * n0 + i) : p1 + 2 * (b * (m - n0) + (i - n0));
    if (to_split) {
        part[0] = vec[2 * e];
        part[1] = vec[2 * e + 1];
    } else {
        vec[2 * e] = part[0];
        vec[2 * e + 1] = part[1];
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
struct StepGeom {
    bool a;          // regime A (small <= 256)
    uint32_t P, K, R;
};
static StepGeom step_geom(size_t big, size_t small) {
    StepGeom g;
    g.a = small <= 256;
    g.K = g.R = 1;
    if (g.a) {
        g.K = (uint32_t)std::max<size_t>(1, (big + 256 * 1024 - 1) / (256 * 1024));
        g.P = (uint32_t)((big + 256ull * g.K - 1) / (256ull * g.K));
    } else {
        const size_t rows = big / small, want = std::max<size_t>(1, (512 * 256) / small);  // >= 512 blocks when the rows allow
        g.P = (uint32_t)std::min(rows, want);
        g.R = (uint32_t)((rows + g.P - 1) / g.P);
        g.P = (uint32_t)((rows + g.R - 1) / g.R);
    }
    return g;
}

size_t zk_dom_scratch_elems(const ZkDomain &d, size_t batch) {
    if (d.kind != ZKHIP_DOMAIN_STEP_RADIX2) return 8;
    const StepGeom g = step_geom(d.n0, d.n1);
    return batch * ((size_t)g.P * d.n1 + d.n1) + 8;
}

template <class U>
static int dom_fft_split_t(zkhip_ctx *ctx, int curve, const ZkDomain &d, uint32_t *p0, uint32_t *p1, size_t batch, int inverse, const uint64_t *coset,
                           uint32_t *scratch) {
    if (d.kind == ZKHIP_DOMAIN_BASIC_RADIX2) return zk_ntt_run(ctx, curve, p0, ceil_log2(d.m), batch, d.omega, inverse, coset);
    DomTables *t = nullptr;
    ZK_TRY(dom_get_tables<U>(ctx, curve, d, coset, &t));
    const size_t l0 = ceil_log2(d.n0), l1 = ceil_log2(d.n1);
    if (batch >= (1u << 16)) return ZKHIP_ERR_RANGE;
    if (d.kind == ZKHIP_DOMAIN_EXTENDED_RADIX2) {
        const uint64_t total = (uint64_t)batch * d.n0;
        const unsigned grid = (unsigned)((total + 255) / 256);
        if (!inverse) ZK_LAUNCH(ctx, "dom_mix", ext_mix<U>, dim3(grid), dim3(256), 0, p0, p1, total, 0, t->d_consts);
        ZK_TRY(zk_ntt_run(ctx, curve, p0, l0, batch, t->w0, inverse, coset));
        ZK_TRY(zk_ntt_run(ctx, curve, p1, l1, batch, t->w1, inverse, t->coset1));
        if (inverse) ZK_LAUNCH(ctx, "dom_mix", ext_mix<U>, dim3(grid), dim3(256), 0, p0, p1, total, 1, t->d_consts);
        return 0;
    }
    // step
    const StepGeom g = step_geom(d.n0, d.n1);
    uint32_t *partial = scratch, *colsum = scratch + batch * (size_t)g.P * d.n1 * 8;
    StepArgs s;
    s.p0 = p0;
    s.big = d.n0;
    s.small = d.n1;
    s.T = t->d_T;
    s.consts = t->d_consts;
    s.partial = partial;
    s.P = g.P, s.K = g.K, s.R = g.R;
    auto colsums = [&](const uint32_t *hi, uint32_t *dst) -> int {
        s.p1 = hi;
        if (g.a) ZK_LAUNCH(ctx, "dom_colsum", step_colsum_a<U>, dim3(g.P, (unsigned)batch), dim3(256), 0, s);
        else ZK_LAUNCH(ctx, "dom_colsum", step_colsum_b<U>, dim3((unsigned)(d.n1 / 256), g.P, (unsigned)batch), dim3(256), 0, s);
        ZK_LAUNCH(ctx, "dom_colsum", step_colsum_finish<U>, dim3((unsigned)std::max<size_t>(1, d.n1 / 256), (unsigned)batch), dim3(256), 0, partial, g.P,
                  (uint64_t)d.n1, dst);
        return 0;
    };
    if (!inverse) {
        ZK_TRY(colsums(p1, p1));  // e overwrites the high coefficients, which the sums have consumed
        ZK_TRY(zk_ntt_run(ctx, curve, p0, l0, batch, t->w0, 0, coset));
        ZK_TRY(zk_ntt_run(ctx, curve, p1, l1, batch, t->w1, 0, nullptr));
    } else {
        ZK_TRY(zk_ntt_run(ctx, curve, p0, l0, batch, t->w0, 1, coset));
        ZK_TRY(zk_ntt_run(ctx, curve, p1, l1, batch, t->w1, 1, nullptr));
        ZK_TRY(colsums(nullptr, colsum));
        ZK_LAUNCH(ctx, "dom_mix", step_inv_post<U>, dim3((unsigned)((d.n1 + 255) / 256), (unsigned)batch), dim3(256), 0, p0, p1, (uint64_t)d.n0, (uint64_t)d.n1,
                  colsum, t->d_Tinv, t->d_consts);
    }
    return 0;
}

int zk_dom_fft_split(zkhip_ctx *ctx, int curve, const ZkDomain &d, uint32_t *p0, uint32_t *p1, size_t batch, int inverse, const uint64_t *coset,
                     uint32_t *scratch) {
    if (batch == 0) return 0;
    if (curve == CURVE_BLS12_381) return dom_fft_split_t<BlsFrU>(ctx, curve, d, p0, p1, batch, inverse, coset, scratch);
    if (curve == CURVE_BN254) return dom_fft_split_t<BnFrU>(ctx, curve, d, p0, p1, batch, inverse, coset, scratch);
    return ZKHIP_ERR_INVALID;
}

// 1 / Z on the coset g * domain: *d_zinv holds nz entries for part 0 (entry i mod nz) followed by the entry of part 1
template <class U>
__global__ void dom_zinv_basic(const uint32_t *__restrict__ coset_c, uint64_t m, uint32_t *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Fu<U> g = fu_from_canonical<U>(coset_c), r = Fu<U>::one();
    for (uint64_t e = m; e; e >>= 1) {
        if (e & 1) r = fu_mul_call(r, g);
        g = fu_mul_call(g, g);
    }
    const Fu<U> z = fu_cond_sub_p(fu_inv(fu_sub<4>(r, Fu<U>::one())));
    e_store<U>(out, 0, z);
    e_store<U>(out, 1, z);
}
template <class U>
static int dom_zinv_t(zkhip_ctx *ctx, int curve, const ZkDomain &d, const uint64_t *coset, const uint32_t **d_zinv, size_t *nz) {
    if (d.kind != ZKHIP_DOMAIN_BASIC_RADIX2) {
        DomTables *t = nullptr;
        ZK_TRY(dom_get_tables<U>(ctx, curve, d, coset, &t));
        *d_zinv = t->d_zinv;
        *nz = t->nz;
        return 0;
    }
    // basic: Z(g x) = g^m - 1 everywhere; cached as a table set of its own (kind basic, with coset)
    for (DomTables *t : ctx->dom_tables)
        if (t->curve == curve && t->kind == d.kind && t->m == d.m && t->has_coset && memcmp(t->coset, coset, 32) == 0) {
            *d_zinv = t->d_zinv;
            *nz = 1;
            return 0;
        }
    DomTables *t = new DomTables();
    t->curve = curve;
    t->kind = d.kind;
    t->m = d.m;
    t->n0 = d.m;
    t->n1 = 0;
    memcpy(t->omega, d.omega, 32);
    t->has_coset = true;
    memcpy(t->coset, coset, 32);
    ctx->dom_tables.push_back(t);
    uint32_t *d_in = nullptr;
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&d_in, 32));
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_zinv, 64));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_in, coset, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "dom_setup", dom_zinv_basic<U>, dim3(1), dim3(64), 0, d_in, (uint64_t)d.m, t->d_zinv);
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_in);
    *d_zinv = t->d_zinv;
    *nz = 1;
    return 0;
}
int zk_dom_zinv(zkhip_ctx *ctx, int curve, const ZkDomain &d, const uint64_t *coset, const uint32_t **d_zinv, size_t *nz) {
    if (!coset) return ZKHIP_ERR_INVALID;
    if (curve == CURVE_BLS12_381) return dom_zinv_t<BlsFrU>(ctx, curve, d, coset, d_zinv, nz);
    if (curve == CURVE_BN254) return dom_zinv_t<BnFrU>(ctx, curve, d, coset, d_zinv, nz);
    return ZKHIP_ERR_INVALID;
}

// ---- every Lagrange polynomial of a domain at one point (key generation: r1cs_to_qap.hpp:152-153) ----------------------------------
// out[i] = c w^i / (t - w^i)   (unit == 0: c l_i(t) n / (t^n - 1) folded into c by the host)
// out[i] = c / (w^i - t)       (unit == 1: the step domain's denominators x^small - omega^small)
// consts: w, w^-1, t, c in Montgomery form.  A lane takes LAG_CHUNK consecutive i: one power, running products, ONE inversion
// (Montgomery's trick), then back down the chunk.  Canonical output.
static constexpr uint32_t LAG_CHUNK = 16;
template <class U>
__global__ __launch_bounds__(64) void dom_lagrange(const uint32_t *__restrict__ consts, uint64_t n, int unit, uint32_t *__restrict__ out, uint64_t out_off) {
    const uint64_t lo = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * LAG_CHUNK;
    if (lo >= n) return;
    const uint32_t cnt = (uint32_t)(n - lo < LAG_CHUNK ? n - lo : LAG_CHUNK);
    const Fu<U> w = e_load<U>(consts, 0), winv = e_load<U>(consts, 1), t = e_load<U>(consts, 2), c = e_load<U>(consts, 3);
    Fu<U> x = Fu<U>::one(), b = w;
    for (uint64_t e = lo; e; e >>= 1) {
        if (e & 1) x = fu_mul_call(x, b);
        b = fu_mul_call(b, b);
    }
    x = fu_cond_sub_p(x);
    Fu<U> pre[LAG_CHUNK];
    Fu<U> acc = Fu<U>::one();
    for (uint32_t k = 0; k < cnt; ++k) {
        pre[k] = acc;
        const Fu<U> den = unit ? fu_sub<2>(x, t) : fu_sub<2>(t, x);
        acc = fu_mul_call(acc, den);
        if (k + 1 < cnt) x = fu_cond_sub_p(fu_mul_call(x, w));
    }
    Fu<U> inv = fu_inv(acc);  // x is now w^(lo + cnt - 1)
    Fu<U> plain = Fu<U>::zero();
    plain.v[0] = 1;
    for (uint32_t k = cnt; k-- > 0;) {
        const Fu<U> den = unit ? fu_sub<2>(x, t) : fu_sub<2>(t, x);
        Fu<U> r = fu_mul_call(fu_mul_call(inv, pre[k]), c);  // c / den_k
        if (!unit) r = fu_mul_call(r, x);
        e_store<U>(out, out_off + lo + k, fu_cond_sub_p(fu_mul_call(r, plain)));  // out of Montgomery form
        inv = fu_mul_call(inv, den);
        x = fu_cond_sub_p(fu_mul_call(x, winv));
    }
}
// out[i] *= tab[i mod nz]  (canonical x canonical -> canonical), i < n
template <class U>
__global__ __launch_bounds__(256) void dom_scale_by_table(uint32_t *__restrict__ out, uint64_t n, const uint32_t *__restrict__ tab, uint64_t nz) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    e_store<U>(out, i, fu_cond_sub_p(fu_mul(fu_mul(e_load<U>(out, i), e_load<U>(tab, i % nz)), Fu<U>::r2())));
}

// host side of zkhip_domain_lagrange_dev: field arithmetic on canonical limbs with the kernels' own code
template <class U>
struct HF {
    Fu<U> v;  // Montgomery
    static HF from(const uint64_t *c) { return {fu_from_canonical<U>(reinterpret_cast<const uint32_t *>(c))}; }
    static HF one() { return {Fu<U>::one()}; }
    static HF u64(uint64_t x) {
        const uint64_t c[4] = {x, 0, 0, 0};
        return from(c);
    }
    HF operator*(const HF &o) const { return {fu_cond_sub_p(fu_mul(v, o.v))}; }
    HF operator-(const HF &o) const { return {fu_canon(fu_sub<4>(v, o.v))}; }
    HF inv() const { return {fu_cond_sub_p(fu_inv(v))}; }
    HF pow(uint64_t e) const {
        HF r = one(), b = *this;
        for (; e; e >>= 1) {
            if (e & 1) r = r * b;
            b = b * b;
        }
        return r;
    }
    bool is_zero() const { return fu_canon(v).limbs_zero(); }
    void store_mont(uint32_t *dst) const {  // 8 words, canonical representative of the Montgomery form
        fu_pack<U>(dst, fu_canon(v));
    }
};

// one dom_lagrange launch: out[off + i] = c w^i / (t - w^i) or c / (w^i - t)
template <class U>
static int lagrange_launch(zkhip_ctx *ctx, uint32_t *d_consts, uint32_t *h_consts, int slot, const HF<U> &w, const HF<U> &t, const HF<U> &c, size_t n, int unit,
                           uint32_t *d_out, size_t off) {
    uint32_t *h = h_consts + slot * 32, *d = d_consts + slot * 32;
    w.store_mont(h);
    w.inv().store_mont(h + 8);
    t.store_mont(h + 16);
    c.store_mont(h + 24);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d, h, 128, hipMemcpyHostToDevice, ctx->stream));
    const size_t lanes = (n + LAG_CHUNK - 1) / LAG_CHUNK;
    ZK_LAUNCH(ctx, "dom_lagrange", dom_lagrange<U>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, d, (uint64_t)n, unit, d_out, (uint64_t)off);
    return ZKHIP_OK;
}

template <class U>
static int dom_lagrange_t(zkhip_ctx *ctx, const ZkDomain &d, const uint64_t *t_c, uint32_t *d_out) {
    typedef HF<U> F;
    const F one = F::one(), t = F::from(t_c), w = F::from(d.omega);
    // constants of up to four launches, staged in page-able host memory that must outlive the copies: kept in the context
    ctx->lagrange_stage.assign(4 * 32, 0u);
    uint32_t *h = ctx->lagrange_stage.data(), *dc = nullptr;
    ZK_TRY(ctx->ws_reserve(zkhip_ctx::ws_round(4 * 128) + zkhip_ctx::ws_round((d.n1 ? d.n0 / d.n1 : 1) * 32)));
    ctx->ws_reset();
    dc = ctx->ws_take<uint32_t>(4 * 32);
    auto basic = [&](int slot, const F &root, const F &at, const F &scale, size_t n, size_t off) {
        // scale * l_i(at) over <root>: l_i = (at^n - 1) root^i / (n (at - root^i))
        const F c = (at.pow(n) - one) * F::u64(n).inv() * scale;
        return lagrange_launch<U>(ctx, dc, h, slot, root, at, c, n, 0, d_out, off);
    };
    int rc = ZKHIP_OK;
    if (d.kind == ZKHIP_DOMAIN_BASIC_RADIX2) {
        if ((t.pow(d.m) - one).is_zero()) return ZKHIP_ERR_INVALID;
        rc = basic(0, w, t, one, d.m, 0);
    } else if (d.kind == ZKHIP_DOMAIN_EXTENDED_RADIX2) {
        const size_t n = d.n0;
        const F sh = F::from(d.shift), t_n = t.pow(n), s_n = sh.pow(n), ood = (s_n - one).inv();
        if ((t_n - one).is_zero() || (t_n - s_n).is_zero()) return ZKHIP_ERR_INVALID;
        rc = basic(0, w, t, (s_n - t_n) * ood, n, 0);
        if (rc == ZKHIP_OK) rc = basic(1, w, t * sh.inv(), (t_n - one) * ood, n, n);
    } else {
        // big part: l_i(t) over <omega^2> times (t^small - omega^small) / (x_i^small - omega^small): the denominator takes big / small
        // distinct values; small part: l_i(t / omega) over <omega^(2 big / small)> times (t^big - 1) / (omega^big - 1)
        const size_t big = d.n0, small = d.n1, compr = big / small;
        const F big_w = w * w, small_w = w.pow(2 * compr), w_sm = w.pow(small), L0 = t.pow(small) - w_sm;
        if ((t.pow(big) - one).is_zero() || L0.is_zero()) return ZKHIP_ERR_INVALID;
        uint32_t *d_dinv = ctx->ws_take<uint32_t>(compr * 8);
        rc = lagrange_launch<U>(ctx, dc, h, 0, big_w.pow(small), w_sm, one, compr, 1, d_dinv, 0);  // 1 / (step^j - omega^small)
        if (rc == ZKHIP_OK) rc = basic(1, big_w, t, L0, big, 0);
        if (rc == ZKHIP_OK)
            ZK_LAUNCH(ctx, "dom_lagrange", dom_scale_by_table<U>, dim3((unsigned)((big + 255) / 256)), dim3(256), 0, d_out, (uint64_t)big, d_dinv, (uint64_t)compr);
        const F L1 = (t.pow(big) - one) * (w.pow(big) - one).inv();
        if (rc == ZKHIP_OK) rc = basic(2, small_w, t * w.inv(), L1, small, big);
    }
    if (rc != ZKHIP_OK) return rc;
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the staged constants and the workspace table may be reused after return
    return ZKHIP_OK;
}

extern "C" {

int zkhip_domain_choice(int curve, size_t min_size, int *kind, size_t *m) {
    if (!kind || !m) return ZKHIP_ERR_INVALID;
    const int s = zk_dom_two_adicity(curve);
    if (s < 0) return ZKHIP_ERR_INVALID;
    return zk_dom_choice(min_size, (size_t)s, kind, m) ? ZKHIP_OK : ZKHIP_ERR_RANGE;
}

int zkhip_domain_lagrange_dev(zkhip_ctx *ctx, int curve, const zkhip_domain *dom, const uint64_t *t, void *d_out) {
    if (!ctx || !dom || !t || !d_out) return ZKHIP_ERR_INVALID;
    ZkDomain d;
    ZK_TRY(zk_dom_parse(curve, dom, &d));
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (curve == CURVE_BLS12_381) return dom_lagrange_t<BlsFrU>(ctx, d, t, (uint32_t *)d_out);
    return dom_lagrange_t<BnFrU>(ctx, d, t, (uint32_t *)d_out);
}

int zkhip_domain_fft_dev(zkhip_ctx *ctx, int curve, const zkhip_domain *dom, void *d_data, size_t batch, int inverse, const uint64_t *coset_gen) {
    if (!ctx || !dom || (batch && !d_data)) return ZKHIP_ERR_INVALID;
    ZkDomain d;
    ZK_TRY(zk_dom_parse(curve, dom, &d));
    if (batch == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (d.kind == ZKHIP_DOMAIN_BASIC_RADIX2) return zk_ntt_run(ctx, curve, (uint32_t *)d_data, ceil_log2(d.m), batch, d.omega, inverse, coset_gen);
    // split copy of the batch + the step domain's partial sums, in a scratch buffer the context keeps
    const size_t need = (batch * d.m + zk_dom_scratch_elems(d, batch)) * 32;
    if (need > ctx->dom_ws_cap) {
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->dom_ws);
        ctx->dom_ws = nullptr;
        ctx->dom_ws_cap = 0;
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&ctx->dom_ws, need + (need >> 3)));
        ctx->dom_ws_cap = need + (need >> 3);
    }
    uint32_t *p0 = (uint32_t *)ctx->dom_ws, *p1 = p0 + batch * d.n0 * 8, *scratch = p1 + batch * d.n1 * 8;
    const uint64_t total = (uint64_t)batch * d.m;
    if (total >= ((uint64_t)1 << 39)) return ZKHIP_ERR_RANGE;
    const unsigned grid = (unsigned)((total + 255) / 256);
    ZK_LAUNCH(ctx, "dom_split", dom_split_copy, dim3(grid), dim3(256), 0, (uint4 *)d_data, (uint4 *)p0, (uint4 *)p1, (uint64_t)d.m, (uint64_t)d.n0, total, 1);
    ZK_TRY(zk_dom_fft_split(ctx, curve, d, p0, p1, batch, inverse, coset_gen, scratch));
    ZK_LAUNCH(ctx, "dom_split", dom_split_copy, dim3(grid), dim3(256), 0, (uint4 *)d_data, (uint4 *)p0, (uint4 *)p1, (uint64_t)d.m, (uint64_t)d.n0, total, 0);
    return ZKHIP_OK;
}

}  // extern "C"
