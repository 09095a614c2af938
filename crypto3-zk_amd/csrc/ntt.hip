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
// conversion pass exists.  Every per-element factor is ONE product against a table entry read next to the element:
//   tw[i]     the factor pass i applies while storing (the next pass's omega_(Ns R)^(k t), times 1/m on the first
//             boundary of an inverse transform), m entries in output order, read next to the element it multiplies;
//   pre/post  g^i (forward coset, first load) / (1/m) g^-i (inverse coset, last store);
//   stage     omega_R^q, q < R/2, staged in LDS per workgroup without any arithmetic.
// The tables are built once per (size, omega, direction, coset, radix plan) from a two-level power table and cached
// in the context.  A last pass with nothing to multiply by (forward, no coset) brings its lazy values (< 64p) back
// below p by a 7-bit quotient estimate instead of a product (fu_reduce_small).
// Two radix-2 stages run per LDS round trip (a thread holds the four rows i0 + {0, h, 2h, 3h}: the same four products
// as two separate stages, half the LDS traffic and barriers).
#include <algorithm>

#include "ctx.hpp"
#include "fu.hpp"

using namespace zkhip;

static constexpr int NTT_MAX_PASSES = 32;  // log_m <= 32 at radix 2
static constexpr uint32_t NTT_PAD = 0;  // extra elements per LDS row (see lds_get)

struct NttTables {
    int curve;
    size_t log_m;
    int inverse;
    bool has_coset;
    int smax, tile_log;  // the radix plan the per-pass tables were built for
    uint64_t omega[4], coset[4];
    int lo_bits;
    uint32_t *d_tw[NTT_MAX_PASSES] = {};     // store factors of pass i (i < passes - 1): m x 8 u32, Montgomery, saturated limbs
    uint32_t *d_stage[NTT_MAX_PASSES] = {};  // omega_R^q, q < R/2, Fu form (SL words)
    uint32_t *d_prepost = nullptr;           // coset: g^i (forward) or (1/m) g^-i (inverse), m x 8 u32
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
    const uint32_t *stage;      // omega_R^q, q < R/2 (Fu form)
    const uint32_t *tw;         // store factors in output order (not on the last pass)
    const uint32_t *pre;        // first pass of a forward coset transform: g^index, else null
    const uint32_t *post;       // last pass of an inverse coset transform: (1/m) g^-index, else null
    const uint32_t *scale;      // single-pass inverse transform: 1/m (Fu form), else null
    uint32_t in_lazy, out_lazy; // the vector read / written is in limb form (an intermediate between passes), else canonical
    size_t planeb;              // words from a limb-form data buffer's base to its limb-8 plane (batch x m x 8)
    // EXTENSION mode (ntt_pass<U, PB, true>; zk_ntt_extend): "polynomial" w of the launch is coset j = w % ext_k1 + 1 of source polynomial
    // w / ext_k1: the first pass reads the source's COEFFICIENTS times g_j^index (table j of ext_pre), the last pass stores its value for
    // index oi at out[(source << (log_m + ext_log_k)) + (oi << ext_log_k) + j] -- the natural order of the K n-point domain
    uint32_t poly_base;         // the launch's first polynomial (an odd batch runs as pairs + one single launch per pass)
    uint32_t ext_k1, ext_log_k;
    const uint32_t *ext_pre;    // ext_k1 tables of g_j^i, i < m, limb form (m x 36 B each)
};

// parameters of the table-building kernel: everything ntt_pass used to look up on the fly
struct NttTwGeom {
    uint32_t log_m, s, log_ns, next_s;
    const uint32_t *lo, *hi;
    uint32_t lo_bits;
    const uint32_t *scale;  // folded into the first boundary of an inverse transform, else null
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

// LDS tile: rows of T elements at a pitch of T + NTT_PAD.  NTT_PAD = 0: the 2^8 x 8 tile (73.7 KiB + 4.6 KiB of stage
// twiddles) then fits twice per CU -- two workgroups, two waves per SIMD -- which padding the rows to 9 elements (83 KiB)
// would halve; the kernel is bound by VALU issue, not by LDS.  Element slot e: limbs 0-3 at plane 0, 4-7 at plane 1
// (uint4 each), limb 8 at plane 2 (u32).  `slots` = rows * pitch.
template <class U>
ZK_D Fu<U> lds_get(const uint4 *lds, uint32_t slots, uint32_t e) {
    static_assert(U::L == 9, "Fr compute form is 9 x 29-bit limbs");
    Fu<U> r;
    uint4 a = lds[e], b = lds[slots + e];
    r.v[0] = a.x, r.v[1] = a.y, r.v[2] = a.z, r.v[3] = a.w;
    r.v[4] = b.x, r.v[5] = b.y, r.v[6] = b.z, r.v[7] = b.w;
    r.v[8] = reinterpret_cast<const uint32_t *>(lds + 2 * slots)[e];
    return r;
}
template <class U>
ZK_D void lds_put(uint4 *lds, uint32_t slots, uint32_t e, const Fu<U> &x) {
    lds[e] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    lds[slots + e] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    reinterpret_cast<uint32_t *>(lds + 2 * slots)[e] = x.v[8];
}
ZK_HD uint32_t ntt_tile_u4(uint32_t slots) { return 2 * slots + (slots + 3) / 4; }  // uint4 units of one tile's three planes

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

// "limb form": the 9 x 29-bit limbs as they are -- limbs 0-7 in 32 bytes at element index i, limb 8 in a 4-byte plane
// `planeb` words behind the array's base.  The factor tables and the intermediate vectors between passes use it: no
// unpack / pack / conditional subtraction at a pass boundary (values there are lazy, < 2p); only the transform's input
// and output are canonical.
template <class U>
ZK_D Fu<U> l_load(const uint32_t *base, size_t planeb, size_t i) {
    const uint4 *q = reinterpret_cast<const uint4 *>(base) + 2 * i;
    const uint4 a = q[0], b = q[1];
    Fu<U> r;
    r.v[0] = a.x, r.v[1] = a.y, r.v[2] = a.z, r.v[3] = a.w;
    r.v[4] = b.x, r.v[5] = b.y, r.v[6] = b.z, r.v[7] = b.w;
    r.v[8] = base[planeb + i];
    return r;
}
template <class U>
ZK_D void l_store(uint32_t *base, size_t planeb, size_t i, const Fu<U> &x) {
    uint4 *q = reinterpret_cast<uint4 *>(base) + 2 * i;
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    base[planeb + i] = x.v[8];
}

// normalised limbs, value < 64 p  ->  [0, p), without a Montgomery product: q = floor(x / p) is estimated from the
// top 13 bits of x with a 24-bit reciprocal (one short too low at most), x - q p < 2p, one conditional subtraction.
template <class U>
ZK_D Fu<U> fu_reduce_small(const Fu<U> &x) {
    constexpr int L = U::L, B = U::B;
    static_assert(L == 9 && B == 29, "sized for the 9 x 29-bit scalar fields");
    // mu = floor(2^272 / p), p < 2^255  =>  mu < 2^24 for p > 2^248; t = x >> 248 < 2^13
    constexpr uint32_t mu = U::MU272;
    const uint32_t t = x.v[8] >> 16;
    const uint32_t q = (uint32_t)(((uint64_t)t * mu) >> 24);
    Fu<U> y;
    uint64_t carry = 0;  // q p, limb by limb
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        carry += (uint64_t)q * U::mod(i);
        const uint32_t qp = (uint32_t)carry & Fu<U>::MASK;
        carry >>= B;
        const uint32_t d = x.v[i] - qp - borrow;  // limbs < 2^30: bit 31 of d is its sign
        borrow = i + 1 < L ? d >> 31 : 0;
        y.v[i] = i + 1 < L ? d & Fu<U>::MASK : d;
    }
    return fu_cond_sub_p(y);
}

// tw[oi] for one pass boundary (output order of the pass being stored = input order of the next): the next pass's
// omega_(Ns' R')^(k' t'), times `scale` when given.  Montgomery form, canonical representative, limb form (m x 36 B).
template <class U>
__global__ __launch_bounds__(256) void ntt_build_tw(NttTwGeom g, uint32_t *__restrict__ out) {
    const uint64_t oi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >> g.log_m) return;
    const uint32_t n_log_ns = g.log_ns + g.s, n_log_stride = g.log_m - g.next_s;
    const uint32_t n_tw_shift = g.log_m - n_log_ns - g.next_s;  // omega_(Ns' R') = omega^(m / (Ns' R'))
    const uint64_t jn = oi & ((1ull << n_log_stride) - 1), tn = oi >> n_log_stride;
    const uint64_t ex = ((jn & ((1ull << n_log_ns) - 1)) * tn) << n_tw_shift;
    Fu<U> f = tw_lookup<U>(g.lo, g.hi, g.lo_bits, ex);
    if (g.scale != nullptr) f = fu_cond_sub_p(fu_mul(f, fu_load<U>(g.scale)));
    l_store<U>(out, (size_t)8 << g.log_m, oi, f);
}
// out[i] = scale * base^i (the coset factors), same format
template <class U>
__global__ __launch_bounds__(256) void ntt_build_powers(const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi, uint32_t lo_bits, uint32_t log_m,
                                                        const uint32_t *__restrict__ scale, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_m) return;
    Fu<U> f = tw_lookup<U>(lo, hi, lo_bits, i);
    if (scale != nullptr) f = fu_cond_sub_p(fu_mul(f, fu_load<U>(scale)));
    l_store<U>(out, (size_t)8 << log_m, i, f);
}
// stage[q] = omega^(q << shift), q < count, Fu form
template <class U>
__global__ __launch_bounds__(256) void ntt_build_stage(const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi, uint32_t lo_bits, uint32_t shift,
                                                       uint32_t count, uint32_t *__restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    fu_store<U>(out + (size_t)q * U::SL, tw_lookup<U>(lo, hi, lo_bits, (uint64_t)q << shift));
}

// One pass over PB polynomials of the batch per workgroup (they share every index, twiddle and table entry).
template <class U, int PB, bool EXT = false>
__global__ __launch_bounds__(256) void ntt_pass(NttPass p) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const uint32_t R = 1u << p.s, T = 1u << p.log_t, nelem = R * T;
    const uint32_t pitch = T + NTT_PAD, slots = R * pitch;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t tile_u4 = ntt_tile_u4(slots);
    uint4 *twr = lds + PB * tile_u4;  // stage twiddles behind the tiles, same three-plane shape
    const uint32_t nhalf = R / 2 > 0 ? R / 2 : 1;

    const uint32_t poly0 = p.poly_base + (blockIdx.x / p.tiles_per_poly) * PB;
    const uint32_t tile = blockIdx.x % p.tiles_per_poly;
    const uint32_t log_stride = p.log_m - p.s;  // m / R
    const uint64_t j0 = (uint64_t)tile << p.log_t;
    const uint32_t ns_mask = (1u << p.log_ns) - 1;  // log_ns < 32 always (m <= 2^32)

    uint32_t ext_src[PB], ext_j[PB];  // EXT: source polynomial and coset (0-based: coset ext_j + 1) of the workgroup's polynomials
    if constexpr (EXT) {
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            ext_src[b] = (poly0 + b) / p.ext_k1;
            ext_j[b] = (poly0 + b) % p.ext_k1;
        }
    }
    for (uint32_t q = tid; q < R / 2; q += nth) lds_put(twr, nhalf, q, fu_load<U>(p.stage + (size_t)q * U::SL));
    // load tile: row bitrev(t), column c  <-  x[j0 + c + t m/R]   [* g^index on the first pass of a coset transform]
    for (uint32_t e = tid; e < nelem; e += nth) {
        const uint32_t t = e >> p.log_t, c = e & (T - 1);
        const uint64_t gi = j0 + c + ((uint64_t)t << log_stride);
        const uint32_t slot = bitrev(t, p.s) * pitch + c;
        Fu<U> g;
        if (p.pre) g = l_load<U>(p.pre, (size_t)8 << p.log_m, gi);
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const size_t ei = (((size_t)(poly0 + b)) << p.log_m) + gi;
            Fu<U> x;
            if constexpr (EXT) {
                if (p.in_lazy) {
                    x = l_load<U>(p.in, p.planeb, ei);
                } else {  // the source's coefficient times this coset's g^index
                    x = g_load<U>(p.in + ((((size_t)ext_src[b]) << p.log_m) + gi) * 8);
                    x = fu_mul(x, l_load<U>(p.ext_pre + (size_t)ext_j[b] * ((size_t)9 << p.log_m), (size_t)8 << p.log_m, gi));
                }
            } else {
                x = p.in_lazy ? l_load<U>(p.in, p.planeb, ei) : g_load<U>(p.in + ei * 8);
                if (p.pre) x = fu_mul(x, g);
            }
            lds_put(lds + b * tile_u4, slots, slot, x);
        }
    }
    __syncthreads();
    // s radix-2 DIT stages over the rows (bit-reversed in, natural out), (a, b) -> (a + w b, a + 4p - w b), two per round trip
    uint32_t st = 0;
    if (p.s & 1) {  // odd stage count: stage 0 alone (w = 1; loaded values are < 2p: keep the subtrahend below 3p)
        for (uint32_t bf = tid; bf < (nelem >> 1); bf += nth) {
            const uint32_t c = bf & (T - 1), q = bf >> p.log_t;
            const uint32_t e0 = (q << 1) * pitch + c, e1 = e0 + pitch;
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                uint4 *tl = lds + b * tile_u4;
                Fu<U> x0 = lds_get<U>(tl, slots, e0), x1 = fu_cond_sub_p(lds_get<U>(tl, slots, e1));
                lds_put(tl, slots, e0, fu_add(x0, x1));
                lds_put(tl, slots, e1, fu_sub<4>(x0, x1));
            }
        }
        st = 1;
        __syncthreads();
    }
    for (; st < p.s; st += 2) {
        const uint32_t h = 1u << st;
        for (uint32_t gq = tid; gq < (nelem >> 2); gq += nth) {
            const uint32_t c = gq & (T - 1), q = gq >> p.log_t;
            const uint32_t qq = q & (h - 1);
            const uint32_t i0 = ((q - qq) << 2) + qq;  // rows i0, i0 + h, i0 + 2h, i0 + 3h
            const uint32_t e0 = i0 * pitch + c, eh = h * pitch;
            // stage st pairs (i0, i0 + h) and (i0 + 2h, i0 + 3h) with omega_R^(qq << (s-1-st)); stage st + 1 pairs
            // (i0, i0 + 2h) with omega_R^(qq << (s-2-st)) and (i0 + h, i0 + 3h) with omega_R^((qq + h) << (s-2-st))
            Fu<U> w1, w2a, w2b;
            if (st != 0) w1 = lds_get<U>(twr, nhalf, qq << (p.s - 1 - st));
            if (st != 0) w2a = lds_get<U>(twr, nhalf, qq << (p.s - 2 - st));
            w2b = lds_get<U>(twr, nhalf, (qq + h) << (p.s - 2 - st));
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                uint4 *tl = lds + b * tile_u4;
                Fu<U> x0 = lds_get<U>(tl, slots, e0), x1 = lds_get<U>(tl, slots, e0 + eh), x2 = lds_get<U>(tl, slots, e0 + 2 * eh),
                      x3 = lds_get<U>(tl, slots, e0 + 3 * eh);
                if (st != 0) {
                    x1 = fu_mul(x1, w1);
                    x3 = fu_mul(x3, w1);
                } else {  // w = 1; values straight from the load are < 2p
                    x1 = fu_cond_sub_p(x1);
                    x3 = fu_cond_sub_p(x3);
                }
                const Fu<U> a0 = fu_add(x0, x1), a1 = fu_sub<4>(x0, x1), a2 = fu_add(x2, x3), a3 = fu_sub<4>(x2, x3);
                // in the first round (h = 1, qq = 0) the even twiddle of the second stage is omega^0 = 1: a2 (< 3p) goes through as it is
                const Fu<U> b2 = st != 0 ? fu_mul(a2, w2a) : a2, b3 = fu_mul(a3, w2b);
                lds_put(tl, slots, e0, fu_add(a0, b2));
                lds_put(tl, slots, e0 + 2 * eh, fu_sub<4>(a0, b2));
                lds_put(tl, slots, e0 + eh, fu_add(a1, b3));
                lds_put(tl, slots, e0 + 3 * eh, fu_sub<4>(a1, b3));
            }
        }
        __syncthreads();
    }
    // store: y[(j - k) R + k + t' Ns] = v[t'] * (this boundary's table entry | coset / scale factor | nothing)
    const uint32_t last = p.next_s == 0;
    Fu<U> scale;
    if (p.scale) scale = fu_load<U>(p.scale);
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
        const uint64_t j = j0 + c;
        const uint64_t k = j & ns_mask;
        const uint64_t oi = ((j - k) << p.s) + k + ((uint64_t)tp << p.log_ns);
        const uint32_t *ftab = last ? p.post : p.tw;
        Fu<U> f;
        if (ftab) f = l_load<U>(ftab, (size_t)8 << p.log_m, oi);
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            Fu<U> x = lds_get<U>(lds + b * tile_u4, slots, tp * pitch + c);
            const size_t eo = (((size_t)(poly0 + b)) << p.log_m) + oi;
            if (p.out_lazy) {  // an intermediate vector: the product (< 2p, limbs normalised) is stored as it is
                l_store<U>(p.out, p.planeb, eo, fu_mul(x, f));
                continue;
            }
            if (ftab) x = fu_cond_sub_p(fu_mul(x, f));
            else if (p.scale) x = fu_cond_sub_p(fu_mul(x, scale));
            else x = fu_reduce_small(x);
            if constexpr (EXT) {
                g_store<U>(p.out + ((((size_t)ext_src[b]) << (p.log_m + p.ext_log_k)) + ((size_t)oi << p.ext_log_k) + ext_j[b] + 1) * 8, x);
            } else {
                g_store<U>(p.out + eo * 8, x);
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
static void ntt_free_one(NttTables *t) {
    (void)hipFree(t->d_lo);
    (void)hipFree(t->d_hi);
    (void)hipFree(t->d_clo);
    (void)hipFree(t->d_chi);
    (void)hipFree(t->d_scale);
    (void)hipFree(t->d_base);
    (void)hipFree(t->d_prepost);
    for (int i = 0; i < NTT_MAX_PASSES; ++i) {
        (void)hipFree(t->d_tw[i]);
        (void)hipFree(t->d_stage[i]);
    }
    delete t;
}

void zk_ntt_free_ext_tables(zkhip_ctx *ctx);
void zk_ntt_free_tables(zkhip_ctx *ctx) {
    for (NttTables *t : ctx->ntt_tables) ntt_free_one(t);
    ctx->ntt_tables.clear();
    zk_ntt_free_ext_tables(ctx);
}

// the radix plan: log_m split into np nearly equal radices (larger first), tile widths
struct NttPlan {
    int np;
    int sv[NTT_MAX_PASSES], log_t[NTT_MAX_PASSES];
};
static NttPlan ntt_plan(size_t log_m, int smax, int tile_log) {
    NttPlan pl;
    pl.np = (int)((log_m + smax - 1) / smax);
    for (int i = 0; i < pl.np; ++i) {
        pl.sv[i] = (int)(log_m / pl.np) + (i < (int)(log_m % pl.np) ? 1 : 0);
        const int log_cols = (int)log_m - pl.sv[i];  // log2(m / R)
        pl.log_t[i] = std::min(std::max(0, tile_log), log_cols);
        while (pl.sv[i] + pl.log_t[i] > 12 && pl.log_t[i] > 0) --pl.log_t[i];  // LDS budget: R * T * 36 B <= 144 KiB
    }
    return pl;
}

template <class U>
static int ntt_get_tables(zkhip_ctx *ctx, int curve, size_t log_m, const uint64_t *omega, int inverse, const uint64_t *coset, int smax, int tile_log,
                          NttTables **out) {
    for (NttTables *t : ctx->ntt_tables) {
        if (t->curve == curve && t->log_m == log_m && t->inverse == inverse && t->has_coset == (coset != nullptr) && t->smax == smax &&
            t->tile_log == tile_log && memcmp(t->omega, omega, 32) == 0 && (coset == nullptr || memcmp(t->coset, coset, 32) == 0)) {
            *out = t;
            return 0;
        }
    }
    if ((int)((log_m + smax - 1) / smax) > NTT_MAX_PASSES) return ZKHIP_ERR_RANGE;
    const NttPlan pl = ntt_plan(log_m, smax, tile_log);
    // the per-index tables are m x 32 B each: keep the cache within a few entries per size (a prover alternates between a
    // handful of (direction, coset) variants of one or two sizes)
    if (ctx->ntt_tables.size() >= 24) {
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ntt_free_one(ctx->ntt_tables.front());
        ctx->ntt_tables.erase(ctx->ntt_tables.begin());
    }
    NttTables *t = new NttTables();
    t->curve = curve;
    t->log_m = log_m;
    t->inverse = inverse;
    t->has_coset = coset != nullptr;
    t->smax = smax;
    t->tile_log = tile_log;
    memcpy(t->omega, omega, 32);
    if (coset) memcpy(t->coset, coset, 32);
    t->lo_bits = (int)((log_m + 1) / 2);
    if (t->lo_bits < 1) t->lo_bits = 1;
    const uint32_t nlo = 1u << t->lo_bits;
    const uint32_t nhi = (uint32_t)(((size_t)1 << log_m) >> t->lo_bits) + 1;
    const size_t eb = U::SL * 4;  // bytes per power-table entry
    const size_t m = (size_t)1 << log_m;
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
    const unsigned gm = (unsigned)((m + 255) / 256);
    if (coset) {
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_clo, (size_t)nlo * eb));
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_chi, (size_t)nhi * eb));
        ZK_LAUNCH(ctx, "ntt_pow_table", ntt_pow_table<U>, dim3((nlo + 255) / 256), dim3(256), 0, t->d_base + U::SL, nlo, 0u, t->d_clo);
        ZK_LAUNCH(ctx, "ntt_pow_table", ntt_pow_table<U>, dim3((nhi + 255) / 256), dim3(256), 0, t->d_base + U::SL, nhi, (uint32_t)t->lo_bits,
                  t->d_chi);
        // g^i while loading (forward) / (1/m) g^-i while storing (inverse): one entry per index
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_prepost, m * 36));
        ZK_LAUNCH(ctx, "ntt_build_tw", ntt_build_powers<U>, dim3(gm), dim3(256), 0, t->d_clo, t->d_chi, (uint32_t)t->lo_bits, (uint32_t)log_m,
                  inverse ? t->d_scale : (const uint32_t *)nullptr, t->d_prepost);
    }
    // per pass: stage twiddles omega_R^q = omega^(q m / R); per boundary: the store factors in output order.  An inverse
    // transform without coset folds 1/m into its first boundary (with a coset it rides on the post table; a single
    // pass multiplies by it directly).
    uint32_t log_ns = 0;
    for (int i = 0; i < pl.np; ++i) {
        const uint32_t s = (uint32_t)pl.sv[i], half = std::max<uint32_t>(1, (1u << s) / 2);
        ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_stage[i], (size_t)half * eb));
        ZK_LAUNCH(ctx, "ntt_build_tw", ntt_build_stage<U>, dim3((half + 255) / 256), dim3(256), 0, t->d_lo, t->d_hi, (uint32_t)t->lo_bits,
                  (uint32_t)log_m - s, half, t->d_stage[i]);
        if (i + 1 < pl.np) {
            NttTwGeom g;
            g.log_m = (uint32_t)log_m;
            g.s = s;
            g.log_ns = log_ns;
            g.next_s = (uint32_t)pl.sv[i + 1];
            g.lo = t->d_lo;
            g.hi = t->d_hi;
            g.lo_bits = (uint32_t)t->lo_bits;
            g.scale = (inverse && !coset && i == 0) ? t->d_scale : nullptr;
            ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_tw[i], m * 36));
            ZK_LAUNCH(ctx, "ntt_build_tw", ntt_build_tw<U>, dim3(gm), dim3(256), 0, g, t->d_tw[i]);
        }
        log_ns += s;
    }
    // omega must be a PRIMITIVE m-th root of unity (the caller's evaluation domain is the basic radix-2 one): omega^m = 1
    // and omega^(m/2) != 1, read back from the tables just built.  Anything else would transform over the wrong domain.
    const size_t half = m >> 1;
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
        ntt_free_one(t);
        ctx->last_error = "omega is not a primitive 2^" + std::to_string(log_m) + "-th root of unity";
        return ZKHIP_ERR_INVALID;
    }
    *out = t;
    return 0;
}

// the extension mode of a forward transform (zk_ntt_extend): `batch` = the launch's polynomials = sources x k1 cosets
struct NttExt {
    uint32_t k1, log_k;
    const uint32_t *pre;
    uint32_t *out;
};

template <class U>
static int ntt_run_t(zkhip_ctx *ctx, int curve, uint32_t *d_data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
                     const uint64_t *coset, const NttExt *ext = nullptr) {
    if (batch == 0 || (log_m == 0 && !ext)) return 0;  // a 1-point transform is the identity (also with coset: g^0 = 1, 1/1 = 1)
    const int smax = std::max(1, std::min(10, ctx->opt_ntt_radix_log));
    NttTables *tb = nullptr;
    ZK_TRY(ntt_get_tables<U>(ctx, curve, log_m, omega, inverse, coset, smax, ctx->opt_ntt_tile_log, &tb));
    const NttPlan pl = ntt_plan(log_m, smax, ctx->opt_ntt_tile_log);
    const int np = pl.np;
    const size_t m = (size_t)1 << log_m;
    // intermediates between passes live in the workspace in limb form (36 B per element: no unpack / pack at a boundary),
    // two buffers alternating; the first pass reads and the last pass writes the caller's canonical vector
    const size_t lbytes = batch * m * 36;
    uint32_t *ws2[2] = {nullptr, nullptr};
    if (np > 1) {
        size_t need = zkhip_ctx::ws_round(lbytes) * (np > 2 ? 2 : 1);
        ZK_TRY(ctx->ws_reserve(need));
        ctx->ws_reset();
        ws2[0] = ctx->ws_take<uint32_t>(lbytes / 4);
        if (np > 2) ws2[1] = ctx->ws_take<uint32_t>(lbytes / 4);
    }
    uint32_t log_ns = 0;
    const uint32_t *src = d_data;
    for (int i = 0; i < np; ++i) {
        uint32_t *dst = i == np - 1 ? (ext ? ext->out : d_data) : ws2[i & 1];  // np == 1: in place (the single tile is read fully into LDS before any store)
        NttPass p;
        p.in = src;
        p.out = dst;
        p.log_m = (uint32_t)log_m;
        p.s = (uint32_t)pl.sv[i];
        p.next_s = i + 1 < np ? (uint32_t)pl.sv[i + 1] : 0u;
        p.log_ns = log_ns;
        p.log_t = (uint32_t)pl.log_t[i];
        p.tiles_per_poly = 1u << ((uint32_t)log_m - p.s - p.log_t);
        p.stage = tb->d_stage[i];
        p.tw = tb->d_tw[i];  // null on the last pass
        p.pre = (!inverse && coset != nullptr && i == 0) ? tb->d_prepost : nullptr;
        p.post = (inverse && coset != nullptr && i == np - 1) ? tb->d_prepost : nullptr;
        p.scale = (inverse && coset == nullptr && np == 1) ? tb->d_scale : nullptr;
        p.in_lazy = i > 0 ? 1u : 0u;
        p.out_lazy = i < np - 1 ? 1u : 0u;
        p.planeb = batch * m * 8;
        p.ext_k1 = ext ? ext->k1 : 0u;
        p.ext_log_k = ext ? ext->log_k : 0u;
        p.ext_pre = ext ? ext->pre : nullptr;
        const size_t nhalf = std::max<size_t>(1, ((size_t)1 << p.s) / 2);
        // PB = 1, 2 or 4 polynomials of the batch per workgroup (they share indices, twiddle registers and the factor-table
        // reads) while the tiles leave room for a second workgroup on the CU; option "ntt_pair" = log2(PB) wanted.  A batch that PB
        // does not divide runs its remainder as one more launch of single polynomials (round 5: an odd batch -- a lone polynomial's
        // 3 or 7 new cosets -- used to run unpaired altogether).
        // The cosets of an extension run ONE per workgroup: as pairs they share no table entry (each coset has its own g^i) and measured 5 % slower
        // (permutation argument 13.9 against 14.6-14.9 ms, profiles/r05_ab_legs_pair_coset.txt).
        const int want = ext ? 1 : 1 << std::max(0, std::min(2, ctx->opt_ntt_pair));
        auto launch = [&](NttPass q, int pb, size_t first, size_t count) -> int {
            if (count == 0) return 0;
            q.poly_base = (uint32_t)first;
            // a narrower tile that lets the polynomials share a workgroup beats a wider one that does not (measured)
            while (pb > 1 && q.log_t > 2 && ((size_t)pb * ntt_tile_u4((1u << q.s) * ((1u << q.log_t) + NTT_PAD)) + ntt_tile_u4((uint32_t)nhalf)) * 16 > 80 * 1024) {
                --q.log_t;
                q.tiles_per_poly <<= 1;
            }
            const size_t slots_p = ((size_t)1 << q.s) * (((size_t)1 << q.log_t) + NTT_PAD);
            while (pb > 1 && (((size_t)pb * ntt_tile_u4((uint32_t)slots_p) + ntt_tile_u4((uint32_t)nhalf)) * 16 > 80 * 1024 || count % pb != 0)) pb >>= 1;
            const size_t nelem_p = (size_t)1 << (q.s + q.log_t);
            const size_t lds = ((size_t)pb * ntt_tile_u4((uint32_t)slots_p) + ntt_tile_u4((uint32_t)nhalf)) * 16;
            const size_t grid = count / pb * q.tiles_per_poly;
            if (grid >= (1ull << 31)) return ZKHIP_ERR_RANGE;
            const unsigned threads = (unsigned)std::min<size_t>(256, std::max<size_t>(64, nelem_p / 4));
            if (ext) {  // pairs or singles (four per workgroup is the slow shape: DESIGN.md section 5)
                if (pb >= 2) {
                    ZK_MAX_LDS(ctx, (ntt_pass<U, 2, true>), 160 * 1024);
                    ZK_LAUNCH(ctx, "ntt_pass_ext", (ntt_pass<U, 2, true>), dim3((unsigned)grid), dim3(threads), lds, q);
                } else {
                    ZK_MAX_LDS(ctx, (ntt_pass<U, 1, true>), 160 * 1024);
                    ZK_LAUNCH(ctx, "ntt_pass_ext", (ntt_pass<U, 1, true>), dim3((unsigned)grid), dim3(threads), lds, q);
                }
            } else if (pb == 4) {
                ZK_MAX_LDS(ctx, (ntt_pass<U, 4>), 160 * 1024);
                ZK_LAUNCH(ctx, "ntt_pass", (ntt_pass<U, 4>), dim3((unsigned)grid), dim3(threads), lds, q);
            } else if (pb == 2) {
                ZK_MAX_LDS(ctx, (ntt_pass<U, 2>), 160 * 1024);
                ZK_LAUNCH(ctx, "ntt_pass", (ntt_pass<U, 2>), dim3((unsigned)grid), dim3(threads), lds, q);
            } else {
                ZK_MAX_LDS(ctx, (ntt_pass<U, 1>), 160 * 1024);
                ZK_LAUNCH(ctx, "ntt_pass", (ntt_pass<U, 1>), dim3((unsigned)grid), dim3(threads), lds, q);
            }
            return 0;
        };
        const size_t whole = want > 1 ? batch / want * want : batch;
        ZK_TRY(launch(p, want, 0, whole));
        ZK_TRY(launch(p, 1, whole, batch - whole));
        src = dst;
        log_ns += p.s;
    }
    return 0;
}

// g_j^i = omega_big^(j i) for the cosets j = 1 .. k1 of the n-point domain inside the K n-point one, i < m: k1 tables in limb form.
// One lane per entry, square-and-multiply over the bits of j i (< 2^36): built once per (size, K, root), ~35 products per entry.
template <class U>
__global__ __launch_bounds__(256) void ntt_build_ext_pre(const uint32_t *__restrict__ omega_big_c, uint32_t log_m, uint32_t k1, uint32_t *__restrict__ out) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((e >> log_m) >= k1) return;
    const uint64_t j = (e >> log_m) + 1, i = e & (((uint64_t)1 << log_m) - 1);
    Fu<U> b = fu_cond_sub_p(fu_from_canonical<U>(omega_big_c)), r = Fu<U>::one();
    for (uint64_t x = j * i; x; x >>= 1) {
        if (x & 1) r = fu_mul(r, b);
        b = fu_mul(b, b);
    }
    l_store<U>(out + (size_t)(j - 1) * ((size_t)9 << log_m), (size_t)8 << log_m, i, fu_cond_sub_p(r));
}

struct NttExtTables {
    int curve;
    size_t log_m, log_k;
    uint64_t omega_big[4];
    uint32_t *d_pre = nullptr;
};
static std::vector<std::pair<zkhip_ctx *, NttExtTables *>> g_ext_tables;  // owned per context; freed with the context's NTT tables
static std::mutex g_ext_mutex;

void zk_ntt_free_ext_tables(zkhip_ctx *ctx) {
    std::lock_guard<std::mutex> lock(g_ext_mutex);
    for (size_t i = g_ext_tables.size(); i-- > 0;)
        if (g_ext_tables[i].first == ctx) {
            (void)hipFree(g_ext_tables[i].second->d_pre);
            delete g_ext_tables[i].second;
            g_ext_tables.erase(g_ext_tables.begin() + i);
        }
}

template <class U>
static int ntt_extend_t(zkhip_ctx *ctx, int curve, uint32_t *d_coeffs, size_t log_m, size_t batch, const uint64_t *omega, uint32_t *d_out, size_t log_k,
                        const uint64_t *omega_big) {
    const size_t m = (size_t)1 << log_m, k1 = ((size_t)1 << log_k) - 1;
    NttExtTables *t = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_ext_mutex);
        size_t mine = 0;
        for (auto &e : g_ext_tables) {
            if (e.first != ctx) continue;
            ++mine;
            if (e.second->curve == curve && e.second->log_m == log_m && e.second->log_k == log_k && memcmp(e.second->omega_big, omega_big, 32) == 0) t = e.second;
        }
        if (!t && mine >= 16) {  // keep the cache bounded: drop this context's oldest entry
            for (size_t i = 0; i < g_ext_tables.size(); ++i)
                if (g_ext_tables[i].first == ctx) {
                    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                    (void)hipFree(g_ext_tables[i].second->d_pre);
                    delete g_ext_tables[i].second;
                    g_ext_tables.erase(g_ext_tables.begin() + i);
                    break;
                }
        }
    }
    if (!t) {
        t = new NttExtTables();
        t->curve = curve;
        t->log_m = log_m;
        t->log_k = log_k;
        memcpy(t->omega_big, omega_big, 32);
        uint32_t *d_w = nullptr;
        // a HIP call that fails half way must not leak the table object or its device memory
        auto build = [&]() -> int {
            ZK_HIP_CHECK(ctx, hipMalloc((void **)&t->d_pre, k1 * m * 36));
            ZK_HIP_CHECK(ctx, hipMalloc((void **)&d_w, 32));
            ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_w, omega_big, 32, hipMemcpyHostToDevice, ctx->stream));
            const size_t entries = k1 * m;
            ZK_LAUNCH(ctx, "ntt_build_tw", ntt_build_ext_pre<U>, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, d_w, (uint32_t)log_m, (uint32_t)k1, t->d_pre);
            ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            return 0;
        };
        const int rc = build();
        if (d_w) (void)hipFree(d_w);
        if (rc != 0) {
            if (t->d_pre) (void)hipFree(t->d_pre);
            delete t;
            return rc;
        }
        std::lock_guard<std::mutex> lock(g_ext_mutex);
        g_ext_tables.emplace_back(ctx, t);
    }
    NttExt ext{(uint32_t)k1, (uint32_t)log_k, t->d_pre, d_out};
    return ntt_run_t<U>(ctx, curve, d_coeffs, log_m, batch * k1, omega, 0, nullptr, &ext);
}

// evaluations on the cosets 1 .. K - 1 of the n-point domain inside the K n-point one, written into their places of the K n-point vector
// (d_out[b][i K + j], j >= 1; the caller puts the n known values at j = 0): d_coeffs holds `batch` coefficient vectors of n = 2^log_m
int zk_ntt_extend(zkhip_ctx *ctx, int curve, uint32_t *d_coeffs, size_t log_m, size_t batch, const uint64_t *omega, uint32_t *d_out, size_t log_k,
                  const uint64_t *omega_big) {
    if (log_m + log_k > 32 || log_k == 0 || log_k > 4) return ZKHIP_ERR_RANGE;
    if (batch * (((size_t)1 << log_k) - 1) >= ((size_t)1 << 20)) return ZKHIP_ERR_RANGE;
    if (curve == CURVE_BLS12_381) return ntt_extend_t<BlsFrU>(ctx, curve, d_coeffs, log_m, batch, omega, d_out, log_k, omega_big);
    if (curve == CURVE_BN254) return ntt_extend_t<BnFrU>(ctx, curve, d_coeffs, log_m, batch, omega, d_out, log_k, omega_big);
    return ZKHIP_ERR_INVALID;
}

int zk_ntt_run(zkhip_ctx *ctx, int curve, uint32_t *d_data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
               const uint64_t *coset) {
    if (log_m > 32) return ZKHIP_ERR_RANGE;
    if (curve == CURVE_BLS12_381) return ntt_run_t<BlsFrU>(ctx, curve, d_data, log_m, batch, omega, inverse, coset);
    if (curve == CURVE_BN254) return ntt_run_t<BnFrU>(ctx, curve, d_data, log_m, batch, omega, inverse, coset);
    return ZKHIP_ERR_INVALID;
}
