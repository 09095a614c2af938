// Polynomial helpers of the LPC / FRI commitment layer that sit directly on the NTT ("next" rows N2 of SURVEY 8f).
//
//   zkhip_poly_resize_dev   polynomial_dfs::resize(new_size) as precommit<FRI> uses it
//                           (zk/commitments/detail/polynomial/basic_fri.hpp:452-455): evaluations on the n-point
//                           domain -> coefficients (inverse NTT) -> evaluations on the larger 2^log_out-point domain.
//   zkhip_fri_fold_dev      detail::fold_polynomial, DFS form (zk/commitments/detail/polynomial/fold_polynomial.hpp:68-93):
//                           f'(i) = 1/2 [ (1 + alpha w^-i) f(i) + (1 - alpha w^-i) f(i + size/2) ],  i < size/2.
// The Merkle / hashing side of LPC is out of scope.
#include <algorithm>

#include "ctx.hpp"
#include "fu.hpp"

using namespace zkhip;

#define ZK_FR_DISPATCH(curve, ...)                 \
    if ((curve) == CURVE_BLS12_381) {           \
        typedef BlsFrU U;                       \
        __VA_ARGS__;                            \
    } else {                                    \
        typedef BnFrU U;                        \
        __VA_ARGS__;                            \
    }

// out[b][i] = in[b][i] for i < n, 0 for n <= i < m   (32-byte elements)
__global__ __launch_bounds__(256) void poly_pad_copy(const uint4 *__restrict__ in, uint32_t log_n, uint32_t log_m, size_t total,
                                                     uint4 *__restrict__ out) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into out, in elements
    if (e >= total) return;
    size_t b = e >> log_m, i = e & (((size_t)1 << log_m) - 1);
    uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
    if (i < ((size_t)1 << log_n)) {
        size_t s = (b << log_n) + i;
        lo = in[2 * s];
        hi = in[2 * s + 1];
    }
    out[2 * e] = lo;
    out[2 * e + 1] = hi;
}


// out[b][i << log_k] = in[b][i]: the n known values of an n -> K n extension at their places (coset 0 of the larger domain)
__global__ __launch_bounds__(256) void poly_spread(const uint4 *__restrict__ in, uint32_t log_k, size_t total, uint32_t log_n, uint4 *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into in, in elements
    if (e >= total) return;
    const size_t b = e >> log_n, i = e & (((size_t)1 << log_n) - 1);
    const size_t o = (b << (log_n + log_k)) + (i << log_k);
    out[2 * o] = in[2 * e];
    out[2 * o + 1] = in[2 * e + 1];
}

// precommit<FRI>'s leaf layout (basic_fri.hpp:456-492, m = 2): leaf x (< D / 2^step) holds, for every polynomial in
// turn, the pairs (f[s_i], f[s_i + D/2 mod D]) for i < 2^step / 2, with s_0 = x and
// s_(2^l + j) = s_j + D / (4 * 2^l) mod D.  out[((x * batch + p) * half + i) * 2 + {0, 1}], 32-byte elements.
__global__ __launch_bounds__(256) void fri_leaf_gather(const uint4 *__restrict__ polys, uint32_t log_d, uint32_t batch, uint32_t step, size_t total,
                                                       uint4 *__restrict__ out) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index: (x * batch + p) * half + i
    if (e >= total) return;
    const uint32_t half_log = step - 1;  // log2(coset_size / 2)
    const size_t i = e & (((size_t)1 << half_log) - 1), xp = e >> half_log;
    const size_t p = xp % batch, x = xp / batch, D = (size_t)1 << log_d;
    size_t s = x;
    for (uint32_t l = 0; l < half_log; ++l)
        if ((i >> l) & 1) s += D >> (2 + l);
    s &= D - 1;
    const size_t s2 = (s + (D >> 1)) & (D - 1);
    const uint4 *f = polys + 2 * (p << log_d);
    out[4 * e] = f[2 * s];
    out[4 * e + 1] = f[2 * s + 1];
    out[4 * e + 2] = f[2 * s2];
    out[4 * e + 3] = f[2 * s2 + 1];
}

static constexpr uint32_t FOLD_CHUNK = 128;  // consecutive i per lane: one power, then a running product

// consts = [alpha, w^-1, 1/2] in Montgomery form, canonical representatives.  One lane: w is a 2^log_size-th root of unity, so w^-1 = w^(2^log_size - 1) =
// prod_k w^(2^k) -- 2 log_size products instead of a field inversion (380 products on a single lane: 0.33 ms, which made every round of an FRI commit
// phase cost 0.7 ms whatever its size); 1/2 = (p + 1) / 2 needs no inversion either.  A w of another order takes the inversion.
template <class U>
__global__ void fri_fold_setup(const uint32_t *__restrict__ alpha_c, const uint32_t *__restrict__ omega_c, uint32_t log_size, uint32_t *__restrict__ consts) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    fu_store<U>(consts, fu_cond_sub_p(fu_from_canonical<U>(alpha_c)));
    const Fu<U> w = fu_cond_sub_p(fu_from_canonical<U>(omega_c));
    Fu<U> winv = Fu<U>::one(), sq = w;
    for (uint32_t k = 0; k < log_size; ++k) {
        winv = fu_cond_sub_p(fu_mul(winv, sq));
        sq = fu_cond_sub_p(fu_mul(sq, sq));
    }
    const Fu<U> one = fu_cond_sub_p(Fu<U>::one());
    const Fu<U> check = fu_cond_sub_p(fu_mul(winv, w));
    bool ok = true;
#pragma unroll
    for (int l = 0; l < U::L; ++l) ok = ok && check.v[l] == one.v[l];
    if (!ok) winv = fu_cond_sub_p(fu_inv(w));
    fu_store<U>(consts + U::SL, winv);
    // (p + 1) / 2, canonical -> Montgomery: p is odd, so p + 1 halves exactly
    uint32_t half[U::NL];
    uint32_t carry = 1;
    for (int l = 0; l < U::NL; ++l) {
        const uint64_t t = (uint64_t)U::sat::mod(l) + carry;
        half[l] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    for (int l = 0; l < U::NL; ++l) half[l] = (half[l] >> 1) | ((l + 1 < U::NL ? half[l + 1] : carry) << 31);
    fu_store<U>(consts + 2 * U::SL, fu_cond_sub_p(fu_from_canonical<U>(half)));
}

template <class U>
__global__ __launch_bounds__(256) void fri_fold(const uint32_t *__restrict__ f, uint32_t log_size, const uint32_t *__restrict__ consts,
                                                uint32_t *__restrict__ out) {
    const uint32_t half = 1u << (log_size - 1);
    const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * FOLD_CHUNK;
    if (i0 >= half) return;
    const Fu<U> winv = fu_load<U>(consts + U::SL), half_inv = fu_load<U>(consts + 2 * U::SL);
    const Fu<U> one = fu_cond_sub_p(Fu<U>::one());
    Fu<U> acc = fu_load<U>(consts), pw = winv;  // acc = alpha w^-i0
    for (uint32_t e = i0; e; e >>= 1) {
        if (e & 1) acc = fu_mul(acc, pw);
        pw = fu_mul(pw, pw);
    }
    const uint32_t hi = min(half, i0 + FOLD_CHUNK);
    for (uint32_t i = i0; i < hi; ++i) {
        acc = fu_cond_sub_p(acc);                                  // alpha w^-i, Montgomery, canonical
        Fu<U> a = fu_unpack<U>(f + (size_t)i * U::NL), b = fu_unpack<U>(f + (size_t)(half + i) * U::NL);  // canonical integers
        Fu<U> x = fu_mul(a, fu_add(one, acc));                     // f(i) (1 + acc): canonical domain, < 2p
        Fu<U> y = fu_mul(b, fu_sub<4>(one, acc));                  // f(i + half) (1 - acc)
        Fu<U> r = fu_mul(fu_add(x, y), half_inv);                  // / 2
        fu_pack<U>(out + (size_t)i * U::NL, fu_cond_sub_p(r));
        acc = fu_mul(acc, winv);
    }
}


// ---- coefficient-form arithmetic of the KZG opening proof (kzg_v2.hpp:236-305) --------------------------------
// Elements are canonical integers in HBM; constants (evaluation points, linear-combination coefficients) are
// converted to Montgomery form once, so `fu_mul(canonical, montgomery)` yields the canonical-domain product < 2p.

// out[i] = a[i] op b[i]   (op 0: +, 1: -, 2: *)   polynomial_dfs::operator+=, -=, *= on equal domains
template <class U>
__global__ __launch_bounds__(256) void fr_vec_op(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t count, int op,
                                                 uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fu<U> x = fu_unpack<U>(a + i * U::NL), y = fu_unpack<U>(b + i * U::NL), r;
    if (op == 0) r = fu_cond_sub_p(fu_add(x, y));
    else if (op == 1) r = fu_cond_sub_p(fu_cond_sub_p(fu_sub<2>(x, y)));  // x + 2p - y in (p, 3p)
    else r = fu_cond_sub_p(fu_mul(fu_mul(x, y), Fu<U>::r2()));  // (x y / R) R^2 / R
    fu_pack<U>(out + i * U::NL, r);
}

// out[i] = a x[i] + b y[i] + c (y nullable): the linear factors placeholder's arguments multiply up, e.g. (1 + beta)(gamma + input) and
// (1 + beta) gamma + value + beta value(omega X) (lookup_argument.hpp:313, 329, 361); a, b, c canonical, by value
struct FrWords {
    uint32_t w[8];
};
template <class U>
__global__ __launch_bounds__(256) void fr_vec_affine(const uint32_t *__restrict__ x, const uint32_t *__restrict__ y, FrWords a, FrWords b, FrWords c, size_t count,
                                                     uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Fu<U> am = fu_mul(fu_unpack<U>(a.w), Fu<U>::r2());  // a R: (a R) x / R = a x
    Fu<U> r = fu_cond_sub_p(fu_add(fu_cond_sub_p(fu_mul(am, fu_unpack<U>(x + i * U::NL))), fu_unpack<U>(c.w)));
    if (y) {
        const Fu<U> bm = fu_mul(fu_unpack<U>(b.w), Fu<U>::r2());
        r = fu_cond_sub_p(fu_add(r, fu_cond_sub_p(fu_mul(bm, fu_unpack<U>(y + i * U::NL)))));
    }
    fu_pack<U>(out + i * U::NL, r);
}

// out[i] = prod_k in_k[i]: math::polynomial_product on polynomials already brought to the product's domain
// (placeholder/permutation_argument.hpp:148, gates_argument.hpp:117); `ptrs` is a device array of `count` pointers.
template <class U>
__global__ __launch_bounds__(256) void fr_vec_prod(const uint32_t *const *__restrict__ ptrs, uint32_t count, size_t n, uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // canonical operands: mul(x, y) = x y / R.  The accumulator stays in Montgomery form (acc R): every factor is lifted
    // (x_k R^2 / R = x_k R) and multiplied in ((acc R)(x_k R) / R = acc x_k R); one last product by the plain 1 drops the R.
    Fu<U> acc = fu_mul(fu_unpack<U>(ptrs[0] + i * U::NL), Fu<U>::r2());
    for (uint32_t k = 1; k < count; ++k) acc = fu_mul(acc, fu_mul(fu_unpack<U>(ptrs[k] + i * U::NL), Fu<U>::r2()));
    Fu<U> one = Fu<U>::zero();
    one.v[0] = 1;
    fu_pack<U>(out + i * U::NL, fu_cond_sub_p(fu_mul(acc, one)));
}

// f / (X^n - 1) in coefficient form: the quotient of placeholder's `F_consolidated_normal / common_data.Z` (prover.hpp:273-275; Z is the
// vanishing polynomial X^n - 1 of the basic n-row domain).  With f = sum_k X^(k n) f_k (blocks of n coefficients) the long division
// gives q_(k-1) = f_k + q_k, i.e. quot[i + (k - 1) n] = sum_{j >= k} f[i + j n]: one lane per residue i < n walks its column top down.
// The remainder f[i] + quot[i] must vanish for an exact division; lanes that see a non-zero one count into *bad.
template <class U>
__global__ __launch_bounds__(256) void poly_div_vanishing(const uint32_t *__restrict__ f, size_t len, size_t n, uint32_t *__restrict__ quot,
                                                          uint32_t *__restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i >= len) return;
    const size_t top = (len - 1 - i) / n;  // f[i + top n] is the column's last coefficient
    Fu<U> s = Fu<U>::zero();
    for (size_t k = top; k >= 1; --k) {
        s = fu_cond_sub_p(fu_add(s, fu_unpack<U>(f + (i + k * n) * U::NL)));
        fu_pack<U>(quot + (i + (k - 1) * n) * U::NL, s);
    }
    const Fu<U> r = fu_cond_sub_p(fu_add(s, fu_unpack<U>(f + i * U::NL)));
    if (!r.limbs_zero()) atomicAdd(bad, 1u);
}

// out[i] = in[(i + rot) mod 2^log_n] on 32-byte elements: math::polynomial_shift (f(X) -> f(omega^shift X) on the
// evaluation vector: index i reads i + shift * (size / domain_size)), permutation_argument.hpp:148, lookup_argument.hpp:232
__global__ __launch_bounds__(256) void poly_rotate(const uint4 *__restrict__ in, uint32_t log_n, size_t rot, uint4 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    const size_t s = (i + rot) & (((size_t)1 << log_n) - 1);
    out[2 * i] = in[2 * s];
    out[2 * i + 1] = in[2 * s + 1];
}

// out[b][i] = in[b][i << shift]: polynomial_dfs::resize to a SMALLER domain -- the evaluations of a polynomial of degree
// < 2^log_out on the 2^log_out-point domain are every 2^shift-th of its evaluations on the larger one
__global__ __launch_bounds__(256) void poly_subsample(const uint4 *__restrict__ in, uint32_t log_n, uint32_t log_out, size_t total, uint4 *__restrict__ out) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const size_t b = e >> log_out, i = e & (((size_t)1 << log_out) - 1);
    const size_t s = (b << log_n) + (i << (log_n - log_out));
    out[2 * e] = in[2 * s];
    out[2 * e + 1] = in[2 * s + 1];
}

// canonical -> Montgomery (canonical representative) for a short table of constants
template <class U>
__global__ void fr_table_to_mont(const uint32_t *__restrict__ canon, uint32_t count, uint32_t *__restrict__ mont) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    fu_store<U>(mont + (size_t)i * U::SL, fu_cond_sub_p(fu_from_canonical<U>(canon + (size_t)i * U::NL)));
}

// Horner machinery shared by evaluation and division by (X - z).  A lane owns HORNER_CHUNK consecutive
// coefficients, a workgroup of 256 lanes owns HORNER_BLOCK = 8192.
//   S_t   = sum_{j in chunk t} f[j] z^(j - chunk start)                      (local Horner)
//   V_t   = sum_{u >= t} S_u z^(C (u - t))  within the workgroup             (suffix scan, log steps, z^(C 2^k))
//   P_b   = V_0 of workgroup b; G over workgroups by a short sequential pass  (poly_block_carry)
static constexpr uint32_t HORNER_CHUNK = 32, HORNER_BLOCK = 256 * HORNER_CHUNK;

// zpow[p][k] = z_p^(C 2^k) for k = 0..8 (k = 8: z^HORNER_BLOCK), zpow[p][9] = z_p; Montgomery form
template <class U>
__global__ void horner_setup(const uint32_t *__restrict__ points_c, uint32_t npoints, uint32_t *__restrict__ zpow) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npoints) return;
    Fu<U> z = fu_cond_sub_p(fu_from_canonical<U>(points_c + (size_t)p * U::NL));
    fu_store<U>(zpow + ((size_t)p * 10 + 9) * U::SL, z);
    Fu<U> w = z;
    for (int i = 0; i < 5; ++i) w = fu_mul(w, w);  // z^32
    for (int k = 0; k <= 8; ++k) {
        w = fu_cond_sub_p(w);
        fu_store<U>(zpow + ((size_t)p * 10 + k) * U::SL, w);
        w = fu_mul(w, w);
    }
}

template <class U>
ZK_D Fu<U> horner_chunk(const uint32_t *__restrict__ f, size_t n, size_t start, const Fu<U> &z) {
    Fu<U> acc = Fu<U>::zero();
    for (int j = HORNER_CHUNK - 1; j >= 0; --j) {
        size_t e = start + j;
        acc = fu_mul(acc, z);
        if (e < n) acc = fu_add(acc, fu_unpack<U>(f + e * U::NL));
    }
    return acc;  // < 3p
}

// suffix scan over the 256 lanes of a workgroup: returns V_t (and leaves all V in lds)
template <class U>
ZK_D Fu<U> horner_block_scan(uint32_t *lds, Fu<U> v, uint32_t t, const uint32_t *__restrict__ zp) {
    fu_store<U>(lds + (size_t)t * U::SL, v);
    __syncthreads();
    for (uint32_t k = 0, d = 1; d < 256; ++k, d <<= 1) {
        Fu<U> add = Fu<U>::zero();
        const bool has = t + d < 256;
        if (has) add = fu_mul(fu_load<U>(lds + (size_t)(t + d) * U::SL), fu_load<U>(zp + (size_t)k * U::SL));
        __syncthreads();
        if (has) {
            v = fu_cond_sub_p(fu_mul(fu_add(v, add), Fu<U>::one()));  // back under p: the next level adds again
            fu_store<U>(lds + (size_t)t * U::SL, v);
        }
        __syncthreads();
    }
    return v;
}

// grid (blocks per polynomial, batch, npoints): part[(poly * npoints + p) * nblk + b] = P_b
template <class U>
__global__ __launch_bounds__(256) void poly_block_horner(const uint32_t *__restrict__ polys, size_t n, size_t stride, uint32_t nblk, uint32_t npoints,
                                                         const uint32_t *__restrict__ zpow, uint32_t *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * U::SL];
    const uint32_t t = threadIdx.x, b = blockIdx.x, poly = blockIdx.y, p = blockIdx.z;
    const uint32_t *f = polys + (size_t)poly * stride * U::NL;
    const uint32_t *zp = zpow + (size_t)p * 10 * U::SL;
    Fu<U> s = horner_chunk<U>(f, n, (size_t)b * HORNER_BLOCK + (size_t)t * HORNER_CHUNK, fu_load<U>(zp + 9 * U::SL));
    Fu<U> v = horner_block_scan<U>(lds, fu_cond_sub_p(fu_mul(s, Fu<U>::one())), t, zp);
    if (t == 0) fu_store<U>(part + (((size_t)poly * npoints + p) * nblk + b) * U::SL, v);
}

// one lane per (polynomial, point): carry[b] = G at the start of workgroup b + 1 (0 for the last), value = G_0 = f(z)
template <class U>
__global__ void poly_block_carry(const uint32_t *__restrict__ part, uint32_t nblk, uint32_t npoints, uint32_t total, const uint32_t *__restrict__ zpow,
                                 uint32_t *__restrict__ carry, uint32_t *__restrict__ values) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;  // = poly * npoints + p
    if (g >= total) return;
    const uint32_t p = g % npoints;
    const Fu<U> zb = fu_load<U>(zpow + ((size_t)p * 10 + 8) * U::SL);
    Fu<U> acc = Fu<U>::zero();
    for (int b = (int)nblk - 1; b >= 0; --b) {
        if (carry) fu_store<U>(carry + ((size_t)g * nblk + b) * U::SL, acc);
        acc = fu_cond_sub_p(fu_mul(fu_add(fu_mul(acc, zb), fu_load<U>(part + ((size_t)g * nblk + b) * U::SL)), Fu<U>::one()));
    }
    fu_pack<U>(values + (size_t)g * U::NL, acc);
}

// division by (X - z): out[j] = G_j = sum_{t >= j} f[t] z^(t - j);  out[0] = f(z), out[1..n) = quotient.  In place allowed.
template <class U>
__global__ __launch_bounds__(256) void poly_div_finish(const uint32_t *f, size_t n, uint32_t nblk, const uint32_t *__restrict__ zpow,
                                                       const uint32_t *__restrict__ carry, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * U::SL];
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const size_t start = (size_t)b * HORNER_BLOCK + (size_t)t * HORNER_CHUNK;
    const Fu<U> z = fu_load<U>(zpow + 9 * U::SL);
    Fu<U> s = horner_chunk<U>(f, n, start, z);
    horner_block_scan<U>(lds, fu_cond_sub_p(fu_mul(s, Fu<U>::one())), t, zpow);
    // G at the end of this lane's chunk: V_{t+1} + z^(C (255 - t)) * carry-in of the workgroup
    Fu<U> cin = fu_load<U>(carry + (size_t)b * U::SL);
    Fu<U> pw = fu_cond_sub_p(Fu<U>::one());
    for (uint32_t e = 255 - t, k = 0; e; e >>= 1, ++k)
        if (e & 1) pw = fu_mul(pw, fu_load<U>(zpow + (size_t)k * U::SL));
    Fu<U> g = fu_mul(cin, pw);
    if (t + 1 < 256) g = fu_add(g, fu_load<U>(lds + (size_t)(t + 1) * U::SL));
    for (int j = HORNER_CHUNK - 1; j >= 0; --j) {
        size_t e = start + j;
        if (e >= n) continue;  // beyond the top coefficient: G = 0 there, g is still 0
        g = fu_cond_sub_p(fu_mul(fu_add(fu_mul(g, z), fu_unpack<U>(f + e * U::NL)), Fu<U>::one()));
        fu_pack<U>(out + e * U::NL, g);
    }
}

// acc[j] (+)= sum_i sum_{t < taps} c[i][t] * poly_i[j - t]   (poly_i = 0 outside [0, len_i)); c in Montgomery form
template <class U>
__global__ __launch_bounds__(256) void poly_lincomb(const uint32_t *const *__restrict__ polys, const uint64_t *__restrict__ lens, uint32_t count,
                                                    uint32_t taps, const uint32_t *__restrict__ coeff, size_t acc_len, int accumulate,
                                                    uint32_t *__restrict__ acc_out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= acc_len) return;
    Fu<U> acc = Fu<U>::zero();
    uint32_t pending = 0;
    if (accumulate) acc = fu_unpack<U>(acc_out + j * U::NL), pending = 1;
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t *f = polys[i];
        const size_t len = lens[i];
        for (uint32_t t = 0; t < taps; ++t) {
            if (j < t || j - t >= len) continue;
            acc = fu_add(acc, fu_mul(fu_unpack<U>(f + (j - t) * U::NL), fu_load<U>(coeff + ((size_t)i * taps + t) * U::SL)));
            if (++pending == 16) {  // each term is < 2p and the limbs hold ~64p: fold back under 2p
                acc = fu_mul(acc, Fu<U>::one());
                pending = 1;
            }
        }
    }
    fu_pack<U>(acc_out + j * U::NL, fu_cond_sub_p(fu_mul(acc, Fu<U>::one())));
}


// shared front half of evaluation and division: per-(polynomial, point) workgroup partials and the pass over them
template <class U>
static int horner_run(zkhip_ctx *ctx, const uint32_t *d_polys, size_t n, size_t stride, size_t batch, const uint64_t *points, size_t npoints,
                      uint32_t **zpow_out, uint32_t **carry_out, uint32_t **values_out, bool want_carry) {
    const uint32_t nblk = (uint32_t)((n + HORNER_BLOCK - 1) / HORNER_BLOCK);
    const size_t total = batch * npoints;
    size_t need = zkhip_ctx::ws_round(npoints * 32) + zkhip_ctx::ws_round(npoints * 10 * U::SL * 4) + zkhip_ctx::ws_round(total * nblk * U::SL * 4) +
                  zkhip_ctx::ws_round(total * nblk * U::SL * 4) + zkhip_ctx::ws_round(total * 32);
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    uint32_t *d_pts = ctx->ws_take<uint32_t>(npoints * 8);
    uint32_t *zpow = ctx->ws_take<uint32_t>(npoints * 10 * U::SL);
    uint32_t *part = ctx->ws_take<uint32_t>(total * nblk * U::SL);
    uint32_t *carry = ctx->ws_take<uint32_t>(total * nblk * U::SL);
    uint32_t *values = ctx->ws_take<uint32_t>(total * 8);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_pts, points, npoints * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(ctx, "poly_horner_setup", horner_setup<U>, dim3((unsigned)((npoints + 63) / 64)), dim3(64), 0, d_pts, (uint32_t)npoints, zpow);
    ZK_LAUNCH(ctx, "poly_block_horner", poly_block_horner<U>, dim3(nblk, (unsigned)batch, (unsigned)npoints), dim3(256), 0, d_polys, n, stride, nblk,
              (uint32_t)npoints, zpow, part);
    ZK_LAUNCH(ctx, "poly_block_carry", poly_block_carry<U>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, part, nblk, (uint32_t)npoints, (uint32_t)total,
              zpow, want_carry ? carry : nullptr, values);
    *zpow_out = zpow, *carry_out = carry, *values_out = values;
    return ZKHIP_OK;
}

// omega_out^(2^log_k) == omega_n ?  (host arithmetic: the C++ bodies of fu.hpp) -- what the coset extension below relies on
template <class U>
static bool roots_nested(const uint64_t *omega_out, size_t log_k, const uint64_t *omega_n) {
    Fu<U> x = fu_from_canonical<U>(reinterpret_cast<const uint32_t *>(omega_out));
    for (size_t i = 0; i < log_k; ++i) x = fu_mul(x, x);
    uint32_t c[U::NL];
    fu_to_canonical<U>(c, x);
    return memcmp(c, omega_n, U::NL * 4) == 0;
}

extern "C" {

int zkhip_poly_resize_dev(zkhip_ctx *ctx, int curve, void *d_in, size_t log_n, size_t batch, const uint64_t *omega_n, void *d_out, size_t log_out,
                          const uint64_t *omega_out) {
    if (!ctx || !omega_n || !omega_out || (batch && (!d_in || !d_out))) return ZKHIP_ERR_INVALID;
    if (log_out > 32 || log_n > 32) return ZKHIP_ERR_RANGE;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (batch == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (log_out < log_n) {  // smaller domain: every 2^(log_n - log_out)-th evaluation (the caller vouches for degree < 2^log_out); d_in is left as it is
        const size_t tot = batch << log_out;
        if (batch >= ((size_t)1 << 31) || tot >= ((size_t)1 << 39)) return ZKHIP_ERR_RANGE;  // (tot + 255) / 256 workgroups < 2^31
        ZK_LAUNCH(ctx, "poly_subsample", poly_subsample, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (const uint4 *)d_in, (uint32_t)log_n,
                  (uint32_t)log_out, tot, (uint4 *)d_out);
        return ZKHIP_OK;
    }
    const size_t log_k = log_out - log_n;
    // The coset extension places the n known values at positions i K and evaluates the other cosets with omega_out^j: that is the K n-point
    // vector only when the two roots are NESTED, omega_out^K == omega_n (roots taken from one generator always are).  Any other pair of
    // primitive roots takes the general path below, which is correct for every pair (the relation used to be assumed).
    bool nested = false;
    if (log_k >= 1 && log_k <= 4) { ZK_FR_DISPATCH(curve, nested = roots_nested<U>(omega_out, log_k, omega_n)); }
    if (ctx->opt_poly_coset_extend && nested && log_n >= 1 && batch * (((size_t)1 << log_k) - 1) < ((size_t)1 << 20)) {
        // Round 5: the K n-point domain is the n-point one and its K - 1 cosets omega_out^j <omega_n>.  The n known values are copied to their
        // places, the coefficients (inverse transform in place: d_in is consumed as before) are evaluated on the K - 1 new cosets by n-point
        // transforms that store straight into theirs: K n transform points instead of (K + 1) n, in transforms of the smaller size, and a lone
        // polynomial still fills the pairs-per-workgroup kernel with two of its cosets.
        const size_t tot = batch << log_n;
        ZK_LAUNCH(ctx, "poly_spread", poly_spread, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (const uint4 *)d_in, (uint32_t)log_k, tot, (uint32_t)log_n,
                  (uint4 *)d_out);
        ZK_TRY(zk_ntt_run(ctx, curve, (uint32_t *)d_in, log_n, batch, omega_n, 1, nullptr));
        return zk_ntt_extend(ctx, curve, (uint32_t *)d_in, log_n, batch, omega_n, (uint32_t *)d_out, log_k, omega_out);
    }
    // coefficients in place (d_in is consumed), zero-extended copy, evaluation on the larger domain
    ZK_TRY(zk_ntt_run(ctx, curve, (uint32_t *)d_in, log_n, batch, omega_n, 1, nullptr));
    size_t total = batch << log_out;
    ZK_LAUNCH(ctx, "poly_pad_copy", poly_pad_copy, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (const uint4 *)d_in, (uint32_t)log_n,
              (uint32_t)log_out, total, (uint4 *)d_out);
    return zk_ntt_run(ctx, curve, (uint32_t *)d_out, log_out, batch, omega_out, 0, nullptr);
}

int zkhip_fri_fold_dev(zkhip_ctx *ctx, int curve, const void *d_f, size_t log_size, const uint64_t *alpha, const uint64_t *omega, void *d_out) {
    if (!ctx || !d_f || !alpha || !omega || !d_out) return ZKHIP_ERR_INVALID;
    if (log_size < 1 || log_size > 32) return ZKHIP_ERR_RANGE;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ZK_TRY(ctx->ws_reserve(4096));
    ctx->ws_reset();
    uint32_t *d_c = ctx->ws_take<uint32_t>(16);
    uint32_t *consts = ctx->ws_take<uint32_t>(64);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_c, alpha, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_c + 8, omega, 32, hipMemcpyHostToDevice, ctx->stream));
    size_t half = (size_t)1 << (log_size - 1), lanes = (half + FOLD_CHUNK - 1) / FOLD_CHUNK;
    dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
    if (curve == CURVE_BLS12_381) {
        ZK_LAUNCH(ctx, "fri_fold_setup", fri_fold_setup<BlsFrU>, dim3(1), dim3(64), 0, d_c, d_c + 8, (uint32_t)log_size, consts);
        ZK_LAUNCH(ctx, "fri_fold", fri_fold<BlsFrU>, grid, block, 0, (const uint32_t *)d_f, (uint32_t)log_size, consts, (uint32_t *)d_out);
    } else {
        ZK_LAUNCH(ctx, "fri_fold_setup", fri_fold_setup<BnFrU>, dim3(1), dim3(64), 0, d_c, d_c + 8, (uint32_t)log_size, consts);
        ZK_LAUNCH(ctx, "fri_fold", fri_fold<BnFrU>, grid, block, 0, (const uint32_t *)d_f, (uint32_t)log_size, consts, (uint32_t *)d_out);
    }
    return ZKHIP_OK;
}

int zkhip_fri_leaves_dev(zkhip_ctx *ctx, const void *d_polys, size_t log_domain, size_t batch, size_t fri_step, void *d_out) {
    if (!ctx || (batch && (!d_polys || !d_out))) return ZKHIP_ERR_INVALID;
    if (fri_step < 1 || fri_step > log_domain || log_domain > 32 || batch >= ((size_t)1 << 31)) return ZKHIP_ERR_RANGE;
    if (batch == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t total = (batch << log_domain) >> 1;  // pairs
    ZK_LAUNCH(ctx, "fri_leaf_gather", fri_leaf_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (const uint4 *)d_polys, (uint32_t)log_domain,
              (uint32_t)batch, (uint32_t)fri_step, total, (uint4 *)d_out);
    return ZKHIP_OK;
}

int zkhip_fr_vec_op_dev(zkhip_ctx *ctx, int curve, int op, const void *d_a, const void *d_b, void *d_out, size_t count) {
    if (!ctx || (count && (!d_a || !d_b || !d_out)) || op < 0 || op > 2) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (count == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ZK_FR_DISPATCH(curve, ZK_LAUNCH(ctx, "fr_vec_op", fr_vec_op<U>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (const uint32_t *)d_a,
                                    (const uint32_t *)d_b, count, op, (uint32_t *)d_out));
    return ZKHIP_OK;
}

int zkhip_fr_vec_affine_dev(zkhip_ctx *ctx, int curve, const void *d_x, const void *d_y, const uint64_t *a, const uint64_t *b, const uint64_t *c, void *d_out,
                            size_t count) {
    if (!ctx || !a || !c || (d_y && !b) || (count && (!d_x || !d_out))) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (count >= ((size_t)1 << 39)) return ZKHIP_ERR_RANGE;
    if (count == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    FrWords wa, wb = {}, wc;
    memcpy(wa.w, a, 32);
    if (b) memcpy(wb.w, b, 32);
    memcpy(wc.w, c, 32);
    ZK_FR_DISPATCH(curve, ZK_LAUNCH(ctx, "fr_vec_affine", fr_vec_affine<U>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (const uint32_t *)d_x,
                                    (const uint32_t *)d_y, wa, wb, wc, count, (uint32_t *)d_out));
    return ZKHIP_OK;
}

int zkhip_poly_div_vanishing_dev(zkhip_ctx *ctx, int curve, const void *d_f, size_t len, size_t n, void *d_quot, uint64_t *nonzero_remainders) {
    if (!ctx || n == 0 || (len && !d_f) || (len > n && !d_quot)) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (n >= ((size_t)1 << 39) || len >= ((size_t)1 << 40)) return ZKHIP_ERR_RANGE;
    if (nonzero_remainders) *nonzero_remainders = 0;
    if (len == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ZK_TRY(ctx->ws_reserve(zkhip_ctx::ws_round(4)));
    ctx->ws_reset();
    uint32_t *d_bad = ctx->ws_take<uint32_t>(1);
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    const size_t lanes = std::min(n, len);
    ZK_FR_DISPATCH(curve, ZK_LAUNCH(ctx, "poly_div_vanishing", poly_div_vanishing<U>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0,
                                    (const uint32_t *)d_f, len, n, (uint32_t *)d_quot, d_bad));
    if (nonzero_remainders) {
        uint32_t bad = 0;
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        *nonzero_remainders = bad;
    }
    return ZKHIP_OK;
}

int zkhip_fr_vec_prod_dev(zkhip_ctx *ctx, int curve, size_t count, const void *const *d_in, void *d_out, size_t n) {
    if (!ctx || !d_in || count == 0 || (n && !d_out)) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (count >= 65536 || n >= ((size_t)1 << 39)) return ZKHIP_ERR_RANGE;  // (n + 255) / 256 workgroups < 2^31
    for (size_t k = 0; k < count; ++k)
        if (n && !d_in[k]) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ZK_TRY(ctx->ws_reserve(zkhip_ctx::ws_round(count * sizeof(void *))));
    ctx->ws_reset();
    const uint32_t **d_ptrs = ctx->ws_take<const uint32_t *>(count);
    ctx->batch_ptrs.assign((uint32_t *const *)d_in, (uint32_t *const *)d_in + count);  // host copy alive until the async copy ran
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_ptrs, ctx->batch_ptrs.data(), count * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    ZK_FR_DISPATCH(curve, ZK_LAUNCH(ctx, "fr_vec_prod", fr_vec_prod<U>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_ptrs, (uint32_t)count, n,
                                    (uint32_t *)d_out));
    return ZKHIP_OK;
}

int zkhip_poly_shift_dev(zkhip_ctx *ctx, const void *d_in, size_t log_size, int64_t rotation, void *d_out) {
    if (!ctx || !d_in || !d_out || d_in == d_out) return ZKHIP_ERR_INVALID;
    if (log_size > 31) return ZKHIP_ERR_RANGE;  // one lane per element: (n + 255) / 256 workgroups must fit the 2^31 grid limit
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_size;
    const int64_t m = (int64_t)n;
    const size_t rot = (size_t)(((rotation % m) + m) % m);
    ZK_LAUNCH(ctx, "poly_rotate", poly_rotate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const uint4 *)d_in, (uint32_t)log_size, rot, (uint4 *)d_out);
    return ZKHIP_OK;
}

int zkhip_poly_eval_dev(zkhip_ctx *ctx, int curve, const void *d_polys, size_t n, size_t stride, size_t batch, const uint64_t *points, size_t npoints,
                        uint64_t *out) {
    if (!ctx || !out || (batch && npoints && (!d_polys || !points))) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (stride < n || batch >= 65536 || npoints >= 65536 || n >= ((size_t)1 << 40)) return ZKHIP_ERR_RANGE;
    if (batch == 0 || npoints == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint32_t *zpow, *carry, *values;
    ZK_FR_DISPATCH(curve, ZK_TRY(horner_run<U>(ctx, (const uint32_t *)d_polys, n, stride, batch, points, npoints, &zpow, &carry, &values, false)));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(out, values, batch * npoints * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_poly_div_linear_dev(zkhip_ctx *ctx, int curve, const void *d_f, size_t n, const uint64_t *z, void *d_out, uint64_t *remainder) {
    if (!ctx || !z || (n && (!d_f || !d_out))) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (n >= ((size_t)1 << 40)) return ZKHIP_ERR_RANGE;
    if (n == 0) {
        if (remainder) remainder[0] = remainder[1] = remainder[2] = remainder[3] = 0;
        return ZKHIP_OK;
    }
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint32_t *zpow, *carry, *values;
    const uint32_t nblk = (uint32_t)((n + HORNER_BLOCK - 1) / HORNER_BLOCK);
    ZK_FR_DISPATCH(curve, ZK_TRY(horner_run<U>(ctx, (const uint32_t *)d_f, n, n, 1, z, 1, &zpow, &carry, &values, true));
                   ZK_LAUNCH(ctx, "poly_div_finish", poly_div_finish<U>, dim3(nblk), dim3(256), 0, (const uint32_t *)d_f, n, nblk, zpow, carry,
                             (uint32_t *)d_out));
    if (remainder) {
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(remainder, values, 32, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ZKHIP_OK;
}

int zkhip_poly_lincomb_dev(zkhip_ctx *ctx, int curve, size_t count, const void *const *d_polys, const size_t *lens, const uint64_t *coeffs, size_t taps,
                           void *d_acc, size_t acc_len, int accumulate) {
    if (!ctx || (acc_len && !d_acc) || (count && (!d_polys || !lens || !coeffs || taps == 0))) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (count >= ((size_t)1 << 24) || taps >= 4096) return ZKHIP_ERR_RANGE;
    if (acc_len == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t nc = std::max<size_t>(1, count * taps);
    size_t need = zkhip_ctx::ws_round(nc * 32) + zkhip_ctx::ws_round(nc * 16 * 4) + zkhip_ctx::ws_round((count + 1) * 8) * 2;
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    uint32_t *d_c = ctx->ws_take<uint32_t>(nc * 8);
    uint32_t *d_m = ctx->ws_take<uint32_t>(nc * 16);
    const uint32_t **d_p = ctx->ws_take<const uint32_t *>(count + 1);
    uint64_t *d_l = ctx->ws_take<uint64_t>(count + 1);
    if (count) {
        // the pointer / length / coefficient tables are staged through the context so the caller's arrays may die on return
        ctx->lincomb_stage.resize(count * 2);
        for (size_t i = 0; i < count; ++i) ctx->lincomb_stage[i] = (uint64_t)(uintptr_t)d_polys[i], ctx->lincomb_stage[count + i] = lens[i];
        ctx->lincomb_coeffs.assign(coeffs, coeffs + count * taps * 4);
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_p, ctx->lincomb_stage.data(), count * 8, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_l, ctx->lincomb_stage.data() + count, count * 8, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_c, ctx->lincomb_coeffs.data(), count * taps * 32, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // pageable sources: the copies above must not outlive the staging vectors' next reuse
    }
    ZK_FR_DISPATCH(curve, static_assert(U::SL <= 16, "coefficient slot");
                   if (count) ZK_LAUNCH(ctx, "poly_lincomb_setup", fr_table_to_mont<U>, dim3((unsigned)((count * taps + 63) / 64)), dim3(64), 0, d_c,
                                        (uint32_t)(count * taps), d_m);
                   ZK_LAUNCH(ctx, "poly_lincomb", poly_lincomb<U>, dim3((unsigned)((acc_len + 255) / 256)), dim3(256), 0, d_p, d_l, (uint32_t)count,
                             (uint32_t)taps, d_m, acc_len, accumulate, (uint32_t *)d_acc));
    return ZKHIP_OK;
}

}  // extern "C"
