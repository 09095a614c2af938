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

static constexpr uint32_t FOLD_CHUNK = 128;  // consecutive i per lane: one power, then a running product

// consts = [alpha, w^-1, 1/2] in Montgomery form, canonical representatives
template <class U>
__global__ void fri_fold_setup(const uint32_t *__restrict__ alpha_c, const uint32_t *__restrict__ omega_c, uint32_t *__restrict__ consts) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    fu_store<U>(consts, fu_cond_sub_p(fu_from_canonical<U>(alpha_c)));
    fu_store<U>(consts + U::SL, fu_cond_sub_p(fu_inv(fu_from_canonical<U>(omega_c))));
    fu_store<U>(consts + 2 * U::SL, fu_cond_sub_p(fu_inv(fu_add(Fu<U>::one(), Fu<U>::one()))));
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

extern "C" {

int zkhip_poly_resize_dev(zkhip_ctx *ctx, int curve, void *d_in, size_t log_n, size_t batch, const uint64_t *omega_n, void *d_out, size_t log_out,
                          const uint64_t *omega_out) {
    if (!ctx || !omega_n || !omega_out || (batch && (!d_in || !d_out))) return ZKHIP_ERR_INVALID;
    if (log_out < log_n || log_out > 32) return ZKHIP_ERR_RANGE;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (batch == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
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
        ZK_LAUNCH(ctx, "fri_fold_setup", fri_fold_setup<BlsFrU>, dim3(1), dim3(64), 0, d_c, d_c + 8, consts);
        ZK_LAUNCH(ctx, "fri_fold", fri_fold<BlsFrU>, grid, block, 0, (const uint32_t *)d_f, (uint32_t)log_size, consts, (uint32_t *)d_out);
    } else {
        ZK_LAUNCH(ctx, "fri_fold_setup", fri_fold_setup<BnFrU>, dim3(1), dim3(64), 0, d_c, d_c + 8, consts);
        ZK_LAUNCH(ctx, "fri_fold", fri_fold<BnFrU>, grid, block, 0, (const uint32_t *)d_f, (uint32_t)log_size, consts, (uint32_t *)d_out);
    }
    return ZKHIP_OK;
}

}  // extern "C"
