// The gate argument's sum in ONE pass over the extended domain (include/zkhip.h: zkhip_gate_eval_dev).
//
// placeholder's gates argument (zk/snark/systems/plonk/placeholder/gates_argument.hpp:93-121, 203-216) evaluates
//     F = mask * sum_gates selector_g * sum_constraints theta^k * constraint(assignment columns, rotated)
// on the extended domain: once the symbolic walk (math::expression, the visitors -- the caller's) has flattened a constraint into
// monomials this is a FLAT PROGRAM: gates, each with an optional selector and a range of terms; terms, each a coefficient and a list of
// factors; factors, each a (column slot, rotation).  Round 5 ran it as two launches per term (a k-way product + a scaled accumulation,
// each streaming whole vectors) and materialised every rotated column; here a lane owns one row of the extended domain and walks the
// whole program: a rotation is index arithmetic ((row + rotation) mod size), every distinct column is read where it lies, the selector
// and the mask are multiplied in the same pass, and the only vector written is F.
//
// Arithmetic.  Vectors are canonical in HBM; fu_mul(x, y) = x y / R.  A term's accumulator starts at c R^(k+1) (k factors; one more R when
// its gate has a selector) -- computed once per term by gate_coeff_setup -- so that k products by CANONICAL factors leave c prod(x) R: no
// per-factor lift (fr_vec_prod pays two products per factor), the coefficient costs nothing.  Terms add up unreduced (each < 2p; folded
// by a product with R mod p every 16 additions), a gate's sum times its canonical selector drops one R, and the last product -- by the
// canonical mask, or by the plain 1 -- drops the other: the result is canonical.  The same entry point evaluates the expressions of
// prepare_lookup_input (lookup_argument.hpp:435-496): no selector, no mask.
#include <algorithm>

#include "ctx.hpp"
#include "fu.hpp"

using namespace zkhip;

namespace {

// device image of the program, u32 words:  gates: [term_begin, term_end, selector_slot | NONE, selector_rotation] x n_gates
//                                          terms: [factor_begin, factor_end] x n_terms        factors: [slot, rotation] x n_factors
constexpr uint32_t GATE_NO_SELECTOR = 0xFFFFFFFFu;

// m[t] = c[t] R^(lift[t]) in storage form (lift >= 1): lift products by R^2
template <class U>
__global__ void gate_coeff_setup(const uint32_t *__restrict__ canon, const uint32_t *__restrict__ lift, uint32_t count, uint32_t *__restrict__ mont) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    Fu<U> m = fu_unpack<U>(canon + (size_t)t * U::NL);
    for (uint32_t k = 0; k < lift[t]; ++k) m = fu_mul(m, Fu<U>::r2());
    fu_store<U>(mont + (size_t)t * U::SL, m);
}

template <class U>
__global__ __launch_bounds__(256) void gate_eval(const uint32_t *const *__restrict__ slots, const uint32_t *__restrict__ gates, const uint32_t *__restrict__ terms,
                                                 const uint32_t *__restrict__ factors, const uint32_t *__restrict__ coeff_m, uint32_t n_gates, uint32_t log_size,
                                                 const uint32_t *__restrict__ mask, int accumulate, uint32_t *__restrict__ out) {
    const size_t size = (size_t)1 << log_size, row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= size) return;
    const size_t wrap = size - 1;
    // the running total carries ONE factor R (dropped by the last product)
    Fu<U> total = Fu<U>::zero();
    uint32_t total_pending = 0;
    if (accumulate) total = fu_mul(fu_unpack<U>(out + row * U::NL), Fu<U>::r2()), total_pending = 1;
    for (uint32_t g = 0; g < n_gates; ++g) {
        const uint32_t t0 = gates[4 * g], t1 = gates[4 * g + 1], sel = gates[4 * g + 2];
        Fu<U> sum = Fu<U>::zero();
        uint32_t pending = 0;
        for (uint32_t t = t0; t < t1; ++t) {
            const uint32_t f0 = terms[2 * t], f1 = terms[2 * t + 1];
            Fu<U> acc = fu_load<U>(coeff_m + (size_t)t * U::SL);
            for (uint32_t f = f0; f < f1; ++f) {
                const uint32_t *col = slots[factors[2 * f]];
                const size_t at = (row + (size_t)(int64_t)(int32_t)factors[2 * f + 1]) & wrap;  // two's complement: a negative rotation wraps too
                acc = fu_mul(acc, fu_unpack<U>(col + at * U::NL));
            }
            sum = fu_add(sum, acc);
            if (++pending == 16) {  // 16 terms of < 2p each: back under 2p (value unchanged: a product by R mod p)
                sum = fu_mul(sum, Fu<U>::one());
                pending = 1;
            }
        }
        if (sel != GATE_NO_SELECTOR) {
            const size_t at = (row + (size_t)(int64_t)(int32_t)gates[4 * g + 3]) & wrap;
            sum = fu_mul(sum, fu_unpack<U>(slots[sel] + at * U::NL));  // terms of this gate were lifted by one more R
            pending = 1;
        }
        if (total_pending + pending > 16) {  // at most 17 summands of < 2p ever sit in `total`: 34p < R
            total = fu_mul(total, Fu<U>::one());
            total_pending = 1;
        }
        total = fu_add(total, sum);
        total_pending += pending;
    }
    Fu<U> last;
    if (mask) last = fu_unpack<U>(mask + row * U::NL);
    else {
        last = Fu<U>::zero();
        last.v[0] = 1;
    }
    fu_pack<U>(out + row * U::NL, fu_cond_sub_p(fu_mul(total, last)));
}

}  // namespace

#define ZK_FR_DISPATCH(curve, ...)      \
    if ((curve) == CURVE_BLS12_381) {   \
        typedef BlsFrU U;               \
        __VA_ARGS__;                    \
    } else {                            \
        typedef BnFrU U;                \
        __VA_ARGS__;                    \
    }

extern "C" int zkhip_gate_eval_dev(zkhip_ctx *ctx, int curve, const zkhip_gate_program *prog, const void *const *d_slots, size_t log_size, const void *d_mask,
                                   int accumulate, void *d_out) {
    if (!ctx || !prog || !d_out) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381 && curve != CURVE_BN254) return ZKHIP_ERR_INVALID;
    if (log_size > 31) return ZKHIP_ERR_RANGE;  // one lane per row: (size + 255) / 256 workgroups within the grid limit
    const uint32_t G = prog->n_gates, T = prog->n_terms, F = prog->n_factors, S = prog->n_slots;
    if ((G && (!prog->gate_terms || !prog->gate_selector || !prog->gate_selector_rot)) || (T && (!prog->term_factors || !prog->term_coeff)) ||
        (F && (!prog->factor_slot || !prog->factor_rot)) || (S && !d_slots))
        return ZKHIP_ERR_INVALID;
    if (T >= (1u << 24) || F >= (1u << 26) || S >= (1u << 20)) return ZKHIP_ERR_RANGE;
    // the program is validated on the host: ranges monotone and inside their tables, every slot named exists -- a kernel walking a
    // malformed program would read out of bounds
    if (G) {
        if (prog->gate_terms[0] != 0 || prog->gate_terms[G] != T) return ZKHIP_ERR_INVALID;
        for (uint32_t g = 0; g < G; ++g) {
            if (prog->gate_terms[g] > prog->gate_terms[g + 1]) return ZKHIP_ERR_INVALID;
            if (prog->gate_selector[g] != GATE_NO_SELECTOR && prog->gate_selector[g] >= S) return ZKHIP_ERR_RANGE;
        }
    } else if (T) return ZKHIP_ERR_INVALID;
    if (T) {
        if (prog->term_factors[0] != 0 || prog->term_factors[T] != F) return ZKHIP_ERR_INVALID;
        for (uint32_t t = 0; t < T; ++t)
            if (prog->term_factors[t] > prog->term_factors[t + 1]) return ZKHIP_ERR_INVALID;
    } else if (F) return ZKHIP_ERR_INVALID;
    for (uint32_t f = 0; f < F; ++f)
        if (prog->factor_slot[f] >= S) return ZKHIP_ERR_RANGE;
    for (uint32_t s = 0; s < S; ++s)
        if (!d_slots[s]) return ZKHIP_ERR_INVALID;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));

    // host image: canonical coefficients (32-byte entries first: aligned) | pointers | gates | terms | factors | lifts, one upload
    const size_t w_coeff = (size_t)T * 8, w_ptr = (size_t)S * 2, w_gate = (size_t)G * 4, w_term = (size_t)T * 2, w_fac = (size_t)F * 2, w_lift = T;
    std::vector<uint32_t> &img = ctx->gate_stage;
    img.assign(w_coeff + w_ptr + w_gate + w_term + w_fac + w_lift + 8, 0);
    uint32_t *h_coeff = img.data(), *h_ptr = h_coeff + w_coeff, *h_gate = h_ptr + w_ptr, *h_term = h_gate + w_gate, *h_fac = h_term + w_term, *h_lift = h_fac + w_fac;
    for (uint32_t s = 0; s < S; ++s) {
        const uint64_t p = (uint64_t)(uintptr_t)d_slots[s];
        h_ptr[2 * s] = (uint32_t)p, h_ptr[2 * s + 1] = (uint32_t)(p >> 32);
    }
    for (uint32_t g = 0; g < G; ++g) {
        h_gate[4 * g] = prog->gate_terms[g], h_gate[4 * g + 1] = prog->gate_terms[g + 1];
        h_gate[4 * g + 2] = prog->gate_selector[g], h_gate[4 * g + 3] = (uint32_t)prog->gate_selector_rot[g];
        for (uint32_t t = prog->gate_terms[g]; t < prog->gate_terms[g + 1]; ++t)
            h_lift[t] = (prog->term_factors[t + 1] - prog->term_factors[t]) + 1 + (prog->gate_selector[g] != GATE_NO_SELECTOR ? 1 : 0);
    }
    for (uint32_t t = 0; t < T; ++t) h_term[2 * t] = prog->term_factors[t], h_term[2 * t + 1] = prog->term_factors[t + 1];
    for (uint32_t f = 0; f < F; ++f) h_fac[2 * f] = prog->factor_slot[f], h_fac[2 * f + 1] = (uint32_t)prog->factor_rot[f];
    if (T) memcpy(h_coeff, prog->term_coeff, (size_t)T * 32);

    const size_t img_bytes = img.size() * 4, mont_bytes = std::max<size_t>(1, T) * 16 * 4;
    ZK_TRY(ctx->ws_reserve(zkhip_ctx::ws_round(img_bytes) + zkhip_ctx::ws_round(mont_bytes)));
    ctx->ws_reset();
    uint32_t *d_img = ctx->ws_take<uint32_t>(img.size());
    uint32_t *d_mont = ctx->ws_take<uint32_t>(std::max<size_t>(1, T) * 16);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_img, img.data(), img_bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // pageable source: the staging vector may be refilled by the next call
    const uint32_t *dd_coeff = d_img;
    const uint32_t *const *dd_ptr = reinterpret_cast<const uint32_t *const *>(d_img + w_coeff);
    const uint32_t *dd_gate = d_img + w_coeff + w_ptr, *dd_term = dd_gate + w_gate, *dd_fac = dd_term + w_term, *dd_lift = dd_fac + w_fac;
    const size_t size = (size_t)1 << log_size;
    ZK_FR_DISPATCH(curve, static_assert(U::SL <= 16, "coefficient slot");
                   if (T) ZK_LAUNCH(ctx, "gate_coeff_setup", gate_coeff_setup<U>, dim3((T + 63) / 64), dim3(64), 0, dd_coeff, dd_lift, T, d_mont);
                   ZK_LAUNCH(ctx, "gate_eval", gate_eval<U>, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, dd_ptr, dd_gate, dd_term, dd_fac, d_mont, G,
                             (uint32_t)log_size, (const uint32_t *)d_mask, accumulate, (uint32_t *)d_out));
    return ZKHIP_OK;
}
