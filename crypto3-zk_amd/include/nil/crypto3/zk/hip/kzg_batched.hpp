// zkhip shim, part 3c: the free-function form of the batched KZG scheme.
//
// Mirrors commitments::batched_kzg and its algorithms (zk/commitments/polynomial/kzg.hpp:223-320, 322-630) over polynomials in
// COEFFICIENT form (math::polynomial: here std::vector<scalar>), the shape the reference's own batched_kzg_basic_test drives
// (test/commitment/kzg.cpp:535-572):
//     merge_eval_points (:480-493)   create_evals_polys (:374-392)   commit (:440-455)   update_transcript (:323-372)
//     proof_eval (:557-600): ONE commitment of  sum_i gamma^i (f_i - r_i) / Z_{S_i}
// proof_eval runs on the device: every f_i goes up once, r_i is subtracted from its low coefficients, one synthetic division per
// point of S_i (zkhip_poly_div_linear_dev, the remainder checked), the gamma-weighted sum (zkhip_poly_lincomb_dev) and one multiexp
// against the resident commitment key.  verify_eval (:604-628) is two pairings over commit_g2 (kzg.hpp of this directory): the
// caller's, as everywhere in this shim.  The scheme CLASS over polynomial_dfs batches is kzg_commitment_scheme_hip (kzg_v2.hpp).
#ifndef ZKHIP_SHIM_KZG_BATCHED_HPP
#define ZKHIP_SHIM_KZG_BATCHED_HPP

#include <algorithm>
#include <stdexcept>
#include <vector>

#include <nil/crypto3/zk/hip/kzg_v2.hpp>

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// batched_kzg::public_key_type (kzg.hpp:300-318)
template <typename CurveType>
struct kzg_batched_public_key_hip {
    typedef typename curve_adapter<CurveType>::scalar_value_type scalar_value_type;
    typedef typename curve_adapter<CurveType>::g1_value_type single_commitment_type;
    std::vector<single_commitment_type> commits;
    std::vector<scalar_value_type> T;                 // merged eval points
    std::vector<std::vector<scalar_value_type>> S;    // eval points
    std::vector<std::vector<scalar_value_type>> r;    // U polynomials (coefficients)
};

/// merge_eval_points (kzg.hpp:480-493): the union, in the order of a std::set of field values
template <typename CurveType>
std::vector<typename curve_adapter<CurveType>::scalar_value_type>
kzg_batched_merge_eval_points(const std::vector<std::vector<typename curve_adapter<CurveType>::scalar_value_type>> &S) {
    typedef typename curve_adapter<CurveType>::scalar_value_type Fr;
    std::vector<Fr> out;
    for (const auto &s : S)
        for (const auto &x : s)
            if (std::find(out.begin(), out.end(), x) == out.end()) out.push_back(x);
    std::sort(out.begin(), out.end(), detail::limbs_less<Fr>);
    return out;
}

/// create_evals_polys (kzg.hpp:374-392): r_i interpolates f_i on S_i
template <typename CurveType>
std::vector<std::vector<typename curve_adapter<CurveType>::scalar_value_type>>
kzg_batched_create_evals_polys(const std::vector<std::vector<typename curve_adapter<CurveType>::scalar_value_type>> &polys,
                               const std::vector<std::vector<typename curve_adapter<CurveType>::scalar_value_type>> &S) {
    typedef typename curve_adapter<CurveType>::scalar_value_type Fr;
    typedef detail::small_poly<Fr> SP;
    if (polys.size() != S.size()) throw std::invalid_argument("create_evals_polys: one point set per polynomial");
    std::vector<std::vector<Fr>> rs(polys.size());
    for (std::size_t i = 0; i < polys.size(); ++i) {
        std::vector<Fr> ys;
        for (const auto &s : S[i]) ys.push_back(SP::evaluate(polys[i], s));
        rs[i] = SP::lagrange(S[i], ys);
    }
    return rs;
}

/// commit (kzg.hpp:440-455): one commitment per polynomial, as one device batch
template <typename CurveType>
std::vector<typename curve_adapter<CurveType>::g1_value_type>
kzg_batched_commit(const kzg_params_hip<CurveType> &params, const std::vector<std::vector<typename curve_adapter<CurveType>::scalar_value_type>> &polys) {
    std::vector<typename curve_adapter<CurveType>::g1_value_type> out;
    for (const auto &p : polys) out.push_back(commit_one<CurveType>(params, p));
    return out;
}

/// update_transcript (kzg.hpp:323-372): commitments, then every point of every S, then every coefficient of every r
template <typename CurveType, typename TranscriptType>
void kzg_batched_update_transcript(const kzg_batched_public_key_hip<CurveType> &public_key, TranscriptType &transcript) {
    for (const auto &c : public_key.commits) transcript(c);
    for (const auto &S : public_key.S)
        for (const auto &s : S) transcript(s);
    for (const auto &r : public_key.r)
        for (const auto &c : r) transcript(c);
}

/// proof_eval (kzg.hpp:557-600)
template <typename CurveType, typename TranscriptType>
typename curve_adapter<CurveType>::g1_value_type
kzg_batched_proof_eval(const kzg_params_hip<CurveType> &params, const std::vector<std::vector<typename curve_adapter<CurveType>::scalar_value_type>> &polys,
                       const kzg_batched_public_key_hip<CurveType> &public_key, TranscriptType &transcript) {
    typedef curve_adapter<CurveType> adapter;
    typedef typename adapter::scalar_value_type Fr;
    const context &ctx = params.ctx;
    if (public_key.S.size() != polys.size() || public_key.r.size() != polys.size())
        throw std::invalid_argument("proof_eval: the public key does not describe this batch");
    kzg_batched_update_transcript<CurveType>(public_key, transcript);
    const Fr gamma = transcript.challenge();

    std::size_t acc_len = 0;
    /* the reference asserts (f_i - r_i) % Z_{S_i} == 0 for EVERY polynomial (kzg.hpp:569-580); a public key whose r_i is not the
       interpolation of f_i over S_i must not yield a proof here either.  f - r spans max(|f|, |r|) coefficients (a longer r_i makes
       a non-vanishing remainder, caught by the divisions below); a short polynomial (deg f < |S|) passes only with f == r. */
    auto coeff = [](const std::vector<Fr> &v, std::size_t j) { return j < v.size() ? v[j] : Fr::zero(); };
    for (std::size_t i = 0; i < polys.size(); ++i) {
        const std::size_t span = std::max(polys[i].size(), public_key.r[i].size());
        std::size_t deg_p1 = 0;    // 1 + degree of f - r
        for (std::size_t j = span; j-- > 0;)
            if (!(coeff(polys[i], j) == coeff(public_key.r[i], j))) {
                deg_p1 = j + 1;
                break;
            }
        if (deg_p1 != 0 && deg_p1 <= public_key.S[i].size()) throw std::runtime_error("proof_eval: (f - r) does not vanish on S");
    }
    for (std::size_t i = 0; i < polys.size(); ++i) {
        const std::size_t span = std::max(polys[i].size(), public_key.r[i].size());
        if (span > public_key.S[i].size()) acc_len = std::max(acc_len, span - public_key.S[i].size());
    }
    if (acc_len == 0) return adapter::g1_value_type::zero();    // every f_i = r_i: all quotients vanish
    if (acc_len > params.commitment_key.size()) throw std::runtime_error("proof_eval: quotient longer than the commitment key");
    auto d_acc = ctx.alloc(acc_len * 32);
    Fr factor = Fr::one();
    bool first = true;
    for (std::size_t i = 0; i < polys.size(); ++i, factor = factor * gamma) {
        const auto &f = polys[i];
        const std::size_t span = std::max(f.size(), public_key.r[i].size());
        if (span <= public_key.S[i].size()) continue;    // deg (f - r) < |S| and f == r (checked above): the quotient is zero
        /* spare_poly = f - r (:571), over the longer of the two */
        std::vector<Fr> spare(span, Fr::zero());
        for (std::size_t j = 0; j < span; ++j) spare[j] = coeff(f, j) - coeff(public_key.r[i], j);
        auto d_q = ctx.alloc(spare.size() * 32);
        upload_scalars<adapter>(ctx, d_q.get(), spare.data(), spare.size());
        /* spare_poly /= create_polynom_by_zeros(S_i) (:572-580): one root at a time, every remainder must vanish */
        char *q_ptr = static_cast<char *>(d_q.get());
        std::size_t q_len = spare.size();
        for (const auto &s : public_key.S[i]) {
            std::uint64_t zl[4], rem[4];
            adapter::scalar_to_limbs(s, zl);
            check(zkhip_poly_div_linear_dev(ctx.get(), adapter::id, q_ptr, q_len, zl, q_ptr, rem), "zkhip_poly_div_linear_dev", ctx.get());
            if (rem[0] | rem[1] | rem[2] | rem[3]) throw std::runtime_error("proof_eval: (f - r) does not vanish on S");
            q_ptr += 32;
            q_len -= 1;
        }
        /* accum += spare_poly * factor (:581) */
        std::uint64_t fl[4];
        adapter::scalar_to_limbs(factor, fl);
        const void *qp = q_ptr;
        check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, 1, &qp, &q_len, fl, 1, d_acc.get(), acc_len, first ? 0 : 1), "zkhip_poly_lincomb_dev", ctx.get());
        first = false;
        ctx.sync();    // d_q is released at the end of the iteration
    }
    if (first) return adapter::g1_value_type::zero();
    /* commit_one(params, accum) (:599) */
    return multiexp_dev<CurveType, ZKHIP_G1>(ctx, params.commitment_key, 0, acc_len, d_acc.get());
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_KZG_BATCHED_HPP
