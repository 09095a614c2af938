//---------------------------------------------------------------------------//
// zkhip shim: KZG commitment of polynomials given in evaluation (DFS) form, on the MI355X.
//
// Mirrors zk/commitments/polynomial/kzg.hpp:
//   params_type { commitment_key = { alpha^i * G1 } }                      (:262-290)
//   commit_one<KZG>(params, polynomial_dfs)  = multiexp(ck, p.coefficients()) (:427-435: iNTT then MSM, chunks = 1)
//   commit_one<KZG>(params, polynomial)      = multiexp(ck, coefficients)        (:409-420)
//   commit_g2<KZG>(params, polynomial)       = multiexp(verification_key, coefficients) in G2 (:497-510, 659-664)
//   kzg_commitment_scheme[_v2]::commit(batch) loops commit_one over the batch (:748-765, kzg_v2.hpp:208-226)
// Here the whole batch is transformed by ONE batched inverse NTT and the coefficient vectors never leave HBM:
// each column's MSM reads them in place against the resident SRS.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_KZG_HPP
#define ZKHIP_SHIM_KZG_HPP

#include <memory>
#include <vector>

#include "multiexp.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// math::polynomial_dfs as far as the commitment layer reads it: evaluations over the size()-point domain.  This is only the
/// DEFAULT polynomial type of the schemes: they are templated on `PolynomialType` exactly as the reference's polys_evaluator is
/// (batched_commitment.hpp:56-64) and read a polynomial through size() and operator[] alone (detail::poly_data below), so
/// crypto3-math's own math::polynomial_dfs -- private storage, begin / end / size / operator[] -- is consumed where it lies.
template <typename CurveType>
struct polynomial_dfs {
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    std::vector<value_type> values;
    std::size_t size() const { return values.size(); }
    const value_type &operator[](std::size_t i) const { return values[i]; }
    value_type &operator[](std::size_t i) { return values[i]; }
    typename std::vector<value_type>::const_iterator begin() const { return values.begin(); }
    typename std::vector<value_type>::const_iterator end() const { return values.end(); }
};

namespace detail {
    /// the contiguous evaluations of a polynomial_dfs-like object (what the uploads read): its element type must be the
    /// adapter's scalar type, its storage contiguous (std::vector inside math::polynomial_dfs)
    template <typename Adapter, typename PolynomialType>
    const typename Adapter::scalar_value_type *poly_data(const PolynomialType &p) {
        static_assert(std::is_same<typename std::decay<decltype(p[0])>::type, typename Adapter::scalar_value_type>::value,
                      "PolynomialType's elements must be the curve adapter's scalar_value_type");
        return p.size() ? &p[0] : nullptr;
    }
}    // namespace detail

/// kzg::params_type with the commitment key resident on the device.
template <typename CurveType>
struct kzg_params_hip {
    typedef multiexp_method_hip multiexp_method;    // shadows kzg.hpp:82's BDLO12 typedef
    template <typename InputIt>
    kzg_params_hip(const context &ctx, InputIt ck_first, InputIt ck_last) : ctx(ctx), commitment_key(ctx, ck_first, ck_last) { }
    /// a commitment key that is already resident (device_bases::from_scalars / from_compressed, the powers-of-tau result)
    kzg_params_hip(const context &ctx, device_bases<CurveType, ZKHIP_G1> &&key) : ctx(ctx), commitment_key(std::move(key)) { }
    /// batched_kzg::params_type (kzg.hpp:236-280): commitment_key = { alpha^i G1 }, verification_key = { alpha^i G2 }, i <= t
    template <typename InputIt, typename VkIt>
    kzg_params_hip(const context &ctx, InputIt ck_first, InputIt ck_last, VkIt vk_first, VkIt vk_last) :
        ctx(ctx), commitment_key(ctx, ck_first, ck_last), verification_key(ctx, vk_first, vk_last) { }
    kzg_params_hip(const context &ctx, device_bases<CurveType, ZKHIP_G1> &&key, device_bases<CurveType, ZKHIP_G2> &&vkey) :
        ctx(ctx), commitment_key(std::move(key)), verification_key(std::move(vkey)) { }
    /// params_type(d, t, alpha) (kzg.hpp:262-275, what the reference's tests construct): the powers alpha^i G1, i < d, and alpha^i G2,
    /// i <= t, computed on the device (fixed-base batch exponentiation) on the calling thread's default context
    kzg_params_hip(std::size_t d, std::size_t t, const typename curve_adapter<CurveType>::scalar_value_type &alpha) :
        ctx(default_context()), commitment_key(powers_of<ZKHIP_G1>(default_context(), d, alpha)),
        verification_key(powers_of<ZKHIP_G2>(default_context(), t + 1, alpha)) { }
    const context &ctx;
    device_bases<CurveType, ZKHIP_G1> commitment_key;
    device_bases<CurveType, ZKHIP_G2> verification_key;    // empty unless given: only commit_g2 reads it

private:
    template <int Group>
    static device_bases<CurveType, Group> powers_of(const context &c, std::size_t count, const typename curve_adapter<CurveType>::scalar_value_type &alpha) {
        typedef typename curve_adapter<CurveType>::scalar_value_type Fr;
        std::vector<Fr> p(count, Fr::one());
        for (std::size_t i = 1; i < count; ++i) p[i] = p[i - 1] * alpha;
        return device_bases<CurveType, Group>::from_scalars(c, p.begin(), p.end());
    }
};

/// kzg::params_type over a device group (backend.hpp): the commitment key REPLICATED on every member -- SURVEY 8e: "each GPU then MSMs its
/// own columns against a replicated SRS and only 96-byte commitments are gathered".  A scheme constructed from it
/// (kzg_commitment_scheme[_v2]_hip, kzg_v2.hpp) deals the columns of commit(batch) over the members; everything else runs on member 0
/// (`root()`), where the coefficient forms are collected for proof_eval.
template <typename CurveType>
struct kzg_params_group_hip {
    typedef multiexp_method_hip multiexp_method;
    /// the key from a range of group values: converted and uploaded once (member 0), replicated device to device from there
    template <typename InputIt>
    kzg_params_group_hip(const device_group &group, InputIt ck_first, InputIt ck_last) : group(group) {
        members.emplace_back(new kzg_params_hip<CurveType>(group[0], ck_first, ck_last));
        replicate();
    }
    /// params_type(d, t, alpha) (kzg.hpp:262-275) with the powers computed on member 0
    kzg_params_group_hip(const device_group &group, std::size_t d, const typename curve_adapter<CurveType>::scalar_value_type &alpha) : group(group) {
        typedef typename curve_adapter<CurveType>::scalar_value_type Fr;
        std::vector<Fr> p(d, Fr::one());
        for (std::size_t i = 1; i < d; ++i) p[i] = p[i - 1] * alpha;
        members.emplace_back(new kzg_params_hip<CurveType>(group[0], device_bases<CurveType, ZKHIP_G1>::from_scalars(group[0], p.begin(), p.end())));
        replicate();
    }
    const kzg_params_hip<CurveType> &root() const { return *members.front(); }
    const device_group &group;
    std::vector<std::unique_ptr<kzg_params_hip<CurveType>>> members;    // members[k] lives on group[k]

private:
    void replicate() {
        for (std::size_t k = 1; k < group.size(); ++k)
            members.emplace_back(new kzg_params_hip<CurveType>(group[k], members[0]->commitment_key.replicate(group[k])));
    }
};

/// commit(batch): one commitment per polynomial; all polynomials must have the same power-of-two size
/// n <= commitment_key.size().  `omega` is the primitive n-th root of the polynomials' evaluation domain.
template <typename CurveType, typename PolynomialType = polynomial_dfs<CurveType>>
std::vector<typename curve_adapter<CurveType>::g1_value_type>
    kzg_commit_batch(const kzg_params_hip<CurveType> &params, const std::vector<PolynomialType> &polys,
                     const typename curve_adapter<CurveType>::scalar_value_type &omega) {
    typedef curve_adapter<CurveType> adapter;
    std::vector<typename adapter::g1_value_type> out;
    if (polys.empty()) return out;
    const context &ctx = params.ctx;
    const std::size_t n = polys[0].size(), batch = polys.size();
    std::size_t log_n = 0;
    while (((std::size_t)1 << log_n) < n) ++log_n;
    if (((std::size_t)1 << log_n) != n || n > params.commitment_key.size()) throw std::runtime_error("kzg_commit_batch: bad polynomial size");
    for (std::size_t b = 0; b < batch; ++b)
        if (polys[b].size() != n) throw std::runtime_error("kzg_commit_batch: ragged batch");
    auto d = ctx.alloc(n * batch * 32);
    for (std::size_t b = 0; b < batch; ++b) upload_scalars<adapter>(ctx, static_cast<char *>(d.get()) + 32 * b * n, detail::poly_data<adapter>(polys[b]), n);
    std::uint64_t w[4];
    adapter::scalar_to_limbs(omega, w);
    /* p.coefficients() for every polynomial of the batch (kzg.hpp:431) */
    check(zkhip_ntt_dev(ctx.get(), adapter::id, d.get(), log_n, batch, w, 1, nullptr), "zkhip_ntt_dev", ctx.get());
    /* multiexp<multiexp_method>(commitment_key[0 .. n), coefficients, 1) for every column (kzg.hpp:433-434), as one
       batch: the columns' bucket reductions share one launch */
    const std::size_t jl = 3 * adapter::g1_coord_limbs;
    auto d_res = ctx.alloc(batch * jl * 8);
    std::vector<const zkhip_bases *> qb(batch, params.commitment_key.get());
    std::vector<std::size_t> qo(batch, 0), qn(batch, n);
    std::vector<const void *> qs(batch);
    std::vector<void *> qr(batch);
    for (std::size_t b = 0; b < batch; ++b) {
        qs[b] = static_cast<const char *>(d.get()) + 32 * b * n;
        qr[b] = static_cast<std::uint64_t *>(d_res.get()) + b * jl;
    }
    check(zkhip_msm_batch_dev(ctx.get(), batch, qb.data(), qo.data(), qn.data(), qs.data(), qr.data()), "zkhip_msm_batch_dev", ctx.get());
    std::vector<std::uint64_t> res(batch * jl);
    ctx.d2h(res.data(), d_res.get(), res.size() * 8);
    for (std::size_t b = 0; b < batch; ++b) out.push_back(adapter::g1_from_jacobian(&res[b * jl]));
    return out;
}

/// commit_one<KZG>(params, polynomial_dfs)
template <typename CurveType, typename PolynomialType = polynomial_dfs<CurveType>,
          typename std::enable_if<!std::is_same<PolynomialType, std::vector<typename curve_adapter<CurveType>::scalar_value_type>>::value, bool>::type = true>
typename curve_adapter<CurveType>::g1_value_type commit_one(const kzg_params_hip<CurveType> &params, const PolynomialType &poly,
                                                            const typename curve_adapter<CurveType>::scalar_value_type &omega) {
    return kzg_commit_batch<CurveType, PolynomialType>(params, std::vector<PolynomialType> {poly}, omega)[0];
}

/// commit_one<KZG>(params, math::polynomial) (kzg.hpp:409-420): the polynomial given by its coefficients
template <typename CurveType>
typename curve_adapter<CurveType>::g1_value_type commit_one(const kzg_params_hip<CurveType> &params,
                                                            const std::vector<typename curve_adapter<CurveType>::scalar_value_type> &poly) {
    if (poly.size() > params.commitment_key.size()) throw std::runtime_error("commit_one: polynomial longer than the commitment key");
    return multiexp<CurveType, ZKHIP_G1>(params.ctx, params.commitment_key, 0, poly.begin(), poly.end(), 1);
}

/// commit_g2<KZG>(params, poly) (kzg.hpp:497-510; kzg_commitment_scheme::commit_g2 :659-664): sum_i poly[i] * verification_key[i]
template <typename CurveType>
typename curve_adapter<CurveType>::g2_value_type commit_g2(const kzg_params_hip<CurveType> &params,
                                                           const std::vector<typename curve_adapter<CurveType>::scalar_value_type> &poly) {
    if (poly.size() > params.verification_key.size()) throw std::runtime_error("commit_g2: polynomial longer than the verification key");
    if (poly.empty()) return curve_adapter<CurveType>::g2_value_type::zero();
    return multiexp<CurveType, ZKHIP_G2>(params.ctx, params.verification_key, 0, poly.begin(), poly.end(), 1);
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_KZG_HPP
