//---------------------------------------------------------------------------//
// zkhip shim: the Lagrange-basis transform of a powers-of-tau accumulator, on the MI355X.
//
// Mirrors evaluation_domain<Fr, G>::evaluate_all_lagrange_polynomials(powers_begin, powers_end) as
// powers_of_tau_result::from_accumulator calls it for coeffs_g1 / coeffs_g2 / alpha_coeffs_g1 / beta_coeffs_g1
// (zk/commitments/detail/polynomial/powers_of_tau/result.hpp:81-94): given P_i = tau^i G for i < m (m a power of two)
// it returns L_j(tau) G for every Lagrange polynomial of the m-point domain -- an inverse DFT over group elements.
// `omega` is the primitive m-th root of unity of that domain (math::make_evaluation_domain's choice).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_POWERS_OF_TAU_HPP
#define ZKHIP_SHIM_POWERS_OF_TAU_HPP

#include <iterator>
#include <vector>

#include "multiexp.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

template <typename CurveType, int Group, typename InputIt>
std::vector<typename detail::jac_result<CurveType, Group>::type>
    evaluate_all_lagrange_polynomials(const context &ctx, InputIt powers_begin, InputIt powers_end,
                                      const typename curve_adapter<CurveType>::scalar_value_type &omega) {
    typedef curve_adapter<CurveType> adapter;
    typedef detail::jac_result<CurveType, Group> R;
    const std::size_t m = std::distance(powers_begin, powers_end), cl = R::limbs / 3;
    std::size_t log_m = 0;
    while (((std::size_t)1 << log_m) < m) ++log_m;
    if (m == 0 || ((std::size_t)1 << log_m) != m) throw std::runtime_error("evaluate_all_lagrange_polynomials: the domain size must be a power of two");
    /* canonical Jacobian (x, y, 1) of every power; (0, 0, 0) for the point at infinity */
    std::vector<std::uint64_t> jac(m * R::limbs, 0);
    std::size_t i = 0;
    for (InputIt it = powers_begin; it != powers_end; ++it, ++i)
        if (adapter::point_to_affine_limbs(*it, &jac[i * R::limbs])) jac[i * R::limbs + 2 * cl] = 1;
    auto d = ctx.alloc(jac.size() * 8);
    ctx.h2d(d.get(), jac.data(), jac.size() * 8);
    std::uint64_t w[4];
    adapter::scalar_to_limbs(omega, w);
    check(zkhip_ec_ntt_dev(ctx.get(), adapter::id, Group, d.get(), log_m, w, 1), "zkhip_ec_ntt_dev", ctx.get());
    ctx.d2h(jac.data(), d.get(), jac.size() * 8);
    std::vector<typename R::type> out;
    for (i = 0; i < m; ++i) out.push_back(R::make(&jac[i * R::limbs]));
    return out;
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_POWERS_OF_TAU_HPP
