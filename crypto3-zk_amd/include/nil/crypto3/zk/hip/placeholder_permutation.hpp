//---------------------------------------------------------------------------//
// zkhip shim: placeholder's permutation argument, prover side, on the device --
//   placeholder_permutation_argument::prove_eval   zk/snark/systems/plonk/placeholder/permutation_argument.hpp:70-224
// (the permutation_parts == 1 form: common_data.max_quotient_chunks == 0, as the reference's tests configure it).
//   :103-124  g_v[i] = column_i + beta S_id[i] + gamma,  h_v[i] = column_i + beta S_sigma[i] + gamma
//   :126-136  V_P[0] = 1, V_P[j] = V_P[j - 1] prod_i g_v[i][j - 1] / prod_i h_v[i][j - 1]    -- zkhip_perm_grand_product_dev: a serial loop with
//             one inversion per row in the reference; chunks sharing an inversion + a three-level prefix-product scan here
//   :140-160  g = polynomial_product(g_v), h = polynomial_product(h_v)
//   :163-218  F_dfs[0] = lagrange_0 (1 - V_P),  F_dfs[1] = (1 - (q_last + q_blind)) (V_P(omega X) h - V_P g),  F_dfs[2] = q_last V_P (V_P - 1)
// What the caller keeps: the transcript (beta, gamma are arguments) and `commitment_scheme.append_to_batch(PERMUTATION_BATCH, V_P)` --
// the returned V_P is a device_polynomial_dfs, which the KZG scheme takes where it lies.
// Domain sizes of the three F polynomials follow their degrees (the smallest power of two that holds them), as polynomial_dfs's
// operator*= does; quotient_polynomial (placeholder_quotient.hpp) brings all parts to one domain anyway.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_PLACEHOLDER_PERMUTATION_HPP
#define ZKHIP_SHIM_PLACEHOLDER_PERMUTATION_HPP

#include <array>
#include <stdexcept>
#include <vector>

#include "fri.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

template <typename CurveType>
struct placeholder_permutation_hip {
    typedef curve_adapter<CurveType> adapter;
    typedef typename adapter::scalar_value_type value_type;
    typedef device_polynomial_dfs<CurveType> dfs_type;
    typedef typename dfs_type::root_of_unity_type root_of_unity_type;

    struct prover_result_type {
        std::array<dfs_type, 3> F_dfs;
        dfs_type permutation_polynomial_dfs;    // V_P
    };

    /// `columns[i]`: the i-th permuted column (column_polynomials[global_indices[i]], :93-97), S_id / S_sigma the preprocessed identity /
    /// permutation polynomials, all over the n-row basic domain.
    static prover_result_type prove_eval(const context &ctx, const std::vector<dfs_type> &columns, const std::vector<dfs_type> &S_id,
                                         const std::vector<dfs_type> &S_sigma, const dfs_type &q_last, const dfs_type &q_blind, const dfs_type &lagrange_0,
                                         const value_type &beta, const value_type &gamma, const root_of_unity_type &root) {
        const std::size_t k = columns.size();
        if (k == 0 || S_id.size() != k || S_sigma.size() != k) throw std::invalid_argument("permutation argument: one S_id / S_sigma per permuted column");
        const std::size_t n = columns[0].size();
        for (std::size_t i = 0; i < k; ++i)
            if (columns[i].size() != n || S_id[i].size() != n || S_sigma[i].size() != n) throw std::invalid_argument("permutation argument: sizes differ from the basic domain's");
        if (q_last.size() != n || q_blind.size() != n || lagrange_0.size() != n) throw std::invalid_argument("permutation argument: selector sizes differ from the basic domain's");
        /* 2.-3.: g_v, h_v and V_P in one device call */
        auto d_g = ctx.alloc(k * n * 32), d_h = ctx.alloc(k * n * 32);
        dfs_type V_P(ctx, n);
        std::vector<const void *> pc, pi, ps;
        for (std::size_t i = 0; i < k; ++i) {
            pc.push_back(columns[i].data());
            pi.push_back(S_id[i].data());
            ps.push_back(S_sigma[i].data());
        }
        std::uint64_t bl[4], gl[4];
        adapter::scalar_to_limbs(beta, bl);
        adapter::scalar_to_limbs(gamma, gl);
        check(zkhip_perm_grand_product_dev(ctx.get(), adapter::id, k, pc.data(), pi.data(), ps.data(), n, bl, gl, d_g.get(), d_h.get(), V_P.data()),
              "zkhip_perm_grand_product_dev", ctx.get());
        /* 5.: g = prod g_v[i], h = prod h_v[i] */
        std::vector<dfs_type> g_v, h_v;
        for (std::size_t i = 0; i < k; ++i) {
            g_v.emplace_back(ctx, n);
            h_v.emplace_back(ctx, n);
            check(zkhip_memcpy_d2d_async(ctx.get(), g_v.back().data(), static_cast<const char *>(d_g.get()) + i * n * 32, n * 32), "zkhip_memcpy_d2d_async", ctx.get());
            check(zkhip_memcpy_d2d_async(ctx.get(), h_v.back().data(), static_cast<const char *>(d_h.get()) + i * n * 32, n * 32), "zkhip_memcpy_d2d_async", ctx.get());
        }
        dfs_type g = polynomial_product<CurveType>(g_v, root), h = polynomial_product<CurveType>(h_v, root);
        dfs_type V_P_shifted = polynomial_shift(V_P, 1, n);
        prover_result_type res {{dfs_type(ctx, 1), dfs_type(ctx, 1), dfs_type(ctx, 1)}, V_P};
        /* F_dfs[0] = lagrange_0 (1 - V_P) = lagrange_0 - lagrange_0 V_P */
        res.F_dfs[0] = minus(lagrange_0, polynomial_product<CurveType>({lagrange_0, V_P}, root), root);
        /* F_dfs[1] = (1 - q)(V_P_shifted h - V_P g) = T - q T,  q = q_last + q_blind */
        dfs_type T = minus(polynomial_product<CurveType>({V_P_shifted, h}, root), polynomial_product<CurveType>({V_P, g}, root), root);
        dfs_type q(ctx, n);    // q_last + q_blind, in a buffer of its own
        q.set_degree(std::max(q_last.degree(), q_blind.degree()));
        check(zkhip_fr_vec_op_dev(ctx.get(), adapter::id, 0, q_last.data(), q_blind.data(), q.data(), n), "zkhip_fr_vec_op_dev", ctx.get());
        res.F_dfs[1] = minus(T, polynomial_product<CurveType>({q, T}, root), root);
        /* F_dfs[2] = q_last V_P (V_P - 1) = q_last V_P V_P - q_last V_P */
        res.F_dfs[2] = minus(polynomial_product<CurveType>({q_last, V_P, V_P}, root), polynomial_product<CurveType>({q_last, V_P}, root), root);
        ctx.sync();
        return res;
    }

private:
    /// a - b on the larger of the two domains, into a buffer of its own (copies of a device_polynomial_dfs share their buffer)
    static dfs_type minus(dfs_type a, dfs_type b, const root_of_unity_type &root) {
        const std::size_t size = std::max(a.size(), b.size());
        a.resize(size, root);
        b.resize(size, root);
        dfs_type out(a.ctx(), size);
        out.set_degree(std::max(a.degree(), b.degree()));
        check(zkhip_fr_vec_op_dev(a.ctx().get(), adapter::id, 1, a.data(), b.data(), out.data(), size), "zkhip_fr_vec_op_dev", a.ctx().get());
        a.ctx().sync();    // a and b are released on return
        return out;
    }
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_PLACEHOLDER_PERMUTATION_HPP
