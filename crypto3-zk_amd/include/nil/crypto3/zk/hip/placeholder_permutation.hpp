//---------------------------------------------------------------------------//
// zkhip shim: placeholder's permutation argument, prover side, on the device --
//   placeholder_permutation_argument::prove_eval   zk/snark/systems/plonk/placeholder/permutation_argument.hpp:70-224
// both forms: permutation_parts == 1 (common_data.max_quotient_chunks == 0) and the multi-part one the reference's tests also run
// (test/systems/plonk/placeholder/placeholder.cpp:1245, 1287-1291: max_quotient_poly_chunks = 8, 10, 30, 50).
//   :103-124  g_v[i] = column_i + beta S_id[i] + gamma,  h_v[i] = column_i + beta S_sigma[i] + gamma
//   :126-136  V_P[0] = 1, V_P[j] = V_P[j - 1] prod_i g_v[i][j - 1] / prod_i h_v[i][j - 1]    -- zkhip_perm_grand_product_dev: a serial loop with
//             one inversion per row in the reference; chunks sharing an inversion + a three-level prefix-product scan here
//   :140-160  g = polynomial_product(g_v), h = polynomial_product(h_v)
//   :163-218  F_dfs[0] = lagrange_0 (1 - V_P),  F_dfs[1] = (1 - (q_last + q_blind)) (V_P(omega X) h - V_P g),  F_dfs[2] = q_last V_P (V_P - 1)
//   :147-160, 188-207  max_quotient_chunks = c != 0: the factors in groups of c - 1; every group but the last gives an intermediate polynomial
//             current[j] = previous[j] g_i[j] / h_i[j] over the usable rows (zkhip_fr_vec_mul_div_dev: chunks sharing an inversion; one inversion per
//             row in the reference), appended to PERMUTATION_BATCH after V_P, and
//             F_dfs[1] = ((q_last + q_blind) - 1)(sum_i alpha_i (previous_i g_i - current_i h_i) + previous_last g_last - V_P(omega X) h_last)
// What the caller keeps: the transcript (beta, gamma are arguments) and `commitment_scheme.append_to_batch(PERMUTATION_BATCH, V_P)` --
// the returned V_P is a device_polynomial_dfs, which the KZG scheme takes where it lies.
// Domain sizes of the three F polynomials follow their degrees (the smallest power of two that holds them), as polynomial_dfs's
// operator*= does; quotient_polynomial (placeholder_quotient.hpp) brings all parts to one domain anyway.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_PLACEHOLDER_PERMUTATION_HPP
#define ZKHIP_SHIM_PLACEHOLDER_PERMUTATION_HPP

#include <array>
#include <functional>
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
        std::vector<dfs_type> parts_dfs;        // the intermediate polynomials of the multi-part form, in PERMUTATION_BATCH order behind V_P
    };

    /// what the reference does BETWEEN the numeric steps (placeholder_arguments.hpp drives them from the transcript and the commitment scheme):
    /// on_V_P: V_P is final (:139 append_to_batch); draw_alphas(permutation_parts): the parts' challenges (:181-183); on_part: an intermediate
    /// polynomial is final (:200 append_to_batch).  Any of them may be empty.
    struct hooks_type {
        std::function<void(const dfs_type &)> on_V_P, on_part;
        std::function<std::vector<value_type>(std::size_t parts)> draw_alphas;
    };

    /// `columns[i]`: the i-th permuted column (column_polynomials[global_indices[i]], :93-97), S_id / S_sigma the preprocessed identity /
    /// permutation polynomials, all over the n-row basic domain.  max_quotient_chunks / alphas / usable_rows: the multi-part form
    /// (alphas: permutation_parts - 1 challenges in drawing order, :181-183).
    static prover_result_type prove_eval(const context &ctx, const std::vector<dfs_type> &columns, const std::vector<dfs_type> &S_id,
                                         const std::vector<dfs_type> &S_sigma, const dfs_type &q_last, const dfs_type &q_blind, const dfs_type &lagrange_0,
                                         const value_type &beta, const value_type &gamma, const root_of_unity_type &root, std::size_t max_quotient_chunks = 0,
                                         const std::vector<value_type> &alphas = {}, std::size_t usable_rows = 0) {
        hooks_type hooks;
        hooks.draw_alphas = [&alphas](std::size_t parts) {
            if (alphas.size() + 1 != parts) throw std::invalid_argument("permutation argument: permutation_parts - 1 alphas");
            return alphas;
        };
        return prove_eval_hooked(ctx, columns, S_id, S_sigma, q_last, q_blind, lagrange_0, beta, gamma, root, max_quotient_chunks, usable_rows, hooks);
    }

    /// the same with the challenges of the parts and the batch appends handed to the caller's hooks, each at the point of the reference's flow
    static prover_result_type prove_eval_hooked(const context &ctx, const std::vector<dfs_type> &columns, const std::vector<dfs_type> &S_id,
                                                const std::vector<dfs_type> &S_sigma, const dfs_type &q_last, const dfs_type &q_blind, const dfs_type &lagrange_0,
                                                const value_type &beta, const value_type &gamma, const root_of_unity_type &root, std::size_t max_quotient_chunks,
                                                std::size_t usable_rows, const hooks_type &hooks) {
        const std::size_t k = columns.size();
        if (k == 0 || S_id.size() != k || S_sigma.size() != k) throw std::invalid_argument("permutation argument: one S_id / S_sigma per permuted column");
        const std::size_t n = columns[0].size();
        for (std::size_t i = 0; i < k; ++i)
            if (columns[i].size() != n || S_id[i].size() != n || S_sigma[i].size() != n) throw std::invalid_argument("permutation argument: sizes differ from the basic domain's");
        if (q_last.size() != n || q_blind.size() != n || lagrange_0.size() != n) throw std::invalid_argument("permutation argument: selector sizes differ from the basic domain's");
        if (max_quotient_chunks == 1) throw std::invalid_argument("permutation argument: max_quotient_chunks = 1 leaves no factor per part");
        const std::size_t step = max_quotient_chunks ? max_quotient_chunks - 1 : k, parts = (k + step - 1) / step;    // preprocessor.hpp:80-87
        if (parts > 1 && (usable_rows == 0 || usable_rows >= n)) throw std::invalid_argument("permutation argument: the multi-part form needs usable_rows");
        /* Round 5: with extension caches on S_id / S_sigma (preprocessed, the same in every proof) the factors g_v, h_v are formed ON the products'
           domain from ONE extension per column -- ext(column) + beta ext(S) + gamma, a pointwise pass -- instead of being extended one by one (2 k
           transforms); a column's extension is shared with whatever else extended it (the gate argument) when the column's cache is on as well. */
        bool factors_from_columns = true;
        for (std::size_t i = 0; i < k; ++i) factors_from_columns = factors_from_columns && S_id[i].extension_cache_enabled() && S_sigma[i].extension_cache_enabled();
        /* 2.-3.: g_v, h_v and V_P in one device call */
        std::shared_ptr<void> d_g, d_h;
        if (!factors_from_columns) {
            d_g = ctx.alloc(k * n * 32);
            d_h = ctx.alloc(k * n * 32);
        }
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
        V_P.set_degree(n - 1);
        if (hooks.on_V_P) hooks.on_V_P(V_P);    // 4. (:139)
        /* 5.: gs[p] = prod of the p-th group's g_v, hs[p] likewise */
        std::vector<dfs_type> gs, hs;
        for (std::size_t lo = 0; lo < k; lo += step) {
            std::vector<dfs_type> g_v, h_v;
            const std::size_t hi = std::min(k, lo + step), size = pow2_holding((hi - lo) * (n - 1));
            if (factors_from_columns) {
                /* the group's g and h on their domain in ONE pass over the extended columns and (cached) permutation polynomials */
                std::vector<dfs_type> ext;    // alive until the pass ran
                std::vector<const void *> ec, ei, es;
                for (std::size_t i = lo; i < hi; ++i) {
                    ext.push_back(extended(columns[i], size, root));
                    ec.push_back(ext.back().data());
                    ext.push_back(extended(S_id[i], size, root));
                    ei.push_back(ext.back().data());
                    ext.push_back(extended(S_sigma[i], size, root));
                    es.push_back(ext.back().data());
                }
                dfs_type g(ctx, size), h(ctx, size);
                g.set_degree((hi - lo) * (n - 1));
                h.set_degree((hi - lo) * (n - 1));
                check(zkhip_perm_factor_products_dev(ctx.get(), adapter::id, hi - lo, ec.data(), ei.data(), es.data(), size, bl, gl, g.data(), h.data()),
                      "zkhip_perm_factor_products_dev", ctx.get());
                gs.push_back(std::move(g));
                hs.push_back(std::move(h));
                continue;
            }
            for (std::size_t i = lo; i < hi; ++i) {
                g_v.emplace_back(ctx, n);
                h_v.emplace_back(ctx, n);
                check(zkhip_memcpy_d2d_async(ctx.get(), g_v.back().data(), static_cast<const char *>(d_g.get()) + i * n * 32, n * 32), "zkhip_memcpy_d2d_async", ctx.get());
                check(zkhip_memcpy_d2d_async(ctx.get(), h_v.back().data(), static_cast<const char *>(d_h.get()) + i * n * 32, n * 32), "zkhip_memcpy_d2d_async", ctx.get());
            }
            gs.push_back(polynomial_product<CurveType>(std::move(g_v), root));
            hs.push_back(polynomial_product<CurveType>(std::move(h_v), root));
        }
        /* V_P takes part in five products on three domains: it is extended ONCE, to the largest of them (that of V_P g), and subsampled for the
           others inside polynomial_product; V_P(omega X) is a rotation of that extension; likewise lagrange_0 and q_last */
        std::size_t deg_g = 0;
        for (const auto &g : gs) deg_g = std::max(deg_g, g.degree());
        const dfs_type V = extended(V_P, pow2_holding(V_P.degree() + deg_g), root);
        const dfs_type V_shifted = polynomial_shift(V, 1, n);
        const dfs_type L0 = extended(lagrange_0, 2 * n, root), QL = extended(q_last, 4 * n, root);
        prover_result_type res {{dfs_type(ctx, 1), dfs_type(ctx, 1), dfs_type(ctx, 1)}, V_P, {}};
        /* F_dfs[0] = lagrange_0 (1 - V_P) = lagrange_0 - lagrange_0 V_P */
        res.F_dfs[0] = minus(L0, polynomial_product<CurveType>({L0, V}, root, L0.size()), root);
        const std::vector<value_type> alphas = hooks.draw_alphas ? hooks.draw_alphas(parts) : std::vector<value_type>();    // :181-183
        if (alphas.size() + 1 != parts) throw std::invalid_argument("permutation argument: permutation_parts - 1 alphas");
        const dfs_type q = selector_sum(q_last, q_blind, pow2_holding(n - 1 + V_P.degree() + deg_g), root);    // q_last + q_blind
        if (parts == 1) {
            /* F_dfs[1] = (1 - q)(V_P_shifted h - V_P g) = T - q T,  q = q_last + q_blind */
            dfs_type T = minus(polynomial_product<CurveType>({V_shifted, hs[0]}, root), polynomial_product<CurveType>({V, gs[0]}, root), root);
            res.F_dfs[1] = minus(T, polynomial_product<CurveType>({q, T}, root), root);
        } else {
            /* F_dfs[1] = (q - 1)(sum_i alpha_i (previous g_i - current h_i) + previous g_last - V_P_shifted h_last) = q S - S */
            dfs_type previous = V_P, previous_ext = V, S(ctx, 1);
            for (std::size_t p = 0; p + 1 < parts; ++p) {
                dfs_type current = multiplied_up(previous, V_P, gs[p], hs[p], n, usable_rows);
                res.parts_dfs.push_back(current);
                if (hooks.on_part) hooks.on_part(current);    // :200
                const dfs_type current_ext = extended(current, V.size(), root);    // once for current h_p and, as the next previous, for previous g_(p + 1)
                dfs_type part = minus(polynomial_product<CurveType>({previous_ext, gs[p]}, root), polynomial_product<CurveType>({current_ext, hs[p]}, root), root);
                scale(part, alphas[p]);
                S = p == 0 ? part : plus(S, part, root);
                previous = current;
                previous_ext = current_ext;
            }
            dfs_type last = minus(polynomial_product<CurveType>({previous_ext, gs[parts - 1]}, root), polynomial_product<CurveType>({V_shifted, hs[parts - 1]}, root), root);
            S = plus(S, last, root);
            res.F_dfs[1] = minus(polynomial_product<CurveType>({q, S}, root), S, root);
        }
        /* F_dfs[2] = q_last V_P (V_P - 1) = q_last V_P V_P - q_last V_P, both products on the first one's domain */
        res.F_dfs[2] = minus(polynomial_product<CurveType>({QL, V, V}, root), polynomial_product<CurveType>({QL, V}, root, QL.size()), root);
        ctx.sync();
        return res;
    }

    /// q_last + q_blind: on the n-point domain, or -- when both hold extension caches (preprocessed selectors) -- directly on the `size`-point one
    /// their sum is multiplied on afterwards: two cached extensions and one addition instead of a transform
    static dfs_type selector_sum(const dfs_type &q_last, const dfs_type &q_blind, std::size_t size, const root_of_unity_type &root) {
        const bool cached = q_last.extension_cache_enabled() && q_blind.extension_cache_enabled();
        const dfs_type a = cached ? extended(q_last, size, root) : q_last, b = cached ? extended(q_blind, size, root) : q_blind;
        dfs_type q(a.ctx(), a.size());
        q.set_degree(std::max(q_last.degree(), q_blind.degree()));
        check(zkhip_fr_vec_op_dev(a.ctx().get(), adapter::id, 0, a.data(), b.data(), q.data(), a.size()), "zkhip_fr_vec_op_dev", a.ctx().get());
        return q;
    }
    /// a x + b y + c over the shared domain, into a buffer of its own
    static dfs_type affine(const dfs_type &x, const dfs_type &y, const value_type &a, const value_type &b, const value_type &c) {
        if (y.size() != x.size()) throw std::invalid_argument("permutation argument: operands must share the domain");
        dfs_type out(x.ctx(), x.size());
        std::uint64_t al[4], bl[4], cl[4];
        adapter::scalar_to_limbs(a, al);
        adapter::scalar_to_limbs(b, bl);
        adapter::scalar_to_limbs(c, cl);
        check(zkhip_fr_vec_affine_dev(x.ctx().get(), adapter::id, x.data(), y.data(), al, bl, cl, out.data(), x.size()), "zkhip_fr_vec_affine_dev", x.ctx().get());
        return out;
    }
    /// the intermediate polynomial of a part (:193-199): current[j] = previous[j] reduced_g[j] / reduced_h[j] for j < usable_rows, `fill`'s values
    /// behind them (V_P's, which `current_poly` starts as); reduced = every (size / n)-th evaluation (reduce_dfs_polynomial_domain)
    static dfs_type multiplied_up(const dfs_type &previous, const dfs_type &fill, const dfs_type &g, const dfs_type &h, std::size_t n, std::size_t usable_rows) {
        const context &ctx = previous.ctx();
        dfs_type current(ctx, n);
        current.set_degree(n - 1);
        check(zkhip_memcpy_d2d_async(ctx.get(), current.data(), fill.data(), n * 32), "zkhip_memcpy_d2d_async", ctx.get());
        dfs_type rg = reduced(g, n), rh = reduced(h, n);
        check(zkhip_fr_vec_mul_div_dev(ctx.get(), adapter::id, previous.data(), rg.data(), rh.data(), current.data(), usable_rows), "zkhip_fr_vec_mul_div_dev", ctx.get());
        ctx.sync();    // rg, rh are released on return
        return current;
    }
    /// p on the `size`-point domain (a copy with a buffer of its own; p itself where it already lives there)
    static dfs_type extended(const dfs_type &p, std::size_t size, const root_of_unity_type &root) { return p.extension(size, root); }
    static std::size_t pow2_holding(std::size_t degree) {
        std::size_t size = 1;
        while (size < degree + 1) size <<= 1;
        return size;
    }
    static dfs_type reduced(const dfs_type &p, std::size_t n) {
        if (p.size() == n) return p;
        if (p.size() < n || p.size() % n) throw std::invalid_argument("permutation argument: not an extension of the basic domain");
        dfs_type out(p.ctx(), n);
        std::size_t lp = 0, ln = 0;
        while (((std::size_t)1 << lp) < p.size()) ++lp;
        while (((std::size_t)1 << ln) < n) ++ln;
        std::uint64_t unused[4] = {1, 0, 0, 0};
        check(zkhip_poly_resize_dev(p.ctx().get(), adapter::id, p.data(), lp, 1, unused, out.data(), ln, unused), "zkhip_poly_resize_dev", p.ctx().get());
        return out;
    }
    /// a + b on the larger of the two domains, into a buffer of its own
    static dfs_type plus(dfs_type a, dfs_type b, const root_of_unity_type &root) { return combine(0, a, b, root); }
    /// a - b likewise (copies of a device_polynomial_dfs share their buffer: never in place)
    static dfs_type minus(dfs_type a, dfs_type b, const root_of_unity_type &root) { return combine(1, a, b, root); }
    /// p *= c, in place (a view of a cached extension gets a buffer of its own first; p's own cached extensions go)
    static void scale(dfs_type &p, const value_type &c) {
        p.make_writable();
        std::uint64_t cl[4], zl[4];
        adapter::scalar_to_limbs(c, cl);
        adapter::scalar_to_limbs(value_type::zero(), zl);
        check(zkhip_fr_vec_affine_dev(p.ctx().get(), adapter::id, p.data(), nullptr, cl, nullptr, zl, p.data(), p.size()), "zkhip_fr_vec_affine_dev", p.ctx().get());
    }

private:
    static dfs_type combine(int op, dfs_type a, dfs_type b, const root_of_unity_type &root) {
        const std::size_t size = std::max(a.size(), b.size());
        a.resize(size, root);
        b.resize(size, root);
        dfs_type out(a.ctx(), size);
        out.set_degree(std::max(a.degree(), b.degree()));
        check(zkhip_fr_vec_op_dev(a.ctx().get(), adapter::id, op, a.data(), b.data(), out.data(), size), "zkhip_fr_vec_op_dev", a.ctx().get());
        a.ctx().sync();    // a and b are released on return
        return out;
    }
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_PLACEHOLDER_PERMUTATION_HPP
