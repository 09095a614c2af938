//---------------------------------------------------------------------------//
// zkhip shim: placeholder's lookup argument, prover side, from the sorted vectors on --
//   placeholder_lookup_argument_prover::prove_eval   zk/snark/systems/plonk/placeholder/lookup_argument.hpp:153-296
// both forms: one part (lookup_parts(max_quotient_chunks = 0) = { sorted_lookup_columns_number }, :56-62) and the multi-part one the reference's
// tests also run (test/systems/plonk/placeholder/placeholder.cpp:1287-1291: max_quotient_poly_chunks = 8, 10, 30, 50); the caller passes
// lookup_parts(...)'s result, which is the constraint system's to compute (:63-107).
//   :175-186  reduce_dfs_polynomial_domain: every (size / n)-th evaluation (:498-517)                 -- reduce_dfs_polynomial_domain below
//   :208-209  compute_V_L (:375-409): V_L[0] = 1, V_L[k] = V_L[k - 1] g(k - 1) / h(k - 1), k <= usable_rows, zero behind
//             -- zkhip_lookup_grand_product_dev: a serial loop with one inversion per row in the reference; the permutation argument's scan here
//   :223-229  compute_gs / compute_hs (:297-373): g = prod_i (1 + beta)(gamma + input_i) prod_i ((1 + beta) gamma + value_i + beta value_i(omega X)),
//             h = prod_i ((1 + beta) gamma + sorted_i + beta sorted_i(omega X))    -- one zkhip_fr_vec_affine_dev pass per factor + polynomial_product
//   :236-251  F_dfs[0] = lagrange_0 (1 - V_L),  F_dfs[1] = q_last (V_L V_L - V_L),  F_dfs[2] = ((q_last + q_blind) - 1)(V_L g - V_L(omega X) h)
//   :252-276  several parts: the g / h factors in groups of part_sizes[i]; every group but the last gives an intermediate polynomial
//             current[j] = previous[j] g_i[j] / h_i[j] over the usable rows (zkhip_fr_vec_mul_div_dev), appended to PERMUTATION_BATCH after V_L, and
//             F_dfs[2] = ((q_last + q_blind) - 1)(sum_i alpha_i (previous_i g_i - current_i h_i) + previous_last g_last - V_L(omega X) h_last)
//   :278-288  F_dfs[3] = sum_i alpha_i lagrange_0 (sorted_(i + 1) - sorted_i(omega^usable_rows X))
//   :187-189  sort_polynomials (:565-638, an unordered_map count + one serial walk over the rows in the reference) -- sort_polynomials below
//             (zkhip_lookup_sort_dev: run detection + scans, a hash table over the run heads, one atomic per input, a binary search per entry)
// What the caller keeps: the constraint system's side -- prepare_lookup_value / prepare_lookup_input (:411-496, a walk over the lookup tables and
// gates with theta); device_polynomial_dfs go to the KZG scheme where they lie.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_PLACEHOLDER_LOOKUP_HPP
#define ZKHIP_SHIM_PLACEHOLDER_LOOKUP_HPP

#include <array>
#include <functional>
#include <stdexcept>
#include <vector>

#include "placeholder_permutation.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

template <typename CurveType>
struct placeholder_lookup_hip {
    typedef curve_adapter<CurveType> adapter;
    typedef typename adapter::scalar_value_type value_type;
    typedef device_polynomial_dfs<CurveType> dfs_type;
    typedef typename dfs_type::root_of_unity_type root_of_unity_type;

    struct prover_result_type {
        std::array<dfs_type, 4> F_dfs;
        dfs_type V_L;
        std::vector<dfs_type> parts_dfs;    // the intermediate polynomials of the multi-part form, in PERMUTATION_BATCH order behind V_L
    };

    /// lookup_argument.hpp:498-517 -- unlike resize() this does not look at the degree: the result is the vector of values on the smaller domain
    static dfs_type reduce_dfs_polynomial_domain(const dfs_type &p, std::size_t new_domain_size) {
        if (new_domain_size > p.size() || p.size() % new_domain_size) throw std::invalid_argument("reduce_dfs_polynomial_domain: not a sub-domain");
        if (p.size() == new_domain_size) return p;
        const context &ctx = p.ctx();
        dfs_type out(ctx, new_domain_size);
        std::uint64_t unused[4] = {1, 0, 0, 0};
        check(zkhip_poly_resize_dev(ctx.get(), adapter::id, p.data(), log2_of(p.size()), 1, unused, out.data(), log2_of(new_domain_size), unused), "zkhip_poly_resize_dev",
              ctx.get());
        return out;
    }

    /// sort_polynomials (:565-638) over the REDUCED vectors, on the device (zkhip_lookup_sort_dev): |input| + |value| vectors of domain_size
    /// entries that never leave HBM on their way to commit(LOOKUP_BATCH) and compute_V_L.  The reference counts in an unordered_map and
    /// emits in one serial walk over the rows.  A looked-up value that is in no table (the reference's BOOST_ASSERT, :583) or a table whose
    /// equal values are not adjacent and overflow the vectors raise bits 2 / 3 of the sticky device status (context::device_status()).
    static std::vector<dfs_type> sort_polynomials(const context &ctx, const std::vector<dfs_type> &reduced_input, const std::vector<dfs_type> &reduced_value,
                                                  std::size_t domain_size, std::size_t usable_rows) {
        std::vector<const void *> pi, pv;
        for (const auto &x : reduced_input) {
            if (x.size() != domain_size) throw std::invalid_argument("sort_polynomials: a reduced input's size differs from the basic domain's");
            pi.push_back(x.data());
        }
        for (const auto &x : reduced_value) {
            if (x.size() != domain_size) throw std::invalid_argument("sort_polynomials: a reduced value's size differs from the basic domain's");
            pv.push_back(x.data());
        }
        std::vector<dfs_type> sorted;
        std::vector<void *> ps;
        for (std::size_t i = 0; i < pi.size() + pv.size(); ++i) {
            sorted.emplace_back(ctx, domain_size);
            ps.push_back(sorted.back().data());
        }
        check(zkhip_lookup_sort_dev(ctx.get(), pi.size(), pi.data(), pv.size(), pv.data(), domain_size, usable_rows, ps.data()), "zkhip_lookup_sort_dev", ctx.get());
        return sorted;
    }

    /// compute_V_L (:375-409) over the REDUCED vectors
    static dfs_type compute_V_L(const context &ctx, const std::vector<dfs_type> &sorted, const std::vector<dfs_type> &reduced_input,
                                const std::vector<dfs_type> &reduced_value, const value_type &beta, const value_type &gamma, std::size_t usable_rows) {
        if (sorted.empty()) throw std::invalid_argument("lookup argument: no sorted vectors");
        const std::size_t n = sorted[0].size();
        auto ptrs = [n](const std::vector<dfs_type> &v) {
            std::vector<const void *> p;
            for (const auto &x : v) {
                if (x.size() != n) throw std::invalid_argument("lookup argument: a reduced vector's size differs from the basic domain's");
                p.push_back(x.data());
            }
            return p;
        };
        const auto pi = ptrs(reduced_input), pv = ptrs(reduced_value), ps = ptrs(sorted);
        std::uint64_t bl[4], gl[4];
        adapter::scalar_to_limbs(beta, bl);
        adapter::scalar_to_limbs(gamma, gl);
        dfs_type V_L(ctx, n);
        check(zkhip_lookup_grand_product_dev(ctx.get(), adapter::id, pi.size(), pi.data(), pv.size(), pv.data(), ps.size(), ps.data(), n, usable_rows, bl, gl, V_L.data()),
              "zkhip_lookup_grand_product_dev", ctx.get());
        return V_L;
    }

    /// `lookup_input` / `lookup_value`: what prepare_lookup_input / prepare_lookup_value return (inputs may live on larger domains than the
    /// n-row basic one), `sorted`: sort_polynomials' result, `alphas`: sorted.size() - 1 challenges in drawing order (F_dfs[3], :281-283).
    /// part_sizes (empty: one part) / part_alphas (part_sizes.size() - 1 challenges, drawn BEFORE V_L is computed, :202-206): the multi-part form.
    static prover_result_type prove_eval(const context &ctx, const std::vector<dfs_type> &lookup_input, const std::vector<dfs_type> &lookup_value,
                                         const std::vector<dfs_type> &sorted, const dfs_type &q_last, const dfs_type &q_blind, const dfs_type &lagrange_0,
                                         const value_type &beta, const value_type &gamma, const std::vector<value_type> &alphas, std::size_t usable_rows,
                                         const root_of_unity_type &root, std::vector<std::size_t> part_sizes = {}, const std::vector<value_type> &part_alphas = {}) {
        if (alphas.size() + 1 != sorted.size()) throw std::invalid_argument("lookup argument: one alpha per sorted vector but the first");
        hooks_type hooks;
        std::size_t next = 0;
        hooks.draw_alpha = [&alphas, &next]() { return alphas.at(next++); };
        return prove_eval_hooked(ctx, lookup_input, lookup_value, sorted, q_last, q_blind, lagrange_0, beta, gamma, usable_rows, root, std::move(part_sizes), part_alphas, hooks);
    }

    /// what the reference does BETWEEN the numeric steps from `sorted` on (placeholder_arguments.hpp drives them from the transcript and the
    /// commitment scheme): on_V_L: V_L is final (:213 append_to_batch(PERMUTATION_BATCH)); on_part: an intermediate polynomial is final (:267);
    /// draw_alpha: the next challenge of F_dfs[3]'s sum, drawn inside its loop (:282-283).  on_V_L / on_part may be empty.
    struct hooks_type {
        std::function<void(const dfs_type &)> on_V_L, on_part;
        std::function<value_type()> draw_alpha;
    };

    static prover_result_type prove_eval_hooked(const context &ctx, const std::vector<dfs_type> &lookup_input, const std::vector<dfs_type> &lookup_value,
                                                const std::vector<dfs_type> &sorted, const dfs_type &q_last, const dfs_type &q_blind, const dfs_type &lagrange_0,
                                                const value_type &beta, const value_type &gamma, std::size_t usable_rows, const root_of_unity_type &root,
                                                std::vector<std::size_t> part_sizes, const std::vector<value_type> &part_alphas, const hooks_type &hooks) {
        typedef placeholder_permutation_hip<CurveType> PA;    // the shared helpers: plus / minus / scale / multiplied_up
        if (part_sizes.empty()) part_sizes.push_back(sorted.size());
        std::size_t covered = 0;
        for (std::size_t sz : part_sizes) {
            if (sz == 0) throw std::invalid_argument("lookup argument: an empty part");
            covered += sz;
        }
        if (covered != sorted.size() || part_alphas.size() + 1 != part_sizes.size())
            throw std::invalid_argument("lookup argument: part_sizes must add up to the sorted vectors, with one alpha per part but the last");
        if (sorted.empty() || sorted.size() != lookup_input.size() + lookup_value.size())
            throw std::invalid_argument("lookup argument: one sorted vector per input and value vector");
        if (sorted.size() > 1 && !hooks.draw_alpha) throw std::invalid_argument("lookup argument: no source for F_dfs[3]'s alphas");
        const std::size_t n = sorted[0].size();
        if (q_last.size() != n || q_blind.size() != n || lagrange_0.size() != n) throw std::invalid_argument("lookup argument: selector sizes differ from the basic domain's");
        if (usable_rows >= n) throw std::invalid_argument("lookup argument: usable_rows must be below the domain size");
        const value_type one = value_type::one(), zero = value_type::zero(), part1 = (one + beta) * gamma;
        /* 3., 5.: reduce, then V_L in one device call */
        std::vector<dfs_type> reduced_input, reduced_value;
        for (const auto &p : lookup_input) reduced_input.push_back(reduce_dfs_polynomial_domain(p, n));
        for (const auto &p : lookup_value) reduced_value.push_back(reduce_dfs_polynomial_domain(p, n));
        dfs_type V_L = compute_V_L(ctx, sorted, reduced_input, reduced_value, beta, gamma, usable_rows);
        V_L.set_degree(n - 1);
        if (hooks.on_V_L) hooks.on_V_L(V_L);    // :213
        reduced_input.clear();
        reduced_value.clear();
        /* compute_gs / compute_hs */
        std::vector<dfs_type> g_multipliers, h_multipliers;
        for (const auto &p : lookup_input) g_multipliers.push_back(affine(p, nullptr, one + beta, zero, part1));
        for (const auto &p : lookup_value) {
            dfs_type shifted = polynomial_shift(p, 1, n);
            g_multipliers.push_back(affine(p, &shifted, one, beta, part1));
        }
        for (const auto &p : sorted) {
            dfs_type shifted = polynomial_shift(p, 1, n);
            h_multipliers.push_back(affine(p, &shifted, one, beta, part1));
        }
        std::vector<dfs_type> gs, hs;
        for (std::size_t p = 0, at = 0; p < part_sizes.size(); at += part_sizes[p++]) {
            gs.push_back(polynomial_product<CurveType>(std::vector<dfs_type>(g_multipliers.begin() + at, g_multipliers.begin() + at + part_sizes[p]), root));
            hs.push_back(polynomial_product<CurveType>(std::vector<dfs_type>(h_multipliers.begin() + at, h_multipliers.begin() + at + part_sizes[p]), root));
        }
        g_multipliers.clear();
        h_multipliers.clear();
        /* V_L takes part in five products on three domains: extended ONCE, to the largest of them (that of V_L g), subsampled for the others
           inside polynomial_product; V_L(omega X) is a rotation of that extension; likewise lagrange_0 and q_last */
        std::size_t deg_g = 0;
        for (const auto &g : gs) deg_g = std::max(deg_g, g.degree());
        const dfs_type V = PA::extended(V_L, PA::pow2_holding(V_L.degree() + deg_g), root);
        const dfs_type V_shifted = polynomial_shift(V, 1, n);
        const dfs_type L0 = PA::extended(lagrange_0, 2 * n, root), QL = PA::extended(q_last, 4 * n, root);
        prover_result_type res {{dfs_type(ctx, 1), dfs_type(ctx, 1), dfs_type(ctx, 1), dfs_type(ctx, 1)}, V_L, {}};
        /* F_dfs[0] = lagrange_0 (1 - V_L) = lagrange_0 - lagrange_0 V_L */
        res.F_dfs[0] = minus(L0, polynomial_product<CurveType>({L0, V}, root, L0.size()), root);
        /* F_dfs[1] = q_last (V_L V_L - V_L), both products on the first one's domain */
        res.F_dfs[1] = minus(polynomial_product<CurveType>({QL, V, V}, root), polynomial_product<CurveType>({QL, V}, root, QL.size()), root);
        /* F_dfs[2] = ((q_last + q_blind) - 1) T = q T - T,  T = V_L g - V_L_shifted h (one part),
           T = sum_i alpha_i (previous g_i - current h_i) + previous g_last - V_L_shifted h_last (several) */
        dfs_type T(ctx, 1), previous = V_L, previous_ext = V;
        const std::size_t parts = part_sizes.size();
        for (std::size_t p = 0; p + 1 < parts; ++p) {
            dfs_type current = PA::multiplied_up(previous, V_L, gs[p], hs[p], n, usable_rows);
            res.parts_dfs.push_back(current);
            if (hooks.on_part) hooks.on_part(current);    // :267
            const dfs_type current_ext = PA::extended(current, V.size(), root);    // once for current h_p and, as the next previous, for previous g_(p + 1)
            dfs_type part = minus(polynomial_product<CurveType>({previous_ext, gs[p]}, root), polynomial_product<CurveType>({current_ext, hs[p]}, root), root);
            PA::scale(part, part_alphas[p]);
            T = p == 0 ? part : PA::plus(T, part, root);
            previous = current;
            previous_ext = current_ext;
        }
        {
            dfs_type last = minus(polynomial_product<CurveType>({previous_ext, gs[parts - 1]}, root), polynomial_product<CurveType>({V_shifted, hs[parts - 1]}, root), root);
            T = parts == 1 ? last : PA::plus(T, last, root);
        }
        const dfs_type q = PA::selector_sum(q_last, q_blind, PA::pow2_holding(std::max(q_last.degree(), q_blind.degree()) + T.degree()), root);
        res.F_dfs[2] = minus(polynomial_product<CurveType>({q, T}, root), T, root);
        /* F_dfs[3] = sum_i alpha_i lagrange_0 (sorted[i + 1] - sorted[i](omega^usable_rows X)) */
        if (sorted.size() > 1) {
            dfs_type sum(ctx, n);
            for (std::size_t i = 0; i + 1 < sorted.size(); ++i) {
                const value_type alpha = hooks.draw_alpha();    // :282
                dfs_type shifted = polynomial_shift(sorted[i], (int)usable_rows, n);
                dfs_type part = affine(sorted[i + 1], &shifted, alpha, zero - alpha, zero);
                if (i == 0)
                    sum = part;
                else
                    sum += part;
            }
            sum.set_degree(n - 1);
            res.F_dfs[3] = polynomial_product<CurveType>({sum, L0}, root);
        } else {
            res.F_dfs[3] = affine(lagrange_0, nullptr, zero, zero, zero);    // zero_polynomial
            res.F_dfs[3].set_degree(0);
        }
        ctx.sync();
        return res;
    }

    /// a x + b y + c over x's domain, into a buffer of its own
    static dfs_type affine(const dfs_type &x, const dfs_type *y, const value_type &a, const value_type &b, const value_type &c) {
        if (y && y->size() != x.size()) throw std::invalid_argument("lookup argument: operands must share the domain");
        dfs_type out(x.ctx(), x.size());
        out.set_degree(y ? std::max(x.degree(), y->degree()) : x.degree());
        std::uint64_t al[4], bl[4], cl[4];
        adapter::scalar_to_limbs(a, al);
        adapter::scalar_to_limbs(b, bl);
        adapter::scalar_to_limbs(c, cl);
        check(zkhip_fr_vec_affine_dev(x.ctx().get(), adapter::id, x.data(), y ? y->data() : nullptr, al, y ? bl : nullptr, cl, out.data(), x.size()),
              "zkhip_fr_vec_affine_dev", x.ctx().get());
        return out;
    }

private:
    static std::size_t log2_of(std::size_t n) {
        std::size_t l = 0;
        while (((std::size_t)1 << l) < n) ++l;
        if (n == 0 || ((std::size_t)1 << l) != n) throw std::invalid_argument("lookup argument: sizes must be powers of two");
        return l;
    }
    static dfs_type minus(const dfs_type &a, const dfs_type &b, const root_of_unity_type &root) { return placeholder_permutation_hip<CurveType>::minus(a, b, root); }
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_PLACEHOLDER_LOOKUP_HPP
