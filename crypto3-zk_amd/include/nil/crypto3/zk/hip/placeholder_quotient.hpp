//---------------------------------------------------------------------------//
// zkhip shim: placeholder's quotient-polynomial chain on the device -- the NTT consumers SURVEY section 2 stars next to the
// commitment schemes (#10-12):
//   placeholder_prover::quotient_polynomial            zk/snark/systems/plonk/placeholder/prover.hpp:262-277
//   placeholder_prover::quotient_polynomial_split_dfs  prover.hpp:220-259   (detail::split_polynomial, prover.hpp:55-69)
//   placeholder_prover::T_commit                       prover.hpp:314-317
//   the polynomial_dfs side of the gates argument      gates_argument.hpp:93-121, 203-216 (polynomial_shift, resize to the extended
//        domain, pointwise products and sums, `F[0] *= mask_polynomial`)
// What stays the reference's: the symbolic expression machinery that decides WHICH products a circuit's gates are
// (math::expression, the visitors; SURVEY section 2 out of scope) -- here the caller names the factors.
//
// Everything below works on device_polynomial_dfs (fri.hpp): the columns stay resident from the table to the commitment
// (kzg_polys_evaluator_hip::append_to_batch(index, device_polynomial_dfs) takes them where they lie).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_PLACEHOLDER_QUOTIENT_HPP
#define ZKHIP_SHIM_PLACEHOLDER_QUOTIENT_HPP

#include <algorithm>
#include <map>
#include <stdexcept>
#include <vector>

#include "fri.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// One gate's contribution as gates_argument.hpp:203-216 evaluates it once the expression is a product: selector x factors, every
/// factor first moved by its rotation (math::polynomial_shift over the ORIGINAL domain, gates_argument.hpp:108-110) and resized to
/// the extended domain (`assignment.resize(extended_domain_size, ...)`, :111-113), times theta^k.
template <typename CurveType>
struct gate_product_hip {
    std::vector<const device_polynomial_dfs<CurveType> *> factors;    // selector first, then the columns
    std::vector<int> rotations;                                       // one per factor (0: none)
    typename curve_adapter<CurveType>::scalar_value_type coefficient; // theta_acc of the constraint
};

template <typename CurveType>
struct placeholder_quotient_hip {
    typedef curve_adapter<CurveType> adapter;
    typedef typename adapter::scalar_value_type value_type;
    typedef device_polynomial_dfs<CurveType> dfs_type;
    typedef typename dfs_type::root_of_unity_type root_of_unity_type;

    /// A coefficient vector resident on the device (math::polynomial as far as this chain needs it)
    struct device_coefficients {
        std::shared_ptr<void> data;
        std::size_t size = 0;    // coefficients (the top ones may be zero: nothing here condenses)
        const void *at(std::size_t i) const { return static_cast<const char *>(data.get()) + 32 * i; }
    };

    /// gates_argument.hpp:203-216: F = (sum over the gates' products) * mask_polynomial on the extended domain of `extended_size`
    /// points (original_domain->m * 2^ceil(log2(max_gates_degree + 1)), :149-150) -- ONE launch over a flat program (zkhip_gate_eval_dev,
    /// round 6).  Every DISTINCT column the products name is extended once, UNROTATED (from its extension cache when the holder switched it
    /// on, else by one batched call); a rotation is index arithmetic on the extended domain inside the kernel (rotation * extended_size / n
    /// rows), so a column used with three rotations costs one extension and no math::polynomial_shift copy; consecutive products that share
    /// their first factor and its rotation (the selector, by gate_product_hip's convention) become one gate whose selector is multiplied
    /// once; selector, coefficients and the mask are applied in the same pass over the rows.  Round 5's two launches per product are kept
    /// below as gate_argument_per_term (A / B measurements, and a second implementation the tests hold this one against).  When the
    /// extensions of all distinct columns pass `slot_budget` bytes the products are evaluated in groups that accumulate into F.
    static dfs_type gate_argument(const context &ctx, const std::vector<gate_product_hip<CurveType>> &products, const dfs_type &mask_polynomial,
                                  std::size_t extended_size, const root_of_unity_type &root, std::size_t slot_budget = (std::size_t)16 << 30) {
        return sum_of_products(ctx, products, &mask_polynomial, extended_size, root, slot_budget);
    }

    /// The evaluator under gate_argument, usable on its own: sum_p coefficient_p prod_f factor_f(omega^rotation_f X) on the `extended_size`-point
    /// domain, times `mask_polynomial` when one is given.  Without a mask it is the numeric side of prepare_lookup_input
    /// (lookup_argument.hpp:435-496: selector * (table_id + sum_k theta^(k+1) expression_k), see prepare_lookup_input_flat in
    /// placeholder_lookup.hpp) -- any sum of monomials over resident columns.
    static dfs_type sum_of_products(const context &ctx, const std::vector<gate_product_hip<CurveType>> &products, const dfs_type *mask_ptr, std::size_t extended_size,
                                    const root_of_unity_type &root, std::size_t slot_budget = (std::size_t)16 << 30) {
        if (products.empty()) throw std::invalid_argument("gate_argument: no products");
        if (products[0].factors.empty()) throw std::invalid_argument("gate_argument: factors / rotations");
        /* without a mask the stand-in below is never read: only its size (the original domain's) and its degree (0) are */
        const dfs_type &mask_polynomial = mask_ptr ? *mask_ptr : *products[0].factors[0];
        const std::size_t mask_degree = mask_ptr ? mask_ptr->degree() : 0;
        const std::size_t n = mask_polynomial.size();
        std::size_t log_n = 0, log_e = 0;
        while (((std::size_t)1 << log_n) < n) ++log_n;
        while (((std::size_t)1 << log_e) < extended_size) ++log_e;
        if (((std::size_t)1 << log_n) != n || ((std::size_t)1 << log_e) != extended_size || extended_size < n)
            throw std::invalid_argument("gate_argument: domain sizes must be powers of two, extended >= original");
        const std::int64_t stretch = (std::int64_t)(extended_size / n);    // one row of the original domain = this many of the extended one
        std::uint64_t wn[4], we[4];
        adapter::scalar_to_limbs(root(log_n), wn);
        adapter::scalar_to_limbs(root(log_e), we);
        dfs_type F(ctx, extended_size);
        std::size_t F_degree = 0;
        const std::size_t max_slots = std::max<std::size_t>(2, slot_budget / (extended_size * 32));
        for (std::size_t lo = 0; lo < products.size();) {
            /* the group [lo, hi): as many products as fit `max_slots` distinct columns (the last group also holds the mask) */
            std::vector<const dfs_type *> unique;
            auto slot_of = [&unique](const dfs_type *f) { return (std::size_t)(std::find(unique.begin(), unique.end(), f) - unique.begin()); };
            std::size_t hi = lo;
            for (; hi < products.size(); ++hi) {
                const auto &g = products[hi];
                if (g.factors.empty() || g.rotations.size() != g.factors.size()) throw std::invalid_argument("gate_argument: factors / rotations");
                std::vector<const dfs_type *> added;
                for (const dfs_type *f : g.factors)
                    if (slot_of(f) == unique.size() && std::find(added.begin(), added.end(), f) == added.end()) added.push_back(f);
                if (hi > lo && unique.size() + added.size() + 1 > max_slots) break;
                unique.insert(unique.end(), added.begin(), added.end());
            }
            const bool last = hi == products.size();
            const std::size_t U = unique.size();
            /* where column u (and, at U, the mask -- needed by the last group only) lives on the extended domain */
            std::vector<const void *> ext_of(U + 1, nullptr);
            std::vector<dfs_type> keep;
            std::vector<std::size_t> batched;
            for (std::size_t u = 0; u <= U; ++u) {
                if (u == U && (!last || !mask_ptr)) break;
                const dfs_type &f = u < U ? *unique[u] : mask_polynomial;
                if (f.size() != n) throw std::invalid_argument("gate_argument: every factor lives on the original domain (the mask's size)");
                if (extended_size == n) ext_of[u] = f.data();
                else if (f.extension_cache_enabled()) {
                    keep.push_back(f.extension(extended_size, root));
                    ext_of[u] = keep.back().data();
                } else batched.push_back(u);
            }
            const std::size_t slots = batched.size();
            std::shared_ptr<void> d_in, d_ext;
            if (slots) {
                d_in = ctx.alloc(slots * n * 32);
                d_ext = ctx.alloc(slots * extended_size * 32);
                for (std::size_t b = 0; b < slots; ++b) {
                    const std::size_t u = batched[b];
                    const dfs_type &f = u < U ? *unique[u] : mask_polynomial;
                    check(zkhip_memcpy_d2d_async(ctx.get(), static_cast<char *>(d_in.get()) + b * n * 32, f.data(), n * 32), "zkhip_memcpy_d2d_async", ctx.get());
                    ext_of[u] = static_cast<const char *>(d_ext.get()) + b * extended_size * 32;
                }
                check(zkhip_poly_resize_dev(ctx.get(), adapter::id, d_in.get(), log_n, slots, wn, d_ext.get(), log_e, we), "zkhip_poly_resize_dev", ctx.get());
            }
            /* the flat program of the group */
            std::vector<std::uint32_t> gate_terms {0}, gate_sel, term_factors {0}, factor_slot;
            std::vector<std::int32_t> gate_sel_rot, factor_rot;
            std::vector<std::uint64_t> term_coeff;
            for (std::size_t gi = lo; gi < hi; ++gi) {
                const auto &g = products[gi];
                std::size_t degree = 0;
                for (const dfs_type *f : g.factors) degree += f->degree();
                if (degree >= extended_size) throw std::invalid_argument("gate_argument: the product's degree does not fit the extended domain");
                F_degree = std::max(F_degree, degree);
                const bool same_selector = gi > lo && products[gi - 1].factors[0] == g.factors[0] && products[gi - 1].rotations[0] == g.rotations[0];
                if (!same_selector) {
                    if (gi > lo) gate_terms.push_back((std::uint32_t)(term_factors.size() - 1));
                    gate_sel.push_back((std::uint32_t)slot_of(g.factors[0]));
                    gate_sel_rot.push_back((std::int32_t)(g.rotations[0] * stretch));
                }
                for (std::size_t k = 1; k < g.factors.size(); ++k) {
                    factor_slot.push_back((std::uint32_t)slot_of(g.factors[k]));
                    factor_rot.push_back((std::int32_t)(g.rotations[k] * stretch));
                }
                term_factors.push_back((std::uint32_t)factor_slot.size());
                term_coeff.resize(term_coeff.size() + 4);
                adapter::scalar_to_limbs(g.coefficient, term_coeff.data() + term_coeff.size() - 4);
            }
            gate_terms.push_back((std::uint32_t)(term_factors.size() - 1));
            zkhip_gate_program prog;
            prog.n_gates = (std::uint32_t)gate_sel.size();
            prog.n_terms = (std::uint32_t)(term_factors.size() - 1);
            prog.n_factors = (std::uint32_t)factor_slot.size();
            prog.n_slots = (std::uint32_t)U;
            prog.gate_terms = gate_terms.data();
            prog.gate_selector = gate_sel.data();
            prog.gate_selector_rot = gate_sel_rot.data();
            prog.term_factors = term_factors.data();
            prog.factor_slot = factor_slot.data();
            prog.factor_rot = factor_rot.data();
            prog.term_coeff = term_coeff.data();
            check(zkhip_gate_eval_dev(ctx.get(), adapter::id, &prog, ext_of.data(), log_e, last && mask_ptr ? ext_of[U] : nullptr, lo != 0 ? 1 : 0, F.data()),
                  "zkhip_gate_eval_dev", ctx.get());
            lo = hi;
            ctx.sync();    // the group's extensions are released at the end of the iteration
        }
        if (F_degree + mask_degree >= extended_size) throw std::invalid_argument("gate_argument: mask * F does not fit the extended domain");
        F.set_degree(F_degree + mask_degree);
        return F;
    }

    /// Round 5's evaluation of the same sum: per product a k-way pointwise product and a scaled accumulation (two launches, each streaming
    /// whole vectors), every (column, rotation) pair extended on its own.  Same polynomial, bit for bit.
    static dfs_type gate_argument_per_term(const context &ctx, const std::vector<gate_product_hip<CurveType>> &products, const dfs_type &mask_polynomial,
                                           std::size_t extended_size, const root_of_unity_type &root, std::size_t slot_budget = (std::size_t)16 << 30) {
        if (products.empty()) throw std::invalid_argument("gate_argument: no products");
        dfs_type F(ctx, extended_size);
        bool first = true;
        const std::size_t n = mask_polynomial.size();
        std::size_t log_n = 0, log_e = 0;
        while (((std::size_t)1 << log_n) < n) ++log_n;
        while (((std::size_t)1 << log_e) < extended_size) ++log_e;
        if (((std::size_t)1 << log_n) != n || ((std::size_t)1 << log_e) != extended_size || extended_size < n)
            throw std::invalid_argument("gate_argument: domain sizes must be powers of two, extended >= original");
        std::uint64_t wn[4], we[4];
        adapter::scalar_to_limbs(root(log_n), wn);
        adapter::scalar_to_limbs(root(log_e), we);
        /* Every DISTINCT (column, rotation) the gates name is extended ONCE (the reference's evaluator caches a variable's extension the same way):
           the distinct factors of a group of gates -- and, with the first group, the mask -- are laid side by side (the rotation of a factor
           writes straight into its slot) and extended by ONE call (a batched inverse NTT of n points + a batched NTT of extended_size points: the
           transforms then run their pairs-per-workgroup kernels); every gate of the group is one k-way pointwise product over its slots.
           A group ends where its extensions would pass `slot_budget` bytes. */
        const std::size_t max_slots = std::max<std::size_t>(2, slot_budget / (extended_size * 32));
        std::shared_ptr<void> d_mask_ext;
        for (std::size_t lo = 0; lo < products.size();) {
            std::vector<std::pair<const dfs_type *, int>> unique;
            auto slot_of = [&unique](const dfs_type *f, int rot) {
                for (std::size_t u = 0; u < unique.size(); ++u)
                    if (unique[u].first == f && unique[u].second == rot) return u;
                return unique.size();
            };
            std::size_t hi = lo;
            for (; hi < products.size(); ++hi) {
                const auto &g = products[hi];
                if (g.factors.empty() || g.rotations.size() != g.factors.size()) throw std::invalid_argument("gate_argument: factors / rotations");
                std::vector<std::pair<const dfs_type *, int>> added;
                for (std::size_t k = 0; k < g.factors.size(); ++k) {
                    const std::pair<const dfs_type *, int> key(g.factors[k], g.rotations[k]);
                    if (slot_of(key.first, key.second) == unique.size() && std::find(added.begin(), added.end(), key) == added.end()) added.push_back(key);
                }
                if (hi > lo && unique.size() + added.size() + (first ? 1 : 0) > max_slots) break;
                unique.insert(unique.end(), added.begin(), added.end());
            }
            const std::size_t U = unique.size();
            /* Round 5: a factor (or the mask) whose holder switched its extension cache on -- a selector, a preprocessed column, a witness column
               other arguments extend as well -- is taken from device_polynomial_dfs::extension() (computed once, rotated on the extended domain when
               the gate asks for a rotation); the others are laid side by side and extended by ONE batched call as before */
            std::vector<const void *> ext_of(U + 1, nullptr);    // where unique factor u (and, at U, the mask) lives on the extended domain
            std::vector<dfs_type> keep;                          // cached / rotated extensions stay alive until the group is done
            std::vector<std::size_t> batched;                    // the factors (U = the mask) that go through the batched extension, in slot order
            const bool mask_here = first;
            for (std::size_t u = 0; u <= U; ++u) {
                if (u == U && !mask_here) break;
                const dfs_type &f = u < U ? *unique[u].first : mask_polynomial;
                if (f.size() != n) throw std::invalid_argument("gate_argument: every factor lives on the original domain (the mask's size)");
                if (extended_size > n && f.extension_cache_enabled()) {
                    dfs_type e = f.extension(extended_size, root);
                    const int rot = u < U ? unique[u].second : 0;
                    if (rot) e = polynomial_shift(e, rot, n);
                    ext_of[u] = e.data();
                    keep.push_back(std::move(e));
                } else {
                    batched.push_back(u);
                }
            }
            const std::size_t slots = batched.size();
            auto d_in = ctx.alloc(std::max<std::size_t>(1, slots) * n * 32);
            auto d_ext = ctx.alloc(std::max<std::size_t>(1, slots) * extended_size * 32);
            for (std::size_t b = 0; b < slots; ++b) {
                const std::size_t u = batched[b];
                const dfs_type &f = u < U ? *unique[u].first : mask_polynomial;
                const int rot = u < U ? unique[u].second : 0;
                char *slot = static_cast<char *>(d_in.get()) + b * n * 32;
                if (rot) check(zkhip_poly_shift_dev(ctx.get(), f.data(), log_n, (std::int64_t)rot, slot), "zkhip_poly_shift_dev", ctx.get());
                else check(zkhip_memcpy_d2d_async(ctx.get(), slot, f.data(), n * 32), "zkhip_memcpy_d2d_async", ctx.get());
                ext_of[u] = static_cast<const char *>(d_ext.get()) + b * extended_size * 32;
            }
            if (slots) {
                if (extended_size == n) check(zkhip_memcpy_d2d_async(ctx.get(), d_ext.get(), d_in.get(), slots * n * 32), "zkhip_memcpy_d2d_async", ctx.get());
                else check(zkhip_poly_resize_dev(ctx.get(), adapter::id, d_in.get(), log_n, slots, wn, d_ext.get(), log_e, we), "zkhip_poly_resize_dev", ctx.get());
            }
            dfs_type term(ctx, extended_size);
            for (std::size_t gi = lo; gi < hi; ++gi) {
                const auto &g = products[gi];
                std::size_t degree = 0;
                std::vector<const void *> ptrs;
                for (std::size_t k = 0; k < g.factors.size(); ++k) {
                    degree += g.factors[k]->degree();
                    ptrs.push_back(ext_of[slot_of(g.factors[k], g.rotations[k])]);
                }
                if (degree >= extended_size) throw std::invalid_argument("gate_argument: the product's degree does not fit the extended domain");
                check(zkhip_fr_vec_prod_dev(ctx.get(), adapter::id, ptrs.size(), ptrs.data(), term.data(), extended_size), "zkhip_fr_vec_prod_dev", ctx.get());
                /* F (+)= coefficient * term */
                std::uint64_t c[4];
                adapter::scalar_to_limbs(g.coefficient, c);
                const void *tp = term.data();
                const bool very_first = first && gi == lo;
                check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, 1, &tp, &extended_size, c, 1, F.data(), extended_size, very_first ? 0 : 1), "zkhip_poly_lincomb_dev",
                      ctx.get());
                F.set_degree(very_first ? degree : std::max(F.degree(), degree));
            }
            if (first) {    // keep the extended mask for the final product
                d_mask_ext = ctx.alloc(extended_size * 32);
                check(zkhip_memcpy_d2d_async(ctx.get(), d_mask_ext.get(), ext_of[U], extended_size * 32), "zkhip_memcpy_d2d_async", ctx.get());
            }
            first = false;
            lo = hi;
            ctx.sync();    // d_in, d_ext and `term` are released at the end of the iteration
        }
        if (F.degree() + mask_polynomial.degree() >= extended_size) throw std::invalid_argument("gate_argument: mask * F does not fit the extended domain");
        check(zkhip_fr_vec_op_dev(ctx.get(), adapter::id, 2, F.data(), d_mask_ext.get(), F.data(), extended_size), "zkhip_fr_vec_op_dev", ctx.get());    // gates_argument.hpp:215
        F.set_degree(F.degree() + mask_polynomial.degree());
        ctx.sync();
        return F;
    }

    /// prover.hpp:262-277: F_consolidated = sum_i alphas[i] * F_dfs[i] (math::polynomial_sum: on the largest of the domains; a part that
    /// is_zero() -- size 0 here -- is skipped as :267-269 does), its coefficients, and the exact quotient by Z = X^rows_amount - 1.
    /// Throws when the division leaves a remainder (the circuit is not satisfied): the reference would commit to a wrong quotient.
    static device_coefficients quotient_polynomial(const context &ctx, const std::vector<dfs_type> &F_dfs, const std::vector<value_type> &alphas,
                                                   std::size_t rows_amount, const root_of_unity_type &root) {
        if (F_dfs.size() != alphas.size() || F_dfs.empty()) throw std::invalid_argument("quotient_polynomial: one alpha per part");
        std::size_t size = 0;
        for (const auto &f : F_dfs) size = std::max(size, f.size());
        if (size < 2 * rows_amount) size = 2 * rows_amount;    // at least one quotient coefficient block
        /* The reference adds the parts in DFS form on the largest domain and inverts the sum (:262-273).  The coefficients of a sum are the sum
           of the coefficients, and a weighted sum of evaluation vectors over ONE domain is the evaluation vector of the weighted sum: the parts
           that live on the same domain are added there first (one pass), every such sum is inverted on ITS OWN domain (round 4 inverted every
           part: 34 n transform points for placeholder's eight parts on 2n / 4n / 8n; round 5: one inverse transform per domain size, 14 n), and
           the coefficient vectors are added in one pass -- the same field elements as the reference's extension of everything to the largest
           domain (74 n) */
        std::map<std::size_t, std::vector<std::size_t>> by_size;
        for (std::size_t i = 0; i < F_dfs.size(); ++i)
            if (F_dfs[i].size() != 0) by_size[F_dfs[i].size()].push_back(i);
        std::vector<std::shared_ptr<void>> parts;
        std::vector<const void *> ptrs;
        std::vector<std::size_t> lens;
        std::vector<std::uint64_t> coeffs;
        std::uint64_t one_limbs[4];
        adapter::scalar_to_limbs(value_type::one(), one_limbs);
        for (const auto &group : by_size) {
            const std::size_t gsize = group.first;
            std::vector<const void *> gp;
            std::vector<std::size_t> gl;
            std::vector<std::uint64_t> gc;
            for (std::size_t i : group.second) {
                gp.push_back(F_dfs[i].data());
                gl.push_back(gsize);
                std::uint64_t a[4];
                adapter::scalar_to_limbs(alphas[i], a);
                gc.insert(gc.end(), a, a + 4);
            }
            auto acc = ctx.alloc(gsize * 32);    // sum_i alpha_i F_i over this domain, then its coefficients in place
            check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, gp.size(), gp.data(), gl.data(), gc.data(), 1, acc.get(), gsize, 0), "zkhip_poly_lincomb_dev", ctx.get());
            std::size_t log_g = 0;
            while (((std::size_t)1 << log_g) < gsize) ++log_g;
            if (((std::size_t)1 << log_g) != gsize) throw std::invalid_argument("quotient_polynomial: a part's size is not a power of two");
            std::uint64_t w[4];
            adapter::scalar_to_limbs(root(log_g), w);
            check(zkhip_ntt_dev(ctx.get(), adapter::id, acc.get(), log_g, 1, w, 1, nullptr), "zkhip_ntt_dev", ctx.get());
            parts.push_back(acc);
            ptrs.push_back(acc.get());
            lens.push_back(gsize);
            coeffs.insert(coeffs.end(), one_limbs, one_limbs + 4);
        }
        dfs_type F(ctx, size);    // holds COEFFICIENTS from here on
        check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, ptrs.size(), ptrs.data(), lens.data(), coeffs.data(), 1, F.data(), size, 0), "zkhip_poly_lincomb_dev",
              ctx.get());
        /* T_consolidated = F_consolidated_normal / common_data.Z (:275) */
        device_coefficients T;
        T.size = size - rows_amount;
        T.data = ctx.alloc(T.size * 32);
        std::uint64_t bad = 0;
        check(zkhip_poly_div_vanishing_dev(ctx.get(), adapter::id, F.data(), size, rows_amount, T.data.get(), &bad), "zkhip_poly_div_vanishing_dev", ctx.get());
        if (bad) throw std::runtime_error("quotient_polynomial: F_consolidated is not divisible by the vanishing polynomial");
        return T;
    }

    /// prover.hpp:220-259: detail::split_polynomial(T, rows_amount - 1) -- chunks of rows_amount coefficients (:55-69) -- and
    /// `T_splitted_dfs[k].from_coefficients(T_splitted[k])` over the `dfs_size`-point domain (|_F_dfs[0]|, :251-252) for
    /// `split_polynomial_size` parts; parts beyond the quotient's length are the zero polynomial, as the reference initialises them.
    static std::vector<dfs_type> quotient_polynomial_split_dfs(const context &ctx, const device_coefficients &T, std::size_t rows_amount,
                                                               std::size_t split_polynomial_size, std::size_t dfs_size, const root_of_unity_type &root) {
        const std::size_t chunks = (T.size + rows_amount - 1) / rows_amount;
        /* the reference writes T_splitted_dfs[k] for EVERY chunk (:255-257): more chunks than parts is out of bounds there -- refused here */
        if (chunks > split_polynomial_size) throw std::invalid_argument("quotient_polynomial_split_dfs: the quotient needs more parts than split_polynomial_size");
        if (dfs_size < rows_amount) throw std::invalid_argument("quotient_polynomial_split_dfs: dfs_size < rows_amount");
        std::vector<dfs_type> out;
        std::size_t log_size = 0;
        while (((std::size_t)1 << log_size) < dfs_size) ++log_size;
        std::uint64_t w[4];
        adapter::scalar_to_limbs(root(log_size), w);
        for (std::size_t k = 0; k < split_polynomial_size; ++k) {
            dfs_type p(ctx, dfs_size);
            const std::size_t lo = k * rows_amount, len = lo < T.size ? std::min(rows_amount, T.size - lo) : 0;
            /* zero-padded chunk -> evaluations (from_coefficients) */
            const void *src = len ? T.at(lo) : nullptr;
            std::uint64_t one[4];
            adapter::scalar_to_limbs(value_type::one(), one);
            check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, len ? 1 : 0, &src, &len, one, 1, p.data(), dfs_size, 0), "zkhip_poly_lincomb_dev", ctx.get());
            check(zkhip_ntt_dev(ctx.get(), adapter::id, p.data(), log_size, 1, w, 0, nullptr), "zkhip_ntt_dev", ctx.get());
            p.set_degree(len ? len - 1 : 0);
            out.push_back(std::move(p));
        }
        ctx.sync();
        return out;
    }

    /// The same parts left in COEFFICIENT form (zero-padded to dfs_size): what T_commit's commitment scheme needs of them (prover.hpp:314-317 --
    /// the only consumer of T_splitted_dfs in the reference).  kzg_commitment_scheme_v2_hip::append_to_batch takes them as they are, which
    /// saves the from_coefficients of :255 and the coefficients() of kzg.hpp:431: 2 dfs_size transform points per part.
    static std::vector<device_polynomial_coefficients<CurveType>> quotient_polynomial_split_coefficients(const context &ctx, const device_coefficients &T,
                                                                                                         std::size_t rows_amount, std::size_t split_polynomial_size,
                                                                                                         std::size_t dfs_size) {
        const std::size_t chunks = (T.size + rows_amount - 1) / rows_amount;
        if (chunks > split_polynomial_size) throw std::invalid_argument("quotient_polynomial_split_coefficients: the quotient needs more parts than split_polynomial_size");
        if (dfs_size < rows_amount || (dfs_size & (dfs_size - 1))) throw std::invalid_argument("quotient_polynomial_split_coefficients: dfs_size must be a power of two >= rows_amount");
        std::vector<device_polynomial_coefficients<CurveType>> out;
        std::uint64_t one[4];
        adapter::scalar_to_limbs(value_type::one(), one);
        for (std::size_t k = 0; k < split_polynomial_size; ++k) {
            device_polynomial_coefficients<CurveType> p(ctx, dfs_size);
            const std::size_t lo = k * rows_amount, len = lo < T.size ? std::min(rows_amount, T.size - lo) : 0;
            const void *src = len ? T.at(lo) : nullptr;
            check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, len ? 1 : 0, &src, &len, one, 1, p.data(), dfs_size, 0), "zkhip_poly_lincomb_dev", ctx.get());
            p.set_known_zero(len == 0);    // a part beyond the quotient's length
            out.push_back(std::move(p));
        }
        return out;
    }
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_PLACEHOLDER_QUOTIENT_HPP
