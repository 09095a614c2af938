//---------------------------------------------------------------------------//
// zkhip shim: placeholder's permutation and lookup arguments behind the REFERENCE'S OWN entry points -- the control flow around
// the numeric bodies of placeholder_permutation.hpp / placeholder_lookup.hpp: which challenges are drawn from the transcript and
// when, what is appended to which batch and when.  That order decides whether a proof is byte-identical, so it is restated here
// once instead of in every caller.
//
//   placeholder_permutation_argument_hip<Curve>::prove_eval(constraint_system, preprocessed_data, table_description,
//                                                           column_polynomials, commitment_scheme, transcript)
//       zk/snark/systems/plonk/placeholder/permutation_argument.hpp:70-224
//         :95-97    beta, gamma = challenge, challenge
//         :139      commitment_scheme.append_to_batch(PERMUTATION_BATCH, V_P)
//         :181-183  permutation_parts - 1 challenges (the parts' alphas)
//         :200      commitment_scheme.append_to_batch(PERMUTATION_BATCH, current_poly) per part but the last
//   placeholder_lookup_argument_prover_hip<Curve, ...>(constraint_system, preprocessed_data, plonk_columns, commitment_scheme,
//                                                      transcript).prove_eval()
//       zk/snark/systems/plonk/placeholder/lookup_argument.hpp:128-296
//         :150      theta = challenge                                              (constructor)
//         :163-170  mask = 1 - q_last - q_blind; prepare_lookup_value (:411-433) -- on the device, below; prepare_lookup_input
//                   (:435-496) is expression evaluation over the columns, the constraint system's symbolic side: a hook
//         :175-189  reduce, sort_polynomials (:565-638)                            -- zkhip_lookup_sort_dev
//         :192-196  append_to_batch(LOOKUP_BATCH, sorted[i]) ..., commit(LOOKUP_BATCH), transcript(lookup_commitment)
//         :199-206  beta, gamma = challenge, challenge; lookup_parts(max_quotient_chunks).size() - 1 challenges
//         :213      append_to_batch(PERMUTATION_BATCH, V_L);  :267 append_to_batch(PERMUTATION_BATCH, current_poly) per part but the last
//         :282-283  one challenge per sorted vector but the first (F_dfs[3]'s alphas)
//
// Everything is duck-typed on the reference's member names (preprocessed_data.{permutation_polynomials, identity_polynomials, q_last,
// q_blind, common_data.{lagrange_0, max_quotient_chunks, permutation_parts, desc.usable_rows_amount}}, constraint_system.{permuted_columns,
// lookup_tables, lookup_parts}, table_description.global_index, plonk_columns.{selector, constant}, commitment_scheme.{append_to_batch,
// commit}, transcript(x), transcript.template challenge<FieldType>() -- or .challenge() as the value-level scheme classes of this shim
// draw theirs).  A polynomial may be any class read through size() / operator[] (uploaded) or a device_polynomial_dfs (taken where it
// lies: preprocessed data that stays resident across proofs costs no upload).  The roots of unity default to the field's
// (curve_adapter::root_of_unity -- the reference reads them from preprocessed_data.common_data.basic_domain, crypto3-math's).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_PLACEHOLDER_ARGUMENTS_HPP
#define ZKHIP_SHIM_PLACEHOLDER_ARGUMENTS_HPP

#include <array>
#include <functional>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "multiexp.hpp"
#include "placeholder_lookup.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// the batch numbering of placeholder's proof (systems/plonk/placeholder/proof.hpp:37-41)
constexpr std::size_t FIXED_VALUES_BATCH = 0, VARIABLE_VALUES_BATCH = 1, PERMUTATION_BATCH = 2, QUOTIENT_BATCH = 3, LOOKUP_BATCH = 4;

namespace detail {
    template <int N>
    struct priority : priority<N - 1> { };
    template <>
    struct priority<0> { };

    /// transcript.template challenge<FieldType>() where the curve names its scalar field (the reference's transcripts), .challenge() otherwise
    template <typename CurveType, typename Transcript>
    auto draw_challenge(Transcript &t, priority<1>) -> decltype(t.template challenge<typename CurveType::scalar_field_type>()) {
        return t.template challenge<typename CurveType::scalar_field_type>();
    }
    template <typename CurveType, typename Transcript>
    auto draw_challenge(Transcript &t, priority<0>) -> decltype(t.challenge()) {
        return t.challenge();
    }

    /// a polynomial as the device sees it: a device_polynomial_dfs is taken where it lies, anything else is uploaded (degree() kept when the class has one)
    template <typename PolynomialType>
    auto degree_of(const PolynomialType &p, priority<1>) -> decltype(p.degree()) {
        return p.degree();
    }
    template <typename PolynomialType>
    std::size_t degree_of(const PolynomialType &p, priority<0>) {
        return p.size() ? p.size() - 1 : 0;
    }
    template <typename CurveType>
    device_polynomial_dfs<CurveType> on_device(const context &, const device_polynomial_dfs<CurveType> &p) {
        return p;
    }
    template <typename CurveType, typename PolynomialType, typename = typename std::enable_if<!std::is_same<PolynomialType, device_polynomial_dfs<CurveType>>::value>::type>
    device_polynomial_dfs<CurveType> on_device(const context &ctx, const PolynomialType &p) {
        return device_polynomial_dfs<CurveType>(ctx, p, (std::size_t)degree_of(p, priority<1>()));
    }
    template <typename CurveType, typename Container>
    std::vector<device_polynomial_dfs<CurveType>> all_on_device(const context &ctx, const Container &c) {
        std::vector<device_polynomial_dfs<CurveType>> out;
        for (const auto &p : c) out.push_back(on_device<CurveType>(ctx, p));
        return out;
    }
    template <typename CurveType>
    typename device_polynomial_dfs<CurveType>::root_of_unity_type field_roots() {
        return [](std::size_t log_n) { return curve_adapter<CurveType>::root_of_unity(log_n); };
    }
}    // namespace detail

template <typename CurveType>
struct placeholder_permutation_argument_hip {
    typedef placeholder_permutation_hip<CurveType> body;
    typedef typename body::value_type value_type;
    typedef typename body::dfs_type dfs_type;
    typedef typename body::root_of_unity_type root_of_unity_type;
    typedef typename body::prover_result_type prover_result_type;
    static constexpr std::size_t argument_size = 3;

    /// permutation_argument.hpp:70-78, the reference's argument list; `root` / `ctx`: see the header comment
    template <typename ConstraintSystem, typename PreprocessedData, typename TableDescription, typename ColumnPolynomials, typename CommitmentScheme, typename Transcript>
    static prover_result_type prove_eval(const ConstraintSystem &constraint_system, const PreprocessedData &preprocessed_data, const TableDescription &table_description,
                                         const ColumnPolynomials &column_polynomials, CommitmentScheme &commitment_scheme, Transcript &transcript,
                                         const root_of_unity_type &root = detail::field_roots<CurveType>(), const context &ctx = default_context()) {
        ZKHIP_PROFILE_SCOPE("permutation_argument_prove_eval_time");
        const auto &common = preprocessed_data.common_data;
        /* :86-97 */
        std::vector<dfs_type> columns;
        for (const auto &var : constraint_system.permuted_columns()) columns.push_back(detail::on_device<CurveType>(ctx, column_polynomials[table_description.global_index(var)]));
        const std::vector<dfs_type> S_sigma = detail::all_on_device<CurveType>(ctx, preprocessed_data.permutation_polynomials),
                                    S_id = detail::all_on_device<CurveType>(ctx, preprocessed_data.identity_polynomials);
        const dfs_type q_last = detail::on_device<CurveType>(ctx, preprocessed_data.q_last), q_blind = detail::on_device<CurveType>(ctx, preprocessed_data.q_blind),
                       lagrange_0 = detail::on_device<CurveType>(ctx, common.lagrange_0);
        /* 1. beta_1, gamma_1 = challenge */
        const value_type beta = detail::draw_challenge<CurveType>(transcript, detail::priority<1>());
        const value_type gamma = detail::draw_challenge<CurveType>(transcript, detail::priority<1>());
        typename body::hooks_type hooks;
        /* 4. V_P to the batch the prover commits after this call (prover.hpp:170-171) */
        hooks.on_V_P = [&commitment_scheme](const dfs_type &V_P) { commitment_scheme.append_to_batch(PERMUTATION_BATCH, V_P); };
        hooks.draw_alphas = [&transcript, &common](std::size_t parts) {
            if (parts != (std::size_t)common.permutation_parts)
                throw std::invalid_argument("permutation argument: common_data.permutation_parts does not follow from max_quotient_chunks and the permuted columns");
            std::vector<value_type> alphas;
            for (std::size_t i = 0; i + 1 < parts; ++i) alphas.push_back(detail::draw_challenge<CurveType>(transcript, detail::priority<1>()));
            return alphas;
        };
        hooks.on_part = [&commitment_scheme](const dfs_type &current) { commitment_scheme.append_to_batch(PERMUTATION_BATCH, current); };
        return body::prove_eval_hooked(ctx, columns, S_id, S_sigma, q_last, q_blind, lagrange_0, beta, gamma, root, (std::size_t)common.max_quotient_chunks,
                                       (std::size_t)common.desc.usable_rows_amount, hooks);
    }
};

/// lookup_argument.hpp:108-296.  PrepareLookupInput: `std::vector<polynomial>(const value_type &theta)` -- prepare_lookup_input (:435-496), the
/// walk over the lookup gates that evaluates their input expressions over the columns (math::expression: the constraint system's symbolic
/// machinery, which stays the caller's); it is called where the reference calls its own, and may return host polynomials or device ones.
template <typename CurveType, typename ConstraintSystem, typename PreprocessedData, typename PlonkColumns, typename CommitmentScheme, typename Transcript>
class placeholder_lookup_argument_prover_hip {
public:
    typedef placeholder_lookup_hip<CurveType> body;
    typedef typename body::value_type value_type;
    typedef typename body::dfs_type dfs_type;
    typedef typename body::root_of_unity_type root_of_unity_type;
    typedef CommitmentScheme commitment_scheme_type;
    static constexpr std::size_t argument_size = 4;

    struct prover_lookup_result {
        std::array<dfs_type, argument_size> F_dfs;
        typename commitment_scheme_type::commitment_type lookup_commitment;
    };

    template <typename PrepareLookupInput>
    placeholder_lookup_argument_prover_hip(const ConstraintSystem &constraint_system, const PreprocessedData &preprocessed_data, const PlonkColumns &plonk_columns,
                                           CommitmentScheme &commitment_scheme, Transcript &transcript, PrepareLookupInput prepare_lookup_input,
                                           const root_of_unity_type &root = detail::field_roots<CurveType>(), const context &ctx = default_context()) :
        constraint_system(constraint_system), preprocessed_data(preprocessed_data), plonk_columns(plonk_columns), commitment_scheme(commitment_scheme), transcript(transcript),
        root(root), ctx(ctx) {
        const context &c = ctx;
        prepare_input = [prepare_lookup_input, &c](const value_type &th) { return detail::all_on_device<CurveType>(c, prepare_lookup_input(th)); };
        /* theta = challenge (:150) */
        theta = detail::draw_challenge<CurveType>(transcript, detail::priority<1>());
    }

    prover_lookup_result prove_eval() {
        ZKHIP_PROFILE_SCOPE("Lookup argument prove eval time");
        const auto &common = preprocessed_data.common_data;
        const std::size_t usable_rows = (std::size_t)common.desc.usable_rows_amount;
        const dfs_type q_last = detail::on_device<CurveType>(ctx, preprocessed_data.q_last), q_blind = detail::on_device<CurveType>(ctx, preprocessed_data.q_blind),
                       lagrange_0 = detail::on_device<CurveType>(ctx, common.lagrange_0);
        const std::size_t n = q_last.size();
        const value_type one = value_type::one(), zero = value_type::zero();
        /* mask_assignment = one - q_last - q_blind (:162-163) */
        dfs_type mask_assignment = body::affine(q_last, &q_blind, zero - one, zero - one, one);
        const std::vector<dfs_type> lookup_value = prepare_lookup_value(mask_assignment);
        const std::vector<dfs_type> lookup_input = prepare_input(theta);
        /* 3. reduce, sort (:175-189) */
        std::vector<dfs_type> reduced_value, reduced_input;
        for (const auto &p : lookup_value) reduced_value.push_back(body::reduce_dfs_polynomial_domain(p, n));
        for (const auto &p : lookup_input) reduced_input.push_back(body::reduce_dfs_polynomial_domain(p, n));
        std::vector<dfs_type> sorted = body::sort_polynomials(ctx, reduced_input, reduced_value, n, usable_rows);
        for (auto &s : sorted) s.set_degree(n - 1);
        /* a looked-up value that is in no table / a table that overflows the sorted vectors: the reference's BOOST_ASSERTs (:583, :613-617).
           Checked HERE -- before anything is appended, committed or absorbed (the throw used to leave the commitment scheme and
           the transcript mutated) --, and only these two bits are the lookup argument's: any other sticky bit is reported as what it is */
        throw_on_device_status("after sort_polynomials");
        reduced_value.clear();
        reduced_input.clear();
        /* 4. commit the sorted polynomials (:192-196) */
        for (std::size_t i = 0; i < sorted.size(); ++i) commitment_scheme.append_to_batch(LOOKUP_BATCH, sorted[i]);
        typename commitment_scheme_type::commitment_type lookup_commitment = commitment_scheme.commit(LOOKUP_BATCH);
        transcript(lookup_commitment);
        /* 5. beta, gamma, the parts' alphas (:199-206) */
        const value_type beta = detail::draw_challenge<CurveType>(transcript, detail::priority<1>());
        const value_type gamma = detail::draw_challenge<CurveType>(transcript, detail::priority<1>());
        const auto parts = constraint_system.lookup_parts(common.max_quotient_chunks);
        std::vector<std::size_t> part_sizes(parts.begin(), parts.end());
        std::vector<value_type> lookup_alphas;
        for (std::size_t i = 0; i + 1 < part_sizes.size(); ++i) lookup_alphas.push_back(detail::draw_challenge<CurveType>(transcript, detail::priority<1>()));
        typename body::hooks_type hooks;
        hooks.on_V_L = [this](const dfs_type &V_L) { commitment_scheme.append_to_batch(PERMUTATION_BATCH, V_L); };
        hooks.on_part = [this](const dfs_type &current) { commitment_scheme.append_to_batch(PERMUTATION_BATCH, current); };
        /* one alpha per sorted vector but the first, drawn while F_dfs[3] is summed up (:281-283) */
        hooks.draw_alpha = [this]() { return detail::draw_challenge<CurveType>(transcript, detail::priority<1>()); };
        auto res = body::prove_eval_hooked(ctx, lookup_input, lookup_value, sorted, q_last, q_blind, lagrange_0, beta, gamma, usable_rows, root,
                                           part_sizes.size() == 1 ? std::vector<std::size_t>() : part_sizes, lookup_alphas, hooks);
        throw_on_device_status("after the grand product");    // nothing of the lookup argument's own can be raised here any more
        V_L_dfs.assign(1, res.V_L);
        parts_dfs = res.parts_dfs;
        sorted_dfs = sorted;
        return prover_lookup_result {std::move(res.F_dfs), std::move(lookup_commitment)};
    }

    /// the sticky device status word (zkhip.h), read and cleared: bits 2 / 3 are sort_polynomials' own (the reference's assertions); bits 0 / 1
    /// belong to whatever ran on this context before (a gather out of range, an MSM plan overflow) and are reported as such
    void throw_on_device_status(const char *where) const {
        const std::uint32_t flags = ctx.device_status();
        if (!flags) return;
        std::string msg;
        if (flags & 4u) msg += "lookup argument: a looked-up value is in no lookup table; ";
        if (flags & 8u) msg += "lookup argument: the sorted sequence does not fit |input| + |value| vectors (equal table values must be adjacent); ";
        if (flags & 1u) msg += "device status: a gather index was out of range (raised before or beside the lookup argument); ";
        if (flags & 2u) msg += "device status: an MSM large-bucket plan overflowed (raised before or beside the lookup argument); ";
        if (flags & ~15u) msg += "device status: unknown flag(s) " + std::to_string(flags & ~15u) + "; ";
        throw std::runtime_error(msg + "[" + where + "]");
    }

    /// prepare_lookup_value (:411-433) on the device: per table t and option o,
    ///   v = mask ((t + 1) tag_t + sum_i theta^(i + 1) tag_t constant(option[o][i])) = tag_t ((t + 1) + sum_i theta^(i + 1) constant_i) mask,
    /// a polynomial of degree up to 3 (n - 1) (the reference's operator* extends the domain with the degree): one linear combination of the
    /// option's constant columns + one polynomial_product.
    std::vector<dfs_type> prepare_lookup_value(const dfs_type &mask_assignment) const {
        std::vector<dfs_type> out;
        const auto &lookup_tables = constraint_system.lookup_tables();
        const std::size_t n = mask_assignment.size();
        std::size_t t_id = 0;
        for (const auto &l_table : lookup_tables) {
            const dfs_type lookup_tag = detail::on_device<CurveType>(ctx, plonk_columns.selector(l_table.tag_index));
            for (const auto &option : l_table.lookup_options) {
                std::vector<dfs_type> cols;
                std::vector<const void *> ptrs;
                std::vector<std::size_t> lens;
                std::vector<std::uint64_t> coeffs;
                value_type theta_acc = theta;
                for (std::size_t i = 0; i < (std::size_t)l_table.columns_number; ++i) {
                    cols.push_back(detail::on_device<CurveType>(ctx, plonk_columns.constant(option[i].index)));
                    if (cols.back().size() != n) throw std::invalid_argument("lookup argument: a table column's size differs from the basic domain's");
                    ptrs.push_back(cols.back().data());
                    lens.push_back(n);
                    coeffs.resize(coeffs.size() + 4);
                    body::adapter::scalar_to_limbs(theta_acc, coeffs.data() + coeffs.size() - 4);
                    theta_acc = theta_acc * theta;
                }
                dfs_type lin(ctx, n);
                if (!cols.empty())
                    check(zkhip_poly_lincomb_dev(ctx.get(), body::adapter::id, ptrs.size(), ptrs.data(), lens.data(), coeffs.data(), 1, lin.data(), n, 0), "zkhip_poly_lincomb_dev",
                          ctx.get());
                /* + (t_id + 1): x + c over the vector (a x with a = 0 when there is no column: the constant polynomial) */
                dfs_type shifted = body::affine(cols.empty() ? mask_assignment : lin, nullptr, cols.empty() ? value_type::zero() : value_type::one(), value_type::zero(),
                                                value_type((std::uint64_t)(t_id + 1)));
                shifted.set_degree(cols.empty() ? 0 : n - 1);
                ctx.sync();
                out.push_back(polynomial_product<CurveType>({lookup_tag, shifted, mask_assignment}, root));
            }
            ++t_id;
        }
        return out;
    }

    /// what the reference keeps in locals, for the caller's bookkeeping and the tests
    value_type theta;
    std::vector<dfs_type> V_L_dfs;    // one entry once prove_eval ran
    std::vector<dfs_type> parts_dfs, sorted_dfs;

private:

    const ConstraintSystem &constraint_system;
    const PreprocessedData &preprocessed_data;
    const PlonkColumns &plonk_columns;
    CommitmentScheme &commitment_scheme;
    Transcript &transcript;
    root_of_unity_type root;
    const context &ctx;
    std::function<std::vector<dfs_type>(const value_type &)> prepare_input;
};

/// deduces the five duck-typed classes (the reference's class template takes FieldType / ParamsType and names them itself)
template <typename CurveType, typename ConstraintSystem, typename PreprocessedData, typename PlonkColumns, typename CommitmentScheme, typename Transcript,
          typename PrepareLookupInput>
placeholder_lookup_argument_prover_hip<CurveType, ConstraintSystem, PreprocessedData, PlonkColumns, CommitmentScheme, Transcript> make_placeholder_lookup_argument_prover(
    const ConstraintSystem &constraint_system, const PreprocessedData &preprocessed_data, const PlonkColumns &plonk_columns, CommitmentScheme &commitment_scheme,
    Transcript &transcript, PrepareLookupInput prepare_lookup_input,
    const typename device_polynomial_dfs<CurveType>::root_of_unity_type &root = detail::field_roots<CurveType>(), const context &ctx = default_context()) {
    return placeholder_lookup_argument_prover_hip<CurveType, ConstraintSystem, PreprocessedData, PlonkColumns, CommitmentScheme, Transcript>(
        constraint_system, preprocessed_data, plonk_columns, commitment_scheme, transcript, prepare_lookup_input, root, ctx);
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_PLACEHOLDER_ARGUMENTS_HPP
