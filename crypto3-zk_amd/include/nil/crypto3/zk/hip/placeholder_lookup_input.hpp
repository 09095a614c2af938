//---------------------------------------------------------------------------//
// zkhip shim: the NUMERIC side of placeholder's prepare_lookup_input on the device.
//
// Mirrors zk/snark/systems/plonk/placeholder/lookup_argument.hpp:435-496: for every lookup gate and every constraint of it
//     l = lookup_selector * table_id + sum_k theta^(k+1) * lookup_selector * evaluate(constraint.lookup_input[k])        (:478-491)
// with the variables read from the plonk columns and moved by their rotations (math::polynomial_shift over the basic domain, :463-465).
// WHICH monomials an expression consists of is decided by the reference's symbolic machinery (math::expression, the converter and the
// cached evaluator: the caller's -- SURVEY section 2 out of scope); once an expression is a list of monomials the whole constraint is
//     l = lookup_selector * ( table_id + sum_k theta^(k+1) sum_m c_m prod_f column_f(omega^rotation_f X) )
// -- ONE gate of the flat-program kernel (zkhip_gate_eval_dev: the selector multiplied once, rotations as index arithmetic, every
// column extended once), on the domain the reference's polynomial_dfs arithmetic ends up on: the smallest power-of-two domain that
// holds the degree (operator* grows the domain with the degree), at least the basic one.
//
// placeholder_lookup_argument_prover_hip (placeholder_arguments.hpp) takes `prepare_lookup_input` as a callable of theta: a caller with
// flattened constraints passes  [&](const value_type &theta) { return prepare_lookup_input_flat<CurveType>(ctx, constraints, theta, root); }.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_PLACEHOLDER_LOOKUP_INPUT_HPP
#define ZKHIP_SHIM_PLACEHOLDER_LOOKUP_INPUT_HPP

#include <stdexcept>
#include <vector>

#include "placeholder_quotient.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// One constraint of a lookup gate (plonk_lookup_constraint: table_id + lookup_input expressions; lookup_argument.hpp:478-491) with its
/// expressions flattened into monomials over resident columns.
template <typename CurveType>
struct lookup_input_constraint_hip {
    typedef device_polynomial_dfs<CurveType> dfs_type;
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    struct monomial {
        value_type coefficient;
        std::vector<const dfs_type *> factors;    // may be empty: a constant
        std::vector<int> rotations;               // one per factor, in rows of the basic domain
    };
    const dfs_type *lookup_selector = nullptr;    // plonk_columns.selector(gate.tag_index) (:476)
    std::size_t table_id = 0;                     // constraint.table_id (:478)
    std::vector<std::vector<monomial>> lookup_input;    // [k]: the monomials of constraint.lookup_input[k] (:480-487)
};

/// prepare_lookup_input (:435-496) over flattened constraints: one polynomial per constraint, in the constraints' order.
template <typename CurveType>
std::vector<device_polynomial_dfs<CurveType>> prepare_lookup_input_flat(const context &ctx, const std::vector<lookup_input_constraint_hip<CurveType>> &constraints,
                                                                        const typename curve_adapter<CurveType>::scalar_value_type &theta,
                                                                        const typename device_polynomial_dfs<CurveType>::root_of_unity_type &root) {
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    typedef placeholder_quotient_hip<CurveType> Q;
    std::vector<device_polynomial_dfs<CurveType>> out;
    for (const auto &c : constraints) {
        if (!c.lookup_selector) throw std::invalid_argument("prepare_lookup_input_flat: a constraint without its gate's selector");
        const std::size_t n = c.lookup_selector->size();
        std::vector<gate_product_hip<CurveType>> products;
        {   /* l = lookup_selector * table_id (:478) */
            gate_product_hip<CurveType> p;
            p.factors = {c.lookup_selector};
            p.rotations = {0};
            p.coefficient = value_type((std::uint64_t)c.table_id);
            products.push_back(std::move(p));
        }
        value_type theta_acc = theta;    // (:479)
        std::size_t degree = c.lookup_selector->degree();
        for (const auto &expression : c.lookup_input) {
            for (const auto &m : expression) {    /* l += theta_acc * lookup_selector * evaluate(expression) (:485), monomial by monomial */
                if (m.factors.size() != m.rotations.size()) throw std::invalid_argument("prepare_lookup_input_flat: factors / rotations");
                gate_product_hip<CurveType> p;
                p.factors.push_back(c.lookup_selector);
                p.rotations.push_back(0);
                std::size_t d = c.lookup_selector->degree();
                for (std::size_t f = 0; f < m.factors.size(); ++f) {
                    if (!m.factors[f] || m.factors[f]->size() != n) throw std::invalid_argument("prepare_lookup_input_flat: every column lives on the basic domain");
                    p.factors.push_back(m.factors[f]);
                    p.rotations.push_back(m.rotations[f]);
                    d += m.factors[f]->degree();
                }
                p.coefficient = theta_acc * m.coefficient;
                degree = std::max(degree, d);
                products.push_back(std::move(p));
            }
            theta_acc = theta_acc * theta;    // (:486)
        }
        std::size_t size = n;
        while (size < degree + 1) size <<= 1;    // polynomial_dfs arithmetic grows the domain with the degree
        out.push_back(Q::sum_of_products(ctx, products, nullptr, size, root));
    }
    return out;
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_PLACEHOLDER_LOOKUP_INPUT_HPP
