//---------------------------------------------------------------------------//
// zkhip shim: Groth16 key generation with the queries computed ON THE DEVICE.
//
// Mirrors
//   reductions::r1cs_to_qap<F>::instance_map_with_evaluation     zk/snark/reductions/r1cs_to_qap.hpp:138-187
//   r1cs_gg_ppzksnark_generator::deterministic_basic_process     .../r1cs_gg_ppzksnark/generator.hpp:240-377
//   r1cs_constraint_system::swap_AB_if_beneficial                .../constraint_satisfaction_problems/r1cs.hpp:192-215
// The reference evaluates the QAP at t on the host (O(nnz + m) field operations) and then runs five fixed-base
// batch exponentiations (algebra::batch_exp / kc_batch_exp over a window table, generator.hpp:187-214) -- the
// part that costs minutes at 2^20 constraints.  Here the host part is the same arithmetic (spread over a few
// threads) and every batch exponentiation is one zkhip_bases_from_scalars call whose result STAYS on the device as
// the proving key's query (A, B.g, B.h, H, L), ready for the prover: no point ever crosses PCIe.
//
// `deterministic_basic_process` takes the toxic waste (t, alpha, beta, gamma, delta) as arguments, exactly like the
// reference's testing entry point; `basic_process` draws it from the operating system's CSPRNG.  The group
// generators are the standard ones (the reference draws random generators, generator.hpp:161,172: any generator
// yields a valid key).  gamma only enters the verification key, which this backend does not build (verifier and
// pairings are out of scope); it is accepted for signature parity.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_R1CS_GG_PPZKSNARK_GENERATOR_HPP
#define ZKHIP_SHIM_R1CS_GG_PPZKSNARK_GENERATOR_HPP

#include <algorithm>
#include <array>
#include <memory>
#include <vector>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "r1cs_gg_ppzksnark.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// qap_instance_evaluation (reductions/qap.hpp) restricted to what the generator reads
template <typename CurveType>
struct qap_instance_evaluation_hip {
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    std::size_t num_variables = 0, degree = 0, num_inputs = 0;
    value_type t, Zt;
    std::vector<value_type> At, Bt, Ct, Ht;    // N + 1 each; Ht = (1, t, ..., t^m)
};

/// r1cs_to_qap<F>::instance_map_with_evaluation(cs, t) (r1cs_to_qap.hpp:138-187) over the evaluation domain `dom` describes:
/// make_evaluation_domain(num_constraints + num_inputs + 1)'s choice unless it names another.  `ConstraintSystem` is
/// duck-typed like device_r1cs.
/// `ctx` (nullable): evaluate the Lagrange basis at t on the device (the dominant cost of this function at 2^20 points on the host).
template <typename CurveType, typename ConstraintSystem>
qap_instance_evaluation_hip<CurveType> instance_map_with_evaluation(const ConstraintSystem &cs, const typename curve_adapter<CurveType>::scalar_value_type &t,
                                                                    const domain_params<CurveType> &dom, const context *ctx = nullptr) {
    typedef typename curve_adapter<CurveType>::scalar_value_type Fr;
    const std::size_t M = cs.num_constraints(), n = cs.num_inputs(), N = cs.num_variables();
    const evaluation_domain_hip<CurveType> domain = evaluation_domain_hip<CurveType>::make(dom, M + n + 1);
    const std::size_t m = domain.m;
    qap_instance_evaluation_hip<CurveType> q;
    q.num_variables = N;
    q.degree = m;
    q.num_inputs = n;
    q.t = t;
    q.Zt = domain.compute_vanishing_polynomial(t);
    const std::vector<Fr> u = ctx ? domain.evaluate_all_lagrange_polynomials(*ctx, t) : domain.evaluate_all_lagrange_polynomials(t);
    q.At.assign(N + 1, Fr::zero());
    q.Bt.assign(N + 1, Fr::zero());
    q.Ct.assign(N + 1, Fr::zero());
    /* the constraints input_i * 0 = 0 that make the input consistent (r1cs_to_qap.hpp:160-162) */
    for (std::size_t i = 0; i <= n; ++i) q.At[i] = u[M + i];
    /* A_j(t) = sum_i u_i A_ij (r1cs_to_qap.hpp:164-178), split over host threads BY VARIABLE RANGE: every thread walks all the
       constraints and takes the terms whose variable falls into its range -- no two threads write the same entry, and the field
       products (the cost) are shared out; coefficients 1 (most of a circuit) need no product at all */
    detail::parallel_chunks(N + 1, [&](std::size_t lo, std::size_t hi) {
        const Fr one = Fr::one();
        auto take = [&](std::vector<Fr> &dst, const decltype(cs.constraints[0].a.terms) &terms, const Fr &ui) {
            for (const auto &term : terms) {
                if (term.index < lo || term.index >= hi) continue;
                dst[term.index] = dst[term.index] + (term.coeff == one ? ui : ui * term.coeff);
            }
        };
        for (std::size_t i = 0; i < M; ++i) {
            const auto &c = cs.constraints[i];
            take(q.At, c.a.terms, u[i]);
            take(q.Bt, c.b.terms, u[i]);
            take(q.Ct, c.c.terms, u[i]);
        }
    });
    q.Ht.resize(m + 1);
    detail::parallel_chunks(m + 1, [&](std::size_t lo, std::size_t hi) {
        Fr ti = detail::pow_u64(t, lo);
        for (std::size_t i = lo; i < hi; ++i) {
            q.Ht[i] = ti;
            ti = ti * t;
        }
    });
    return q;
}

/// A generated key: the host part (five group elements + the constraint system the prover's witness map runs over, A / B
/// swapped when that makes the B query lighter) and the device part (the resident queries).  `device` refers to `host`.
template <typename CurveType>
struct generated_proving_key {
    r1cs_gg_ppzksnark_proving_key<CurveType> host;    // its query vectors stay empty: the queries live on the device
    std::unique_ptr<r1cs_gg_ppzksnark_proving_key_hip<CurveType>> device;
};

/// A key generated over a device group: one generated_proving_key per member (slice k of every query, generated ON member k's GPU)
/// and the group key over them.
template <typename CurveType>
struct generated_proving_key_group {
    std::vector<std::unique_ptr<generated_proving_key<CurveType>>> parts;
    std::unique_ptr<r1cs_gg_ppzksnark_proving_key_group_hip<CurveType>> device;
};

template <typename CurveType>
class r1cs_gg_ppzksnark_generator_hip {
    typedef curve_adapter<CurveType> adapter;

public:
    typedef typename adapter::scalar_value_type scalar_value_type;
    typedef r1cs_constraint_system<CurveType> constraint_system_type;

    /// generator.hpp:240-377 (testing entry point: the toxic waste is an argument)
    static std::unique_ptr<generated_proving_key<CurveType>> deterministic_basic_process(const context &ctx, const constraint_system_type &constraint_system,
                                                                                         const domain_params<CurveType> &dom, const scalar_value_type &t,
                                                                                         const scalar_value_type &alpha, const scalar_value_type &beta,
                                                                                         const scalar_value_type & /*gamma*/, const scalar_value_type &delta,
                                                                                         std::size_t rank = 0, std::size_t world = 1) {
        typedef scalar_value_type Fr;
        std::unique_ptr<generated_proving_key<CurveType>> key(new generated_proving_key<CurveType>());
        auto &pk = key->host;
        /* ZKHIP_GEN_PHASES=1: host wall time of the phases on stderr (where a 2^20-constraint key's seconds go) */
        const bool phases = std::getenv("ZKHIP_GEN_PHASES") != nullptr;
        auto tp = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!phases) return;
            ctx.sync();
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "generator phase %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
            tp = now;
        };
        /* Make the B_query "lighter" if possible (generator.hpp:250-252) */
        pk.constraint_system = constraint_system;
        swap_AB_if_beneficial(pk.constraint_system);
        lap("copy + swap_AB");
        const Fr delta_inverse = delta.inversed();
        /* A quadratic arithmetic program evaluated at t. */
        const auto qap = instance_map_with_evaluation<CurveType>(pk.constraint_system, t, dom, &ctx);
        const std::size_t N = qap.num_variables, n = qap.num_inputs, m = qap.degree;
        lap("instance_map_with_evaluation");
        /* The delta inverse product component: (beta*A_i(t) + alpha*B_i(t) + C_i(t)) * delta^{-1} (generator.hpp:296-304) */
        std::vector<Fr> Lt(N - n);
        detail::parallel_chunks(N - n, [&](std::size_t lo, std::size_t hi) {
            for (std::size_t i = lo; i < hi; ++i) Lt[i] = (beta * qap.At[n + 1 + i] + alpha * qap.Bt[n + 1 + i] + qap.Ct[n + 1 + i]) * delta_inverse;
        });
        /* H for Groth's proof system is degree d - 2: Ht loses its top two entries (:310-315); coefficient Zt / delta (:353-355) */
        std::vector<Fr> Hs(m - 1);
        const Fr zd = qap.Zt * delta_inverse;
        detail::parallel_chunks(m - 1, [&](std::size_t lo, std::size_t hi) {
            for (std::size_t i = lo; i < hi; ++i) Hs[i] = qap.Ht[i] * zd;
        });
        /* B query: sparse over the non-zero B_i(t) (kc_batch_exp, knowledge_commitment_multiexp.hpp:143-208) */
        std::vector<Fr> Bnz;
        std::vector<std::uint32_t> b_indices;
        for (std::size_t i = 0; i <= N; ++i)
            if (!qap.Bt[i].is_zero()) {
                Bnz.push_back(qap.Bt[i]);
                b_indices.push_back((std::uint32_t)i);
            }
        pk.B_query.domain_size_ = N + 1;
        lap("Lt, Hs, B filter (host)");
        /* alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2 (:333-337): the same device path, read back */
        {
            std::vector<Fr> fx = {alpha, beta, delta};
            auto f1 = device_bases<CurveType, ZKHIP_G1>::from_scalars(ctx, fx.begin(), fx.end());
            auto f2 = device_bases<CurveType, ZKHIP_G2>::from_scalars(ctx, fx.begin(), fx.end());
            pk.alpha_g1 = f1.at(0);
            pk.beta_g1 = f1.at(1);
            pk.delta_g1 = f1.at(2);
            pk.beta_g2 = f2.at(1);
            pk.delta_g2 = f2.at(2);
        }
        /* the five batch exponentiations (:339-366), left resident as the key's queries.  rank / world > 1: this process
           generates (and will hold) only its slice of every query -- the point-range partition of a sharded proof. */
        const query_shard sh = query_shard::make(rank, world, N + 1, Bnz.size(), m - 1, N - n);
        lap("fixed elements");
        auto a_query = device_bases<CurveType, ZKHIP_G1>::from_scalars(ctx, qap.At.begin() + sh.A_lo, qap.At.begin() + sh.A_lo + sh.A_n);
        lap("A query");
        auto b_query_g = device_bases<CurveType, ZKHIP_G2>::from_scalars(ctx, Bnz.begin() + sh.B_lo, Bnz.begin() + sh.B_lo + sh.B_n);
        auto b_query_h = device_bases<CurveType, ZKHIP_G1>::from_scalars(ctx, Bnz.begin() + sh.B_lo, Bnz.begin() + sh.B_lo + sh.B_n);
        lap("B query (G2 + G1)");
        auto h_query = device_bases<CurveType, ZKHIP_G1>::from_scalars(ctx, Hs.begin() + sh.H_lo, Hs.begin() + sh.H_lo + sh.H_n);
        lap("H query");
        auto l_query = device_bases<CurveType, ZKHIP_G1>::from_scalars(ctx, Lt.begin() + sh.L_lo, Lt.begin() + sh.L_lo + sh.L_n);
        lap("L query");
        b_indices = std::vector<std::uint32_t>(b_indices.begin() + sh.B_lo, b_indices.begin() + sh.B_lo + sh.B_n);
        /* the device key reduces over the very domain the queries were evaluated on (explicit: a rank's H slice does not tell) */
        domain_params<CurveType> key_dom = dom;
        {
            const evaluation_domain_hip<CurveType> ed = evaluation_domain_hip<CurveType>::make(dom, pk.constraint_system.num_constraints() + n + 1);
            key_dom.kind = ed.kind;
            key_dom.m = ed.m;
        }
        key->device.reset(new r1cs_gg_ppzksnark_proving_key_hip<CurveType>(ctx, pk, key_dom, std::move(a_query), std::move(b_query_g), std::move(b_query_h),
                                                                          b_indices, std::move(h_query), std::move(l_query), world > 1 ? &sh : nullptr));
        lap("device key (r1cs upload)");
        return key;
    }

    /// The same over a device group: member k generates slice k of every query on its own GPU (the fixed-base batch exponentiations of
    /// generator.hpp:165-214 cut by point range), nothing crosses between the members.
    static std::unique_ptr<generated_proving_key_group<CurveType>> deterministic_basic_process(const device_group &group, const constraint_system_type &constraint_system,
                                                                                               const domain_params<CurveType> &dom, const scalar_value_type &t,
                                                                                               const scalar_value_type &alpha, const scalar_value_type &beta,
                                                                                               const scalar_value_type &gamma, const scalar_value_type &delta) {
        std::unique_ptr<generated_proving_key_group<CurveType>> out(new generated_proving_key_group<CurveType>());
        std::vector<std::shared_ptr<r1cs_gg_ppzksnark_proving_key_hip<CurveType>>> members;
        for (std::size_t k = 0; k < group.size(); ++k) {
            out->parts.push_back(deterministic_basic_process(group[k], constraint_system, dom, t, alpha, beta, gamma, delta, k, group.size()));
            /* the part owns its device key; the group key refers to it (`out` keeps both alive together) */
            members.emplace_back(out->parts.back()->device.get(), [](r1cs_gg_ppzksnark_proving_key_hip<CurveType> *) { });
        }
        out->device.reset(new r1cs_gg_ppzksnark_proving_key_group_hip<CurveType>(group, std::move(members)));
        return out;
    }

    /// generator.hpp:84-236: fresh toxic waste from the operating system's CSPRNG
    static std::unique_ptr<generated_proving_key<CurveType>> basic_process(const context &ctx, const constraint_system_type &constraint_system,
                                                                           const domain_params<CurveType> &dom) {
        typedef r1cs_gg_ppzksnark_prover_hip<CurveType> prover;
        return deterministic_basic_process(ctx, constraint_system, dom, prover::random_scalar(), prover::random_scalar(), prover::random_scalar(),
                                           prover::random_scalar(), prover::random_scalar());
    }

    /// r1cs_gg_ppzksnark_generator<CurveType>::process(constraint_system) (generator.hpp:84-86) with the reference's argument list: the
    /// calling thread's default context, the domain make_evaluation_domain(M + n + 1) returns (constants from the curve adapter)
    static std::unique_ptr<generated_proving_key<CurveType>> process(const constraint_system_type &constraint_system) {
        return basic_process(default_context(), constraint_system,
                             standard_domain_params<CurveType>(constraint_system.num_constraints() + constraint_system.num_inputs() + 1));
    }
    /// ... and the testing entry point (generator.hpp:240-247): the toxic waste as arguments
    static std::unique_ptr<generated_proving_key<CurveType>> deterministic_basic_process(const constraint_system_type &constraint_system,
                                                                                         const scalar_value_type &t, const scalar_value_type &alpha,
                                                                                         const scalar_value_type &beta, const scalar_value_type &gamma,
                                                                                         const scalar_value_type &delta) {
        return deterministic_basic_process(default_context(), constraint_system,
                                           standard_domain_params<CurveType>(constraint_system.num_constraints() + constraint_system.num_inputs() + 1), t,
                                           alpha, beta, gamma, delta);
    }

    /// r1cs.hpp:192-215
    static void swap_AB_if_beneficial(constraint_system_type &cs) {
        std::vector<bool> touched_by_A(cs.num_variables() + 1, false), touched_by_B(cs.num_variables() + 1, false);
        for (const auto &c : cs.constraints) {
            for (const auto &term : c.a.terms) touched_by_A[term.index] = true;
            for (const auto &term : c.b.terms) touched_by_B[term.index] = true;
        }
        std::size_t non_zero_A_count = 0, non_zero_B_count = 0;
        for (std::size_t i = 0; i < cs.num_variables() + 1; ++i) {
            non_zero_A_count += touched_by_A[i] ? 1 : 0;
            non_zero_B_count += touched_by_B[i] ? 1 : 0;
        }
        if (non_zero_B_count > non_zero_A_count)
            for (auto &c : cs.constraints) std::swap(c.a, c.b);
    }
};

/// The proof a correct prover outputs for (primary, auxiliary) under the key of trapdoor (t, alpha, beta, delta) with
/// blinders (r, s), as three EXPONENTS (prover.hpp:141,145,151-153 comment formulas):
///   A = a G1, B = b G2, C = c G1 with
///   a = alpha + sum z_i A_i(t) + r delta,   b = beta + sum z_i B_i(t) + s delta,
///   c = (sum_{i > n} z_i (beta A_i + alpha B_i + C_i)(t) + A(t) B(t) - C(t)) / delta + s a + r b - r s delta
/// (H(t) Z(t) = A(t) B(t) - C(t) for a satisfying assignment).  O(nnz + m) field operations and no group
/// arithmetic: what bench.py and the tests hold a full-size proof against.
template <typename CurveType, typename ConstraintSystem>
std::array<typename curve_adapter<CurveType>::scalar_value_type, 3> groth16_proof_exponents(
    const ConstraintSystem &swapped_cs, const domain_params<CurveType> &dom, const std::vector<typename curve_adapter<CurveType>::scalar_value_type> &primary_input,
    const std::vector<typename curve_adapter<CurveType>::scalar_value_type> &auxiliary_input, const typename curve_adapter<CurveType>::scalar_value_type &t,
    const typename curve_adapter<CurveType>::scalar_value_type &alpha, const typename curve_adapter<CurveType>::scalar_value_type &beta,
    const typename curve_adapter<CurveType>::scalar_value_type &delta, const typename curve_adapter<CurveType>::scalar_value_type &r,
    const typename curve_adapter<CurveType>::scalar_value_type &s) {
    typedef typename curve_adapter<CurveType>::scalar_value_type Fr;
    const auto qap = instance_map_with_evaluation<CurveType>(swapped_cs, t, dom);
    const std::size_t N = qap.num_variables, n = qap.num_inputs;
    std::vector<Fr> z;
    z.push_back(Fr::one());
    z.insert(z.end(), primary_input.begin(), primary_input.end());
    z.insert(z.end(), auxiliary_input.begin(), auxiliary_input.end());
    if (z.size() != N + 1) throw std::invalid_argument("groth16_proof_exponents: assignment size");
    Fr at = Fr::zero(), bt = Fr::zero(), ct = Fr::zero(), lt = Fr::zero();
    for (std::size_t i = 0; i <= N; ++i) {
        at = at + z[i] * qap.At[i];
        bt = bt + z[i] * qap.Bt[i];
        ct = ct + z[i] * qap.Ct[i];
        if (i > n) lt = lt + z[i] * (beta * qap.At[i] + alpha * qap.Bt[i] + qap.Ct[i]);
    }
    const Fr a = alpha + at + r * delta, b = beta + bt + s * delta;
    const Fr c = (lt + at * bt - ct) * delta.inversed() + s * a + r * b - r * s * delta;
    return {a, b, c};
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_R1CS_GG_PPZKSNARK_GENERATOR_HPP
