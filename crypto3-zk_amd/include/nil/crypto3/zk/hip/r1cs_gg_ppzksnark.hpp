//---------------------------------------------------------------------------//
// zkhip shim: Groth16 prover with the reference's interface, MSM and NTT on the MI355X.
//
// Mirrors (same member names, same argument meaning):
//   r1cs_constraint_system / r1cs_constraint / linear_combination
//        zk/snark/arithmetization/constraint_satisfaction_problems/r1cs.hpp:61-64, 125-133
//   r1cs_gg_ppzksnark_proving_key         .../r1cs_gg_ppzksnark/proving_key.hpp:43-56
//   r1cs_gg_ppzksnark_proof               .../r1cs_gg_ppzksnark/proof.hpp:41-46
//   reductions::r1cs_to_qap::witness_map  zk/snark/reductions/r1cs_to_qap.hpp:219-325
//   r1cs_gg_ppzksnark_prover::process     .../r1cs_gg_ppzksnark/prover.hpp:73-158
// `r1cs_gg_ppzksnark_prover_hip<CurveType>::process(pk, primary_input, auxiliary_input)` has the reference's
// static signature; the facade `r1cs_gg_ppzksnark<...>` insists on the exact reference prover type
// (r1cs_gg_ppzksnark.hpp:50-59, 112-115), so the sibling class is called directly.
// The overload taking (r, s) exists for TESTS (reproducible proofs); the three-argument form draws the blinders from
// the operating system's CSPRNG, as prover.hpp:92-93 draws them with algebra::random_element.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_R1CS_GG_PPZKSNARK_HPP
#define ZKHIP_SHIM_R1CS_GG_PPZKSNARK_HPP

#include <algorithm>
#include <chrono>
#include <exception>
#include <future>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <map>

#include "evaluation_domain.hpp"
#include "multiexp.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

// ---- arithmetization (r1cs.hpp) --------------------------------------------------------------------------
template <typename CurveType>
struct linear_term {
    std::size_t index;    // 0 = the constant one, j >= 1 = variable j - 1
    typename curve_adapter<CurveType>::scalar_value_type coeff;
};

template <typename CurveType>
struct linear_combination {
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    std::vector<linear_term<CurveType>> terms;
    void add_term(std::size_t index, const value_type &coeff) { terms.push_back({index, coeff}); }
    void add_term(std::size_t index, std::uint64_t coeff) { terms.push_back({index, value_type(coeff)}); }
    /// sum_j coeff_j * assignment[index_j - 1]  (r1cs.hpp `evaluate`; host reference path, used by is_satisfied)
    value_type evaluate(const std::vector<value_type> &assignment) const {
        value_type acc = value_type::zero();
        for (const auto &t : terms) acc = acc + t.coeff * (t.index == 0 ? value_type::one() : assignment[t.index - 1]);
        return acc;
    }
};

template <typename CurveType>
struct r1cs_constraint {
    linear_combination<CurveType> a, b, c;
};

template <typename CurveType>
struct r1cs_constraint_system {
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    std::size_t primary_input_size = 0;
    std::size_t auxiliary_input_size = 0;
    std::vector<r1cs_constraint<CurveType>> constraints;

    std::size_t num_inputs() const { return primary_input_size; }
    std::size_t num_variables() const { return primary_input_size + auxiliary_input_size; }
    std::size_t num_constraints() const { return constraints.size(); }
    void add_constraint(const r1cs_constraint<CurveType> &c) { constraints.emplace_back(c); }
    bool is_satisfied(const std::vector<value_type> &primary_input, const std::vector<value_type> &auxiliary_input) const {
        std::vector<value_type> full(primary_input);
        full.insert(full.end(), auxiliary_input.begin(), auxiliary_input.end());
        for (const auto &c : constraints)
            if (c.a.evaluate(full) * c.b.evaluate(full) != c.c.evaluate(full)) return false;
        return true;
    }
};

// ---- proving key / proof (proving_key.hpp, proof.hpp, knowledge_commitment.hpp) ---------------------------
template <typename CurveType>
struct element_kc {    // element_knowledge_commitment.hpp:54-59
    typename curve_adapter<CurveType>::g2_value_type g;
    typename curve_adapter<CurveType>::g1_value_type h;
};

template <typename CurveType>
struct knowledge_commitment_vector {    // container::sparse_vector of (g, h) pairs
    std::vector<std::size_t> indices;
    std::vector<element_kc<CurveType>> values;
    std::size_t domain_size_ = 0;
};

template <typename CurveType>
struct r1cs_gg_ppzksnark_proving_key {
    typedef curve_adapter<CurveType> adapter;
    typename adapter::g1_value_type alpha_g1, beta_g1;
    typename adapter::g2_value_type beta_g2;
    typename adapter::g1_value_type delta_g1;
    typename adapter::g2_value_type delta_g2;
    std::vector<typename adapter::g1_value_type> A_query;
    knowledge_commitment_vector<CurveType> B_query;
    std::vector<typename adapter::g1_value_type> H_query;
    std::vector<typename adapter::g1_value_type> L_query;
    r1cs_constraint_system<CurveType> constraint_system;
};

template <typename CurveType>
struct r1cs_gg_ppzksnark_proof {
    typename curve_adapter<CurveType>::g1_value_type g_A;
    typename curve_adapter<CurveType>::g2_value_type g_B;
    typename curve_adapter<CurveType>::g1_value_type g_C;
};

// ---- device-resident constraint system -------------------------------------------------------------------
template <typename CurveType>
class device_r1cs {
public:
    typedef curve_adapter<CurveType> adapter;
    /// `ConstraintSystem` is duck-typed on the reference's member names (r1cs.hpp:61-64, 125-133): `constraints[i].{a,b,c}.terms[j].
    /// {index, coeff}`, `num_constraints()`, `num_inputs()`, `num_variables()` -- the reference's own r1cs_constraint_system
    /// is consumed as it is (coefficients go through curve_adapter::scalar_to_limbs), no look-alike copy.
    template <typename ConstraintSystem>
    device_r1cs(const context &ctx, const ConstraintSystem &cs) : ctx_(&ctx), min_size_(cs.num_constraints() + cs.num_inputs() + 1) {
        std::vector<std::uint32_t> rp[3], cl[3];
        std::vector<std::uint64_t> cf[3];
        for (int k = 0; k < 3; ++k) rp[k].push_back(0);
        for (const auto &c : cs.constraints) {
            const decltype(c.a) *lc[3] = {&c.a, &c.b, &c.c};
            for (int k = 0; k < 3; ++k) {
                for (const auto &t : lc[k]->terms) {
                    cl[k].push_back((std::uint32_t)t.index);
                    cf[k].resize(cf[k].size() + 4);
                    adapter::scalar_to_limbs(t.coeff, cf[k].data() + cf[k].size() - 4);
                }
                rp[k].push_back((std::uint32_t)cl[k].size());
            }
        }
        check(zkhip_r1cs_upload(ctx.get(), adapter::id, cs.num_constraints(), cs.num_inputs(), cs.num_variables(), rp[0].data(), cl[0].data(),
                                cf[0].data(), rp[1].data(), cl[1].data(), cf[1].data(), rp[2].data(), cl[2].data(), cf[2].data(), &r_),
              "zkhip_r1cs_upload", ctx.get());
    }
    /// a second handle on the same resident matrices (read-only after construction and set_domain), for another prover lane
    struct alias_tag { };
    device_r1cs(alias_tag, const device_r1cs &o) : ctx_(o.ctx_), min_size_(o.min_size_), r_(o.r_), owner_(false), aliases_(o.aliases_) { ++*aliases_; }
    ~device_r1cs() {
        if (owner_) zkhip_r1cs_free(ctx_->get(), r_);
    }
    device_r1cs(const device_r1cs &) = delete;
    device_r1cs &operator=(const device_r1cs &) = delete;
    const zkhip_r1cs *get() const { return r_; }
    /// the evaluation domain the witness map reduces over: after construction what make_evaluation_domain(M + n + 1) picks
    /// (r1cs_to_qap.hpp:229-230), until set_domain installs another (e.g. the basic one a key was generated over)
    std::size_t domain_size() const { return zkhip_r1cs_domain_size(r_); }
    int domain_kind() const { return zkhip_r1cs_domain_kind(r_); }
    /// The domain lives in the resident object every alias handle shares: installing the domain it already has is a no-op, and a
    /// CHANGE is refused once another lane holds an alias (it may be reading kind / n0 / n1 in the middle of a proof).
    void set_domain(int kind, std::size_t m) {
        if (domain_kind() == kind && domain_size() == m) return;
        if (*aliases_ != 0) throw std::logic_error("device_r1cs::set_domain: the constraint system is shared with other prover lanes");
        check(zkhip_r1cs_set_domain(r_, kind, m), "zkhip_r1cs_set_domain", ctx_->get());
    }
    void set_domain(const evaluation_domain_hip<CurveType> &d) { set_domain(d.kind, d.m); }
    std::size_t min_domain_size() const { return min_size_; }

private:
    const context *ctx_;
    std::size_t min_size_ = 0;
    zkhip_r1cs *r_ = nullptr;
    bool owner_ = true;
    std::shared_ptr<std::size_t> aliases_ = std::make_shared<std::size_t>(0);    // alias handles ever taken on r_ (shared with them)
};

// ---- reductions::r1cs_to_qap<F>::witness_map on the device ---------------------------------------------------
template <typename CurveType>
struct r1cs_to_qap_hip {
    typedef curve_adapter<CurveType> adapter;
    typedef typename adapter::scalar_value_type value_type;

    /// coefficients_for_H (m + 1 elements) left RESIDENT on the device; `d_assignment` receives (1, x, w).
    static std::shared_ptr<void> witness_map(const context &ctx, device_r1cs<CurveType> &cs, const domain_params<CurveType> &dom,
                                             const std::vector<value_type> &primary_input, const std::vector<value_type> &auxiliary_input,
                                             std::shared_ptr<void> &d_assignment) {
        /* make_evaluation_domain(cs.num_constraints() + cs.num_inputs() + 1) (r1cs_to_qap.hpp:229-230), or the domain `dom` names */
        const evaluation_domain_hip<CurveType> domain = evaluation_domain_hip<CurveType>::make(dom, cs.min_domain_size());
        cs.set_domain(domain);
        const std::size_t N = primary_input.size() + auxiliary_input.size(), m = domain.m;
        std::vector<std::uint64_t> z(4 * (N + 1), 0);
        z[0] = 1;
        std::size_t k = 1;
        for (const auto &v : primary_input) adapter::scalar_to_limbs(v, &z[4 * k++]);
        for (const auto &v : auxiliary_input) adapter::scalar_to_limbs(v, &z[4 * k++]);
        d_assignment = ctx.alloc(z.size() * 8);
        ctx.h2d(d_assignment.get(), z.data(), z.size() * 8);
        auto d_h = ctx.alloc((m + 1) * 32);
        auto d_scratch = ctx.alloc(zkhip_groth16_scratch_bytes(cs.get()));
        std::uint64_t g[4];
        const zkhip_domain dd = domain.c_desc();
        adapter::scalar_to_limbs(dom.coset_generator, g);
        check(zkhip_groth16_witness_h_domain_dev(ctx.get(), cs.get(), d_assignment.get(), &dd, g, d_h.get(), d_scratch.get()),
              "zkhip_groth16_witness_h_domain_dev", ctx.get());
        ctx.sync();    // d_scratch is released on return
        return d_h;
    }
    /// r1cs_to_qap<F>::witness_map(cs, primary_input, auxiliary_input, d1 = d2 = d3 = 0) (r1cs_to_qap.hpp:219-225) with the reference's
    /// argument list: the calling thread's default context, the domain make_evaluation_domain(M + n + 1) returns; -> coefficients_for_H
    template <typename ConstraintSystem>
    static std::vector<value_type> witness_map(const ConstraintSystem &cs, const std::vector<value_type> &primary_input,
                                               const std::vector<value_type> &auxiliary_input) {
        const context &ctx = default_context();
        device_r1cs<CurveType> dcs(ctx, cs);
        return witness_map_host(ctx, dcs, standard_domain_params<CurveType>(cs.num_constraints() + cs.num_inputs() + 1), primary_input, auxiliary_input);
    }
    /// host copy of the same (the reference's return value, qap_witness::coefficients_for_H)
    static std::vector<value_type> witness_map_host(const context &ctx, device_r1cs<CurveType> &cs, const domain_params<CurveType> &dom,
                                                    const std::vector<value_type> &primary_input,
                                                    const std::vector<value_type> &auxiliary_input) {
        std::shared_ptr<void> d_z;
        auto d_h = witness_map(ctx, cs, dom, primary_input, auxiliary_input, d_z);
        const std::size_t m = cs.domain_size();
        std::vector<value_type> out;
        download_scalars<adapter>(ctx, d_h.get(), m + 1, out);
        return out;
    }
};

/// Contiguous, balanced split of [0, n) over `world` ranks (the first n % world ranks get one extra element).
inline std::pair<std::size_t, std::size_t> shard_range(std::size_t n, std::size_t rank, std::size_t world) {
    const std::size_t base = n / world, extra = n % world, lo = rank * base + std::min(rank, extra);
    return {lo, lo + base + (rank < extra ? 1 : 0)};
}

/// Which slice of every query a proving-key object holds when a proof is sharded over several GPUs (SURVEY 8e:
/// point-range partition, one process per GPU).  Offsets are positions in the FULL query; *_n the slice length.
/// B counts entries of the sparse query (its index list is sliced, not the variable range).
struct query_shard {
    std::size_t rank = 0, world = 1;
    std::size_t A_lo = 0, A_n = 0, B_lo = 0, B_n = 0, H_lo = 0, H_n = 0, L_lo = 0, L_n = 0;
    /// slices for `rank` of `world` given the sizes a proof uses: A: N + 1, B: entries, H: degree - 1, L: N - n
    static query_shard make(std::size_t rank, std::size_t world, std::size_t a, std::size_t b, std::size_t h, std::size_t l) {
        query_shard q;
        q.rank = rank;
        q.world = world;
        auto ra = shard_range(a, rank, world), rb = shard_range(b, rank, world), rh = shard_range(h, rank, world), rl = shard_range(l, rank, world);
        q.A_lo = ra.first, q.A_n = ra.second - ra.first;
        q.B_lo = rb.first, q.B_n = rb.second - rb.first;
        q.H_lo = rh.first, q.H_n = rh.second - rh.first;
        q.L_lo = rl.first, q.L_n = rl.second - rl.first;
        return q;
    }
};

// ---- proving key with its device-resident queries -------------------------------------------------------------
/// `KeyType` is duck-typed on the reference's member names (proving_key.hpp:43-56): alpha_g1, beta_g1, beta_g2, delta_g1,
/// delta_g2, A_query, B_query.{indices, values[i].{g, h}, domain_size_}, H_query, L_query, constraint_system -- a key
/// object of the reference is consumed where it lies (its group / field values go through curve_adapter), the shim's
/// own r1cs_gg_ppzksnark_proving_key is just the default.
template <typename CurveType, typename KeyType = r1cs_gg_ppzksnark_proving_key<CurveType>>
class r1cs_gg_ppzksnark_proving_key_hip {
public:
    typedef curve_adapter<CurveType> adapter;
    typedef KeyType host_key_type;

    /// Rank `rank` of `world`: uploads only this rank's slice of the four queries (and the whole constraint system:
    /// the witness map is replicated).  Proofs then go through prover::process_partial + an all-gather of the partial
    /// sums + prover::finish.
    r1cs_gg_ppzksnark_proving_key_hip(const context &ctx, const host_key_type &pk, const domain_params<CurveType> &dom, std::size_t rank,
                                      std::size_t world) :
        ctx(ctx), host(pk), domain(dom), constraint_system(ctx, pk.constraint_system) {
        resolve_domain(pk.H_query.size());
        check_host_key(pk);
        const std::size_t N = pk.constraint_system.num_variables(), n = pk.constraint_system.num_inputs();
        shard = query_shard::make(rank, world, N + 1, pk.B_query.values.size(), constraint_system.domain_size() - 1, N - n);
        A_query = device_bases<CurveType, ZKHIP_G1>(ctx, pk.A_query.begin() + shard.A_lo, pk.A_query.begin() + shard.A_lo + shard.A_n);
        H_query = device_bases<CurveType, ZKHIP_G1>(ctx, pk.H_query.begin() + shard.H_lo, pk.H_query.begin() + shard.H_lo + shard.H_n);
        L_query = device_bases<CurveType, ZKHIP_G1>(ctx, pk.L_query.begin() + shard.L_lo, pk.L_query.begin() + shard.L_lo + shard.L_n);
        upload_b_query(pk, shard.B_lo, shard.B_n);
    }

    /// Uploads the four queries and the constraint system once; proofs then only move the assignment.
    r1cs_gg_ppzksnark_proving_key_hip(const context &ctx, const host_key_type &pk, const domain_params<CurveType> &dom) :
        r1cs_gg_ppzksnark_proving_key_hip(ctx, pk, dom, 0, 1) {}
    /// The reference's own argument list -- a proving key, nothing else: the calling thread's default context (multiexp.hpp) and the
    /// domain make_evaluation_domain(M + n + 1) returns, its constants from the curve adapter (standard_domain_params).  With it
    ///     r1cs_gg_ppzksnark_proving_key_hip<curve> dpk(pk);  proof = r1cs_gg_ppzksnark_prover_hip<curve>::process(dpk, x, w);
    /// is the whole change at a call site of r1cs_gg_ppzksnark_prover::process(pk, x, w) (prover.hpp:73-75).
    explicit r1cs_gg_ppzksnark_proving_key_hip(const host_key_type &pk) :
        r1cs_gg_ppzksnark_proving_key_hip(default_context(), pk,
                                          standard_domain_params<CurveType>(pk.constraint_system.num_constraints() + pk.constraint_system.num_inputs() + 1), 0,
                                          1) {}

    /// Adopts queries that already live on the device (e.g. produced by device_bases::from_scalars or decoded from the
    /// wire form); `pk` supplies the five single group elements and the constraint system, its query vectors are not read.
    r1cs_gg_ppzksnark_proving_key_hip(const context &ctx, const host_key_type &pk, const domain_params<CurveType> &dom,
                                      device_bases<CurveType, ZKHIP_G1> &&a_query, device_bases<CurveType, ZKHIP_G2> &&b_query_g,
                                      device_bases<CurveType, ZKHIP_G1> &&b_query_h, const std::vector<std::uint32_t> &b_indices,
                                      device_bases<CurveType, ZKHIP_G1> &&h_query, device_bases<CurveType, ZKHIP_G1> &&l_query,
                                      const query_shard *slice = nullptr) :
        ctx(ctx), host(pk), domain(dom), A_query(std::move(a_query)), H_query(std::move(h_query)), L_query(std::move(l_query)),
        B_query_h(std::move(b_query_h)), B_query_g(std::move(b_query_g)), constraint_system(ctx, pk.constraint_system) {
        const std::size_t N = pk.constraint_system.num_variables(), n = pk.constraint_system.num_inputs();
        resolve_domain(slice ? 0 : H_query.size());
        check_b_indices(b_indices.begin(), b_indices.end(), N);
        if (B_query_g.size() != b_indices.size() || B_query_h.size() != b_indices.size())
            throw std::invalid_argument("proving key: the B query's index list and its (g, h) values differ in length");
        d_B_indices = ctx.alloc(std::max<std::size_t>(1, b_indices.size()) * 4);
        if (!b_indices.empty()) ctx.h2d(d_B_indices.get(), b_indices.data(), b_indices.size() * 4);
        B_count = b_indices.size();
        if (slice) shard = *slice;    // the adopted bases are this rank's slices
        else shard = query_shard::make(0, 1, N + 1, B_count, constraint_system.domain_size() - 1, N - n);
        if (A_query.size() != shard.A_n || H_query.size() != shard.H_n || L_query.size() != shard.L_n) throw_query_sizes();
    }

    /// A second prover LANE over the same key: the queries and the constraint system of `other` are shared where they lie in HBM
    /// (nothing is copied; they are read-only), the lane has its own context / stream, work buffers and G2 stream.  Two host threads,
    /// one proving over `other` and one over the lane, fill each other's latency-bound phases: 2^20 constraints, one MI355X:
    /// 51-52 proofs/s against 44-45 one at a time (tools/groth16_two_provers.py).  `other` must outlive the lane and must not be
    /// proving while the lane is constructed.
    r1cs_gg_ppzksnark_proving_key_hip(const context &lane_ctx, const r1cs_gg_ppzksnark_proving_key_hip &other) :
        ctx(lane_ctx), host(other.host), domain(other.domain), evaluation_domain(other.evaluation_domain),
        A_query((other.align_rows(), device_bases<CurveType, ZKHIP_G1>::alias(other.A_query))),
        H_query(device_bases<CurveType, ZKHIP_G1>::alias(other.H_query)), L_query(device_bases<CurveType, ZKHIP_G1>::alias(other.L_query)),
        B_query_h(device_bases<CurveType, ZKHIP_G1>::alias(other.B_query_h)), B_query_g(device_bases<CurveType, ZKHIP_G2>::alias(other.B_query_g)),
        constraint_system(typename device_r1cs<CurveType>::alias_tag(), other.constraint_system), d_B_indices(other.d_B_indices),
        B_count(other.B_count), shard(other.shard), share_sorts(other.share_sorts), L_rows_aligned(other.L_rows_aligned),
        B_rows_aligned(other.B_rows_aligned), overlap_g2(other.overlap_g2), side_stream_priority(other.side_stream_priority),
        direct_assignment_upload(other.direct_assignment_upload) {
        if (lane_ctx.device() != other.ctx.device()) throw std::invalid_argument("proving key lane: the lane's context is on another GPU than the key");
        other.lanes_taken_ = true;    // from here on `other` keeps its queries where they are (align_rows)
    }

    const context &ctx;
    const host_key_type &host;
    domain_params<CurveType> domain;
    /// the domain the key was generated over and proofs reduce over: `domain` resolved against the constraint system (and, for
    /// kind "auto", against the key's H_query size: make_evaluation_domain's choice, or the basic domain of the next power of two)
    evaluation_domain_hip<CurveType> evaluation_domain;
    device_bases<CurveType, ZKHIP_G1> A_query, H_query;
    /// L_query and the G1 half of the B query: as uploaded (N - n entries / one per index of the sparse query) until the first proof
    /// of an unsharded key lays them out over A_query's rows (share_sorts below)
    mutable device_bases<CurveType, ZKHIP_G1> L_query, B_query_h;
    device_bases<CurveType, ZKHIP_G2> B_query_g;
    device_r1cs<CurveType> constraint_system;
    std::shared_ptr<void> d_B_indices;
    std::size_t B_count = 0;
    query_shard shard;

    /// per-proof device buffers, kept across proofs: (1, x, w), coefficients_for_H, witness-map scratch, gathered B
    /// scalars, the five Jacobian MSM results
    mutable std::shared_ptr<void> d_cpa, d_h, d_scratch, d_bs, d_results;
    mutable std::shared_ptr<void> h_cpa;    // page-locked staging for (1, x, w): H2D at link speed, asynchronous
    /// evaluation_At, evaluation_Bt.h and evaluation_Lt multiply by the SAME assignment vector (prover.hpp:108-139): with the three
    /// queries laid out over the same N + 1 rows (L_query behind n + 1 points at infinity; the B query's G1 half spread over its index
    /// list when all but 64 or at least 63 of 64 variables occur in it -- the padding is gathered, too) the device extracts and sorts the window
    /// digits of the assignment once instead of three times (zkhip_bases_spread, "msm_share_sort").  Decided at the first proof.
    /// Measured on a 2^20-constraint proof: the main stream's sort kernels 4.1 -> 1.8 ms, the proof itself 22.3 -> 22.2 ms (the sorts
    /// it removes used to run under the G2 accumulation's multiply-adds), two provers sharing the GPU 51.3 -> 51.8 proofs/s.
    bool share_sorts = true;
    mutable bool L_rows_aligned = false, B_rows_aligned = false;
    mutable bool lanes_taken_ = false;    // a lane key aliases this key's queries
    /// second in-order stream on the same GPU for the G2 multiexp (its own workspace); false: everything on `ctx`
    bool overlap_g2 = true;
    /// scheduling priority of that second stream (< 0: above, > 0: below the main stream's; "stream_priority" in include/zkhip.h)
    bool experiment_skip_g2 = false;    // timing experiments only: see enqueue()
    int side_stream_priority = 1;    // measured: 22.7 -> 22.4 ms per 2^20 proof with the G2 stream below the main one (tools/exp_stream_priority.sh)
    /// send the auxiliary input as it lies in the caller's vector (possible when the scalar type is canonical limbs in memory);
    /// false: through the page-locked staging buffer, converted by host threads (any scalar representation)
    bool direct_assignment_upload = detail::canonical_scalars<curve_adapter<CurveType>>::value;
    /// host wall time of the last proof's phases, ms: staging + launches | host products (device busy) | waiting for the device | assembly
    mutable double last_phase_ms[4] = {0, 0, 0, 0};
    mutable std::unique_ptr<context> side;
    mutable std::int64_t saved_sort_tile_log = 14;    // the caller's "msm_sort_tile_log", set aside for the duration of a proof
    /// lays L_query and (when dense) the B query's G1 half out over A_query's rows -- see share_sorts; once, before the first proof
    void align_rows() const {
        /* a lane aliases L_query / B_query_h by their device pointers: once one exists the layout is frozen -- a share_sorts switched
           on afterwards must not move (and free) what the lanes read */
        if (!share_sorts || shard.world != 1 || L_rows_aligned || lanes_taken_) return;
        const std::size_t N = host.constraint_system.num_variables(), n = host.constraint_system.num_inputs();
        if (L_query.size() != N - n || A_query.size() != N + 1) return;
        L_query = L_query.spread(nullptr, n + 1, N + 1);
        L_rows_aligned = true;
        if (B_count == B_query_h.size() && (B_count + 64 >= N + 1 || B_count * 64 >= (N + 1) * 63)) {
            B_query_h = B_query_h.spread(d_B_indices.get(), 0, N + 1);
            B_rows_aligned = true;
        }
    }
    void reserve_work(std::size_t cpa_elems, std::size_t degree, std::size_t result_bytes) const {
        if (d_cpa && work_cpa_ >= cpa_elems) return;
        align_rows();
        d_cpa = ctx.alloc(cpa_elems * 32);
        {
            void *hp = nullptr;
            check(zkhip_host_alloc(ctx.get(), cpa_elems * 32, &hp), "zkhip_host_alloc", ctx.get());
            zkhip_ctx *c = ctx.get();
            h_cpa = std::shared_ptr<void>(hp, [c](void *p) { zkhip_host_free(c, p); });
        }
        d_h = ctx.alloc((degree + 1) * 32);
        d_scratch = ctx.alloc(zkhip_groth16_scratch_bytes(constraint_system.get()));
        d_bs = ctx.alloc(std::max<std::size_t>(1, B_count) * 32);
        d_results = ctx.alloc(result_bytes);
        if (overlap_g2 && !side && B_count >= ((std::size_t)1 << 14)) {    // pays off for large queries only
            side.reset(new context(ctx.device()));
            /* while two streams share the GPU the sorts use tiles small enough to sit next to the other stream's resident accumulation
               workgroups ("msm_sort_tile_log" in include/zkhip.h): set on the side context for good, on the CALLER's context only for
               the duration of a proof (sort_tiles_scope in enqueue / collect) */
            side->set_option("msm_sort_tile_log", 12);
            if (side_stream_priority != 0) side->set_option("stream_priority", side_stream_priority);
        }
        work_cpa_ = cpa_elems;
    }

private:
    /// `h_entries`: the size of the FULL H query when known (0: a slice of it -- then `domain` decides alone)
    void resolve_domain(std::size_t h_entries) {
        typedef evaluation_domain_hip<CurveType> ED;
        const std::size_t min_size = constraint_system.min_domain_size();
        if (domain.kind >= 0) {
            evaluation_domain = ED::make(domain, min_size);
        } else {
            const auto c = ED::choice(min_size);
            const std::size_t pow2 = (std::size_t)1 << detail::ceil_log2(min_size);
            if (h_entries != 0 && h_entries + 1 != c.second && h_entries + 1 == pow2)
                evaluation_domain = ED(ZKHIP_DOMAIN_BASIC_RADIX2, pow2, domain.omega);    // a key generated over the basic domain
            else
                evaluation_domain = ED(c.first, c.second, domain.omega, domain.shift);
        }
        constraint_system.set_domain(evaluation_domain);
    }
    /// The key must have been generated over the evaluation domain proofs will reduce over (m points): H_query has m - 1
    /// entries, A_query N + 1, L_query N - n, and the sparse B query's indices are strictly increasing and <= N.
    void check_host_key(const host_key_type &pk) const {
        const std::size_t N = pk.constraint_system.num_variables(), n = pk.constraint_system.num_inputs(), m = constraint_system.domain_size();
        if (pk.H_query.size() != m - 1)
            throw std::invalid_argument("proving key: H_query has " + std::to_string(pk.H_query.size()) + " entries, the evaluation domain of " +
                                        std::to_string(m) + " points needs " + std::to_string(m - 1));
        if (pk.A_query.size() != N + 1 || pk.L_query.size() != N - n) throw_query_sizes();
        if (pk.B_query.indices.size() != pk.B_query.values.size())
            throw std::invalid_argument("proving key: the B query's index list and its values differ in length");
        check_b_indices(pk.B_query.indices.begin(), pk.B_query.indices.end(), N);
    }
    template <typename It>
    static void check_b_indices(It first, It last, std::size_t N) {
        bool have = false;
        std::size_t prev = 0;
        for (It it = first; it != last; ++it) {
            const std::size_t i = (std::size_t)*it;
            if (i > N || (have && i <= prev)) throw std::invalid_argument("proving key: B query indices must be strictly increasing and <= num_variables");
            prev = i;
            have = true;
        }
    }
    [[noreturn]] static void throw_query_sizes() {
        throw std::invalid_argument("proving key: query sizes do not match the constraint system (A: N + 1, H: m - 1, L: N - n entries, or the slice of them)");
    }
    void upload_b_query(const host_key_type &pk, std::size_t lo, std::size_t cnt) {
        std::vector<typename adapter::g2_value_type> g;
        std::vector<typename adapter::g1_value_type> h;
        std::vector<std::uint32_t> idx;
        for (std::size_t i = lo; i < lo + cnt; ++i) {
            g.push_back(pk.B_query.values[i].g);
            h.push_back(pk.B_query.values[i].h);
            idx.push_back((std::uint32_t)pk.B_query.indices[i]);
        }
        B_query_g = device_bases<CurveType, ZKHIP_G2>(ctx, g.begin(), g.end());
        B_query_h = device_bases<CurveType, ZKHIP_G1>(ctx, h.begin(), h.end());
        d_B_indices = ctx.alloc(std::max<std::size_t>(1, idx.size()) * 4);
        if (!idx.empty()) ctx.h2d(d_B_indices.get(), idx.data(), idx.size() * 4);
        B_count = idx.size();
    }
    mutable std::size_t work_cpa_ = 0;
};

// ---- the same key over a device group ---------------------------------------------------------------------------------
/// One proving key over the N GPUs of a device_group: member k holds the k-th point-range slice of every query and the
/// whole constraint system (the witness map is replicated: 7 transforms, no exchange).  r1cs_gg_ppzksnark_prover_hip::process(group_key,
/// primary_input, auxiliary_input) then proves over all of them from the ONE calling thread and does the exchange -- one all-gather of
/// partial_limbs() u64 per member, 864 bytes for BLS12-381 -- inside the library: what `chunks = omp_get_max_threads()` is inside the
/// reference's process (prover.hpp:94-99, 108-139).  Same proof as over one device, bit for bit (the group law is exact; only the order
/// of the additions differs).
template <typename CurveType, typename KeyType = r1cs_gg_ppzksnark_proving_key<CurveType>>
class r1cs_gg_ppzksnark_proving_key_group_hip {
public:
    typedef r1cs_gg_ppzksnark_proving_key_hip<CurveType, KeyType> member_key_type;
    typedef KeyType host_key_type;

    r1cs_gg_ppzksnark_proving_key_group_hip(const device_group &group, const host_key_type &pk, const domain_params<CurveType> &dom) : group(group) {
        for (std::size_t k = 0; k < group.size(); ++k) members.emplace_back(new member_key_type(group[k], pk, dom, k, group.size()));
    }
    /// adopts member keys that exist already -- generated on the devices (r1cs_gg_ppzksnark_generator_hip over a group), decoded from the
    /// wire form --: keys[k] lives on group[k] and holds slice k of `group.size()`
    r1cs_gg_ppzksnark_proving_key_group_hip(const device_group &group, std::vector<std::shared_ptr<member_key_type>> keys) :
        group(group), members(std::move(keys)) {
        if (members.size() != group.size()) throw std::invalid_argument("proving key group: one member key per device of the group");
        for (std::size_t k = 0; k < members.size(); ++k)
            if (!members[k] || members[k]->ctx.get() != group[k].get() || members[k]->shard.rank != k || members[k]->shard.world != members.size())
                throw std::invalid_argument("proving key group: member key " + std::to_string(k) + " is not slice k of the group on the group's context k");
    }
    /// the reference's argument list plus the group: the domain make_evaluation_domain(M + n + 1) returns, constants from the curve adapter
    r1cs_gg_ppzksnark_proving_key_group_hip(const device_group &group, const host_key_type &pk) :
        r1cs_gg_ppzksnark_proving_key_group_hip(
            group, pk, standard_domain_params<CurveType>(pk.constraint_system.num_constraints() + pk.constraint_system.num_inputs() + 1)) { }

    const device_group &group;
    std::vector<std::shared_ptr<member_key_type>> members;
    /// member 0's receive buffer of the all-gather (world x partial sums), allocated at the first proof
    mutable std::shared_ptr<void> d_all;
    /// host wall time of the last proof's phases, ms: staging + launches on all members | host products | exchange + waiting | assembly
    mutable double last_phase_ms[4] = {0, 0, 0, 0};
};

// ---- r1cs_gg_ppzksnark_prover<CurveType, basic>::process -------------------------------------------------------
/// `KeyType` / `ProofType`: the host-side key the device key was built from and the proof type to return (the
/// reference's r1cs_gg_ppzksnark_proving_key / r1cs_gg_ppzksnark_proof, proof.hpp:41-46: constructible from (g_A, g_B, g_C)).
template <typename CurveType, typename KeyType = r1cs_gg_ppzksnark_proving_key<CurveType>, typename ProofType = r1cs_gg_ppzksnark_proof<CurveType>>
class r1cs_gg_ppzksnark_prover_hip {
    typedef curve_adapter<CurveType> adapter;

public:
    typedef typename adapter::scalar_value_type scalar_value_type;
    typedef std::vector<scalar_value_type> primary_input_type;
    typedef std::vector<scalar_value_type> auxiliary_input_type;
    typedef r1cs_gg_ppzksnark_proving_key_hip<CurveType, KeyType> proving_key_type;
    typedef ProofType proof_type;

    static proof_type process(const proving_key_type &proving_key, const primary_input_type &primary_input,
                              const auxiliary_input_type &auxiliary_input) {
        /* Choose two random field elements for prover zero-knowledge (prover.hpp:92-93). */
        return process(proving_key, primary_input, auxiliary_input, random_scalar(), random_scalar());
    }

    /// THE REFERENCE'S STATIC SIGNATURE, argument for argument (prover.hpp:73-75):
    ///     proof_type process(const proving_key_type &proving_key, const primary_input_type &, const auxiliary_input_type &)
    /// with `proving_key` the reference's HOST key.  The device key the proof needs (queries in HBM, window tables) is built on the
    /// first call with a given key object and kept in a per-thread cache for the later ones (keyed by the key's address and query
    /// sizes; `forget(proving_key)` drops it when the host key goes away), on the calling thread's default context and over the
    /// domain make_evaluation_domain(M + n + 1) returns.  A call site then changes by the class name alone; callers that manage
    /// contexts / domains / shards themselves keep using the device-key overloads above.
    /// With a default device group (set_default_group, or ZKHIP_DEVICES naming several GPUs: multiexp.hpp) the SAME call proves over all of
    /// its devices -- slices of the key on every member, the exchange inside the library: a reference caller reaches BASELINE cfg 4's
    /// 8-GPU configuration by setting ZKHIP_DEVICES=0,1,2,3,4,5,6,7 and nothing else.
    template <typename K = KeyType, typename std::enable_if<std::is_same<K, KeyType>::value && curve_adapter<CurveType>::has_field_constants, bool>::type = true>
    static proof_type process(const K &proving_key, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input) {
        return process(proving_key, primary_input, auxiliary_input, random_scalar(), random_scalar());
    }
    template <typename K = KeyType, typename std::enable_if<std::is_same<K, KeyType>::value && curve_adapter<CurveType>::has_field_constants, bool>::type = true>
    static proof_type process(const K &proving_key, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input,
                              const scalar_value_type &r, const scalar_value_type &s) {
        if (const device_group *g = default_group()) return process(cached_group_key(*g, proving_key), primary_input, auxiliary_input, r, s);
        return process(cached_device_key(proving_key), primary_input, auxiliary_input, r, s);
    }
    /// drop the cached device key(s) of `proving_key` (and with them the resident queries); true if there was one
    static bool forget(const KeyType &proving_key) {
        const bool one = device_key_cache().erase(&proving_key) != 0, many = group_key_cache().erase(&proving_key) != 0;
        return one || many;
    }

private:
    /// What identifies a key's CONTENT cheaply (the address and the query sizes alone also match a second key of the same
    /// circuit that reuses the slot -- a fresh trapdoor in a loop --, which would then be proved with the first key's resident
    /// queries): alpha_g1, delta_g1 and the first and last entry of the A, H and L queries, as affine limbs.
    static std::vector<std::uint64_t> key_fingerprint(const KeyType &pk) {
        std::vector<std::uint64_t> fp = {(std::uint64_t)pk.A_query.size(), (std::uint64_t)pk.H_query.size(), (std::uint64_t)pk.L_query.size(),
                                         (std::uint64_t)pk.B_query.values.size()};
        auto add = [&fp](const auto &point) {
            std::uint64_t limbs[2 * adapter::g1_coord_limbs];
            const bool finite = adapter::point_to_affine_limbs(point, limbs);
            fp.push_back(finite ? 1 : 0);
            if (finite) fp.insert(fp.end(), limbs, limbs + 2 * adapter::g1_coord_limbs);
        };
        add(pk.alpha_g1);
        add(pk.delta_g1);
        for (const auto *q : {&pk.A_query, &pk.H_query, &pk.L_query})
            if (!q->empty()) {
                add(*q->begin());
                add(*(q->begin() + (q->size() - 1)));
            }
        return fp;
    }
    struct cached_key {
        std::vector<std::uint64_t> fingerprint;    // of the host key the device key was built from: another key in the same object is rebuilt
        std::unique_ptr<proving_key_type> device;
    };
    static std::map<const KeyType *, cached_key> &device_key_cache() {
        /* the context first: thread_local objects die in reverse order of construction, and the cached keys release their device memory
           THROUGH the default context (with the cache constructed first, its keys outlived the context they free into) */
        (void)default_context();
        thread_local std::map<const KeyType *, cached_key> cache;    // per thread, like default_context(): a context is not thread-safe
        return cache;
    }
    static const proving_key_type &cached_device_key(const KeyType &pk) {
        cached_key &e = device_key_cache()[&pk];
        std::vector<std::uint64_t> fp = key_fingerprint(pk);
        if (!e.device || e.fingerprint != fp) {
            e.device.reset();    // the old key's queries go first: two resident keys of 2^20 constraints are 1.5 GB
            e.device.reset(new proving_key_type(pk));
            e.fingerprint = std::move(fp);
        }
        return *e.device;
    }

    struct cached_group_key_entry {
        std::vector<std::uint64_t> fingerprint;
        const device_group *group = nullptr;    // the group the member keys live on: another default group rebuilds them
        std::unique_ptr<r1cs_gg_ppzksnark_proving_key_group_hip<CurveType, KeyType>> device;
    };
    static std::map<const KeyType *, cached_group_key_entry> &group_key_cache() {
        (void)default_group();    // constructed first, destroyed last: the cached keys release their device memory through its contexts
        thread_local std::map<const KeyType *, cached_group_key_entry> cache;
        return cache;
    }
    static const r1cs_gg_ppzksnark_proving_key_group_hip<CurveType, KeyType> &cached_group_key(const device_group &g, const KeyType &pk) {
        cached_group_key_entry &e = group_key_cache()[&pk];
        std::vector<std::uint64_t> fp = key_fingerprint(pk);
        if (!e.device || e.group != &g || e.fingerprint != fp) {
            e.device.reset();
            e.device.reset(new r1cs_gg_ppzksnark_proving_key_group_hip<CurveType, KeyType>(g, pk));
            e.group = &g;
            e.fingerprint = std::move(fp);
        }
        return *e.device;
    }

public:

    static proof_type process(const proving_key_type &pk, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input,
                              const scalar_value_type &r, const scalar_value_type &s) {
        if (pk.shard.world != 1) throw std::runtime_error("process: sharded key -- use process_partial / finish (or the all-gather overload)");
        typedef std::chrono::steady_clock clock;
        const auto t0 = clock::now();
        drain_on_unwind guard {pk};
        enqueue(pk, primary_input, auxiliary_input);
        const auto t1 = clock::now();
        /* host products that do not depend on the MSM results, computed while the GPU works (prover.hpp:142-155) */
        const host_terms t = host_products(pk, r, s);
        const auto t2 = clock::now();
        const std::vector<std::uint64_t> sums = collect(pk);    // the one synchronisation of the proof
        guard.armed = false;
        const auto t3 = clock::now();
        proof_type proof = assemble(pk, sums, 1, r, s, t);
        const clock::time_point marks[5] = {t0, t1, t2, t3, clock::now()};
        for (int i = 0; i < 4; ++i) pk.last_phase_ms[i] = std::chrono::duration<double, std::milli>(marks[i + 1] - marks[i]).count();
        return proof;
    }

    // ---- one proof sharded over several GPUs (one process per GPU) ---------------------------------------
    // Every rank holds a slice of each query (proving_key_type(ctx, pk, dom, rank, world)), runs the witness map in
    // full (replicated: 7 NTTs, no exchange) and the five MSMs over its slices; the only exchange is one all-gather
    // of partial_limbs() u64 words per rank (4 G1 + 1 G2 Jacobian points, 864 bytes for BLS12-381), after which
    // every rank can assemble the proof.

    /// u64 words of one rank's partial sums: A, B.h, H, L (G1) then B.g (G2), Jacobian
    static constexpr std::size_t partial_limbs() { return 4 * 3 * adapter::g1_coord_limbs + 3 * adapter::g2_coord_limbs; }

    /// this rank's partial sums (synchronises)
    static std::vector<std::uint64_t> process_partial(const proving_key_type &pk, const primary_input_type &primary_input,
                                                      const auxiliary_input_type &auxiliary_input) {
        drain_on_unwind guard {pk};
        enqueue(pk, primary_input, auxiliary_input);
        std::vector<std::uint64_t> sums = collect(pk);
        guard.armed = false;
        return sums;
    }
    /// `gathered`: world x partial_limbs() words, rank-major (what an all-gather of process_partial's result yields)
    static proof_type finish(const proving_key_type &pk, const std::vector<std::uint64_t> &gathered, const scalar_value_type &r,
                             const scalar_value_type &s) {
        if (gathered.size() != pk.shard.world * partial_limbs()) throw std::runtime_error("finish: gathered partial sums have the wrong size");
        return assemble(pk, gathered, pk.shard.world, r, s, host_products(pk, r, s));
    }
    /// process with the exchange supplied by the caller: all_gather(mine, words, all) fills `all` with world x words
    /// (RCCL through torch.distributed, MPI, ...)
    template <typename AllGather>
    static proof_type process(const proving_key_type &pk, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input,
                              const scalar_value_type &r, const scalar_value_type &s, AllGather all_gather) {
        drain_on_unwind guard {pk};
        enqueue(pk, primary_input, auxiliary_input);
        const host_terms t = host_products(pk, r, s);
        const std::vector<std::uint64_t> mine = collect(pk);
        guard.armed = false;
        std::vector<std::uint64_t> all(pk.shard.world * partial_limbs());
        all_gather(mine.data(), mine.size(), all.data());
        return assemble(pk, all, pk.shard.world, r, s, t);
    }

    /// The same with the partial sums STAYING ON THE DEVICE for the exchange: `d_mine` (partial_limbs() u64) and `d_all`
    /// (world x partial_limbs() u64) are device buffers of the caller's collective library (RCCL through torch.distributed: tensors);
    /// this rank's sums are copied into d_mine on the device, `all_gather()` -- no arguments: the caller knows its buffers -- must
    /// return when d_all is complete, and ONE download of d_all follows.  No host round trip before the collective.
    template <typename AllGather>
    static proof_type process_device_gather(const proving_key_type &pk, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input,
                                            const scalar_value_type &r, const scalar_value_type &s, void *d_mine, void *d_all, AllGather all_gather) {
        drain_on_unwind guard {pk};
        enqueue(pk, primary_input, auxiliary_input);
        const host_terms t = host_products(pk, r, s);
        if (pk.side) {
            pk.ctx.wait_for(*pk.side);
            pk.ctx.set_option("msm_sort_tile_log", pk.saved_sort_tile_log);
        }
        check(zkhip_memcpy_d2d_async(pk.ctx.get(), d_mine, pk.d_results.get(), partial_limbs() * 8), "zkhip_memcpy_d2d_async", pk.ctx.get());
        /* kernels flag what they cannot signal otherwise; zkhip_device_status synchronises the stream: d_mine is complete after it */
        check(zkhip_device_status(pk.ctx.get(), nullptr), "zkhip_device_status", pk.ctx.get());
        if (pk.side) check(zkhip_device_status(pk.side->get(), nullptr), "zkhip_device_status", pk.side->get());
        guard.armed = false;    // both streams have drained
        all_gather();
        std::vector<std::uint64_t> all(pk.shard.world * partial_limbs());
        pk.ctx.d2h(all.data(), d_all, all.size() * 8);
        return assemble(pk, all, pk.shard.world, r, s, t);
    }

    // ---- one proof over a device group: N GPUs, ONE caller, the exchange inside the library ---------------------------------------
    typedef r1cs_gg_ppzksnark_proving_key_group_hip<CurveType, KeyType> group_key_type;

    static proof_type process(const group_key_type &gk, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input) {
        return process(gk, primary_input, auxiliary_input, random_scalar(), random_scalar());
    }
    /// Every member runs the witness map and its five partial multiexps (all enqueued from this thread, nothing waits in between); the
    /// 4 G1 + 1 G2 Jacobian partial sums of all members meet on member 0 through device_group::all_gather -- RCCL over xGMI, peer copies
    /// or a staged copy, in stream order -- and ONE download + the host's assembly follow.
    static proof_type process(const group_key_type &gk, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input,
                              const scalar_value_type &r, const scalar_value_type &s) {
        typedef std::chrono::steady_clock clock;
        const std::size_t world = gk.members.size();
        if (world == 0) throw std::runtime_error("process: empty device group key");
        const auto t0 = clock::now();
        std::vector<std::unique_ptr<drain_on_unwind>> guards;
        for (const auto &m : gk.members) guards.emplace_back(new drain_on_unwind {*m});
        /* The assignment crosses PCIe ONCE, to member 0; the other members pull it from there (peer copies over xGMI, each on its own
           stream behind an event that marks the end of member 0's upload -- NOT of its proof): a copy out of the caller's pageable vector
           blocks this thread for its duration, and eight of them in a row would cost more than the sharded proof itself. */
        const std::size_t num_inputs = primary_input.size(), num_variables = primary_input.size() + auxiliary_input.size();
        upload_assignment(*gk.members[0], primary_input, auxiliary_input);
        for (std::size_t k = 1; k < world; ++k) {
            const proving_key_type &pk = *gk.members[k];
            pk.reserve_work(num_variables + 1, pk.constraint_system.domain_size(), partial_limbs() * 8);
            gk.group.copy(k, pk.d_cpa.get(), 0, gk.members[0]->d_cpa.get(), 32 * (num_variables + 1));
        }
        /* ~60 launches per member: from three members on, every member but the first is enqueued by a host thread of its own (a context
           is used by one thread at a time: distinct contexts are independent), so that the last member does not start N x 0.3 ms late */
        if (world <= 2) {
            for (std::size_t k = 0; k < world; ++k) enqueue_compute(*gk.members[k], num_inputs, num_variables);
        } else {
            std::vector<std::future<void>> others;
            for (std::size_t k = 1; k < world; ++k)
                others.push_back(std::async(std::launch::async, [&gk, k, num_inputs, num_variables]() { enqueue_compute(*gk.members[k], num_inputs, num_variables); }));
            std::exception_ptr failed;
            try {
                enqueue_compute(*gk.members[0], num_inputs, num_variables);
            } catch (...) {
                failed = std::current_exception();
            }
            for (auto &f : others) {
                try {
                    f.get();
                } catch (...) {
                    if (!failed) failed = std::current_exception();
                }
            }
            if (failed) std::rethrow_exception(failed);
        }
        const auto t1 = clock::now();
        const host_terms t = host_products(*gk.members[0], r, s);
        const auto t2 = clock::now();
        std::vector<const void *> send(world);
        std::vector<void *> recv(world, nullptr);
        for (std::size_t k = 0; k < world; ++k) {
            const proving_key_type &pk = *gk.members[k];
            if (pk.side) {
                pk.ctx.wait_for(*pk.side);
                pk.ctx.set_option("msm_sort_tile_log", pk.saved_sort_tile_log);
            }
            send[k] = pk.d_results.get();
        }
        const context &root = gk.members[0]->ctx;
        if (!gk.d_all) gk.d_all = root.alloc(world * partial_limbs() * 8);
        recv[0] = gk.d_all.get();
        gk.group.all_gather(send, recv, partial_limbs() * 8);
        std::vector<std::uint64_t> all(world * partial_limbs());
        root.d2h(all.data(), gk.d_all.get(), all.size() * 8);    // the root's stream is behind the exchange
        /* kernels flag what they cannot signal otherwise; zkhip_device_status drains every member's streams */
        for (std::size_t k = 0; k < world; ++k) {
            const proving_key_type &pk = *gk.members[k];
            check(zkhip_device_status(pk.ctx.get(), nullptr), "zkhip_device_status", pk.ctx.get());
            if (pk.side) check(zkhip_device_status(pk.side->get(), nullptr), "zkhip_device_status", pk.side->get());
            guards[k]->armed = false;
        }
        const auto t3 = clock::now();
        proof_type proof = assemble(*gk.members[0], all, world, r, s, t);
        const clock::time_point marks[5] = {t0, t1, t2, t3, clock::now()};
        for (int i = 0; i < 4; ++i) gk.last_phase_ms[i] = std::chrono::duration<double, std::milli>(marks[i + 1] - marks[i]).count();
        return proof;
    }

private:
    /* What the proof needs besides the five multiexps (prover.hpp:141-155), regrouped so that everything that does not depend on
       a device result is computed while the device works:
           A = alpha + At + r delta                           B = beta + Bt + s delta     (G2; the G1 copy only feeds C)
           C = Ht + Lt + s A + r B_g1 - r s delta
             = Ht + Lt + s At + r Bt_h + [s alpha + r beta_g1 + r s delta]
       -- the bracket and r delta, s delta_g2 are `host_terms`; s At and r Bt_h are the two products left for after the results
       arrive (`assemble` runs them on two threads). */
    /// enqueue() leaves asynchronous copies out of the CALLER's vectors (and kernels over the key's buffers) in flight; if anything
    /// throws before collect() has drained them -- a failed check, std::async out of threads -- the stack must not unwind under
    /// that DMA: the guard waits for both streams (and gives the caller's context its sort-tile setting back) on the way out.
    struct drain_on_unwind {
        const proving_key_type &pk;
        bool armed = true;
        ~drain_on_unwind() {
            if (!armed) return;
            try {
                pk.ctx.sync();
                if (pk.side) {
                    pk.side->sync();
                    pk.ctx.set_option("msm_sort_tile_log", pk.saved_sort_tile_log);
                }
            } catch (...) {
            }
        }
    };
    struct host_terms {
        typename adapter::g1_value_type r_delta, c_base;
        typename adapter::g2_value_type s_delta2;
    };
    static host_terms host_products(const proving_key_type &pk, const scalar_value_type &r, const scalar_value_type &s) {
        const auto &k = pk.host;
        auto g2 = std::async(std::launch::async, [&]() { return s * k.delta_g2; });
        auto r_delta = r * k.delta_g1;
        auto c_base = s * k.alpha_g1 + r * k.beta_g1 + s * r_delta;
        return {r_delta, c_base, g2.get()};
    }

    /// enqueue the whole device side of a proof on the context's stream (no synchronisation)
    static void enqueue(const proving_key_type &pk, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input) {
        upload_assignment(pk, primary_input, auxiliary_input);
        enqueue_compute(pk, primary_input.size(), primary_input.size() + auxiliary_input.size());
    }
    /// the key's work buffers (first use) and const_padded_assignment = (1, x, w) on its way to d_cpa (asynchronous)
    static void upload_assignment(const proving_key_type &pk, const primary_input_type &primary_input, const auxiliary_input_type &auxiliary_input) {
        const context &ctx = pk.ctx;
        const std::size_t num_inputs = primary_input.size();
        const std::size_t num_variables = primary_input.size() + auxiliary_input.size();
        /* Everything below is enqueued on the context's stream without intermediate synchronisation; the device
           buffers live in the key object (allocated on first use) so a proof costs no hipMalloc. */
        pk.reserve_work(num_variables + 1, pk.constraint_system.domain_size(), partial_limbs() * 8);
        char *cpa = static_cast<char *>(pk.d_cpa.get());

        /* const_padded_assignment = (1, x, w) (prover.hpp:102-106) */
        std::uint64_t *z = static_cast<std::uint64_t *>(pk.h_cpa.get());
        z[0] = 1;
        z[1] = z[2] = z[3] = 0;
        for (std::size_t i = 0; i < num_inputs; ++i) adapter::scalar_to_limbs(primary_input[i], &z[4 * (1 + i)]);
        /* the auxiliary input (almost all of the assignment) is converted into the page-locked staging buffer in
           slices by a few host threads, and every slice is sent as soon as it is ready: the conversion of slice k + 1
           overlaps the PCIe copy of slice k */
        if (detail::canonical_scalars<adapter>::value && pk.direct_assignment_upload) {
            /* scalar values that ARE canonical limbs in memory: the auxiliary input goes out as it lies (0.3 ms per 2^20-constraint
               proof less than through the staging buffer) */
            static_assert(!detail::canonical_scalars<adapter>::value || sizeof(scalar_value_type) == 32, "canonical-limb scalars are 4 x u64");
            check(zkhip_memcpy_h2d_async(ctx.get(), cpa, z, 32 * (1 + num_inputs)), "zkhip_memcpy_h2d_async", ctx.get());
            if (!auxiliary_input.empty())
                check(zkhip_memcpy_h2d_async(ctx.get(), cpa + 32 * (1 + num_inputs), auxiliary_input.data(), 32 * auxiliary_input.size()),
                      "zkhip_memcpy_h2d_async", ctx.get());
        } else {
            const std::size_t aux = auxiliary_input.size(), slices = aux >= (std::size_t)1 << 16 ? 8 : 1, per = (aux + slices - 1) / slices;
            std::uint64_t *za = z + 4 * (1 + num_inputs);
            std::vector<std::future<void>> ready;
            for (std::size_t k = 1; k < slices; ++k)
                ready.push_back(std::async(std::launch::async, [&, k]() {
                    for (std::size_t i = k * per; i < std::min(aux, (k + 1) * per); ++i) adapter::scalar_to_limbs(auxiliary_input[i], &za[4 * i]);
                }));
            for (std::size_t i = 0; i < std::min(aux, per); ++i) adapter::scalar_to_limbs(auxiliary_input[i], &za[4 * i]);
            check(zkhip_memcpy_h2d_async(ctx.get(), cpa, z, 32 * (1 + num_inputs + std::min(aux, per))), "zkhip_memcpy_h2d_async", ctx.get());
            for (std::size_t k = 1; k < slices; ++k) {
                ready[k - 1].get();
                const std::size_t lo = k * per, hi = std::min(aux, (k + 1) * per);
                if (hi > lo)
                    check(zkhip_memcpy_h2d_async(ctx.get(), cpa + 32 * (1 + num_inputs + lo), za + 4 * lo, 32 * (hi - lo)), "zkhip_memcpy_h2d_async",
                          ctx.get());
            }
        }
    }
    /// the witness map and the five multiexps over the assignment at d_cpa (enqueued only)
    static void enqueue_compute(const proving_key_type &pk, std::size_t num_inputs, std::size_t num_variables) {
        const context &ctx = pk.ctx;
        const query_shard &sh = pk.shard;
        const std::size_t degree = pk.constraint_system.domain_size();
        const std::size_t jl1 = 3 * adapter::g1_coord_limbs;
        pk.reserve_work(num_variables + 1, degree, partial_limbs() * 8);
        if (pk.side) {    // for this proof only: collect() gives the caller's context its own setting back
            pk.saved_sort_tile_log = ctx.get_option("msm_sort_tile_log");
            ctx.set_option("msm_sort_tile_log", 12);
        }
        char *cpa = static_cast<char *>(pk.d_cpa.get());
        std::uint64_t *d_res = static_cast<std::uint64_t *>(pk.d_results.get());
        /* qap_wit.coefficients_for_H, resident (prover.hpp:79-83) */
        std::uint64_t g[4];
        const zkhip_domain dd = pk.evaluation_domain.c_desc();
        adapter::scalar_to_limbs(pk.domain.coset_generator, g);
        check(zkhip_groth16_witness_h_domain_dev(ctx.get(), pk.constraint_system.get(), cpa, &dd, g, pk.d_h.get(), pk.d_scratch.get()),
              "zkhip_groth16_witness_h_domain_dev", ctx.get());
        /* evaluation_Bt: kc_multiexp_with_mixed_addition over the sparse (G2, G1) query (prover.hpp:116-123); a sharded key
           holds a slice of the index list */
        check(zkhip_fr_gather_dev(ctx.get(), cpa, num_variables + 1, pk.d_B_indices.get(), pk.B_count, pk.d_bs.get()), "zkhip_fr_gather_dev", ctx.get());
        if (pk.experiment_skip_g2) {
            /* EXPERIMENT ONLY (wrong proof): what a proof would cost if the G2 multiexp were free -- the ceiling of anything that could be
               gained on the second stream */
        } else if (pk.side) {
            /* the G2 part on the second stream, after the gather; the four G1 multiexps below do not depend on it */
            pk.side->wait_for(ctx);
            check(zkhip_msm_dev(pk.side->get(), pk.B_query_g.get(), 0, pk.B_count, pk.d_bs.get(), d_res + 4 * jl1), "zkhip_msm_dev(B.g)", pk.side->get());
        } else {
            check(zkhip_msm_dev(ctx.get(), pk.B_query_g.get(), 0, pk.B_count, pk.d_bs.get(), d_res + 4 * jl1), "zkhip_msm_dev(B.g)", ctx.get());
        }
        /* the four G1 multiexps as one batch (their bucket reductions share one launch):
           evaluation_At (prover.hpp:108-114), evaluation_Bt.h (:116-123), evaluation_Ht over H_query[0 .. degree - 1)
           (:125-131), evaluation_Lt over the auxiliary part of the assignment (:133-139) -- each over this key's slice */
        if (sh.A_n > num_variables + 1 - sh.A_lo || sh.H_n > degree - 1 - sh.H_lo || sh.L_n > num_variables - num_inputs - sh.L_lo)
            throw std::runtime_error("prover: the key's query slices do not fit this assignment");
        /* members over the same scalars and rows sit next to each other: the batch sorts the assignment's digits once for them */
        const zkhip_bases *qb[4] = {pk.A_query.get(), pk.L_query.get(), pk.B_query_h.get(), pk.H_query.get()};
        const std::size_t qo[4] = {0, 0, 0, 0};
        const std::size_t qn[4] = {sh.A_n, pk.L_rows_aligned ? num_variables + 1 : sh.L_n, pk.B_rows_aligned ? num_variables + 1 : pk.B_count, sh.H_n};
        const void *qs[4] = {cpa + 32 * sh.A_lo, pk.L_rows_aligned ? cpa : cpa + 32 * (num_inputs + 1 + sh.L_lo),
                             pk.B_rows_aligned ? static_cast<const void *>(cpa) : pk.d_bs.get(),
                             static_cast<const char *>(pk.d_h.get()) + 32 * sh.H_lo};
        void *qr[4] = {d_res, d_res + 3 * jl1, d_res + jl1, d_res + 2 * jl1};
        check(zkhip_msm_batch_dev(ctx.get(), 4, qb, qo, qn, qs, qr), "zkhip_msm_batch_dev", ctx.get());
    }
    /// the five partial sums of this rank, after the stream has drained
    static std::vector<std::uint64_t> collect(const proving_key_type &pk) {
        std::vector<std::uint64_t> res(partial_limbs());
        if (pk.side) {
            pk.ctx.wait_for(*pk.side);
            pk.ctx.set_option("msm_sort_tile_log", pk.saved_sort_tile_log);    // the caller's context gets its own sort tiles back
        }
        pk.ctx.d2h(res.data(), pk.d_results.get(), res.size() * 8);
        /* kernels flag what they cannot signal otherwise (a B-query index beyond the assignment, a plan overflow) */
        check(zkhip_device_status(pk.ctx.get(), nullptr), "zkhip_device_status", pk.ctx.get());
        if (pk.side) check(zkhip_device_status(pk.side->get(), nullptr), "zkhip_device_status", pk.side->get());
        return res;
    }
    /// fold the ranks' partial sums and build the proof (prover.hpp:141-157)
    static proof_type assemble(const proving_key_type &pk, const std::vector<std::uint64_t> &gathered, std::size_t world, const scalar_value_type &r,
                               const scalar_value_type &s, const host_terms &t) {
        const std::size_t jl1 = 3 * adapter::g1_coord_limbs, pl = partial_limbs();
        auto evaluation_At = adapter::g1_value_type::zero(), evaluation_Bt_h = evaluation_At, evaluation_Ht = evaluation_At, evaluation_Lt = evaluation_At;
        auto evaluation_Bt_g = adapter::g2_value_type::zero();
        for (std::size_t k = 0; k < world; ++k) {
            const std::uint64_t *res = gathered.data() + k * pl;
            evaluation_At = evaluation_At + adapter::g1_from_jacobian(&res[0]);
            evaluation_Bt_h = evaluation_Bt_h + adapter::g1_from_jacobian(&res[jl1]);
            evaluation_Ht = evaluation_Ht + adapter::g1_from_jacobian(&res[2 * jl1]);
            evaluation_Lt = evaluation_Lt + adapter::g1_from_jacobian(&res[3 * jl1]);
            evaluation_Bt_g = evaluation_Bt_g + adapter::g2_from_jacobian(&res[4 * jl1]);
        }
        const auto &k = pk.host;
        /* the two products that need a device result, side by side */
        auto s_At = std::async(std::launch::async, [&]() { return s * evaluation_At; });
        const auto r_Bt_h = r * evaluation_Bt_h;
        /* A = alpha + sum_i(a_i*A_i(t)) + r*delta */
        auto g1_A = k.alpha_g1 + evaluation_At + t.r_delta;
        /* B = beta + sum_i(a_i*B_i(t)) + s*delta */
        auto g2_B = k.beta_g2 + evaluation_Bt_g + t.s_delta2;
        /* C = sum_i(a_i*((beta*A_i(t) + alpha*B_i(t) + C_i(t)) + H(t)*Z(t))/delta) + A*s + r*b - r*s*delta  (regrouped: host_terms) */
        auto g1_C = evaluation_Ht + evaluation_Lt + s_At.get() + r_Bt_h + t.c_base;
        return proof_type {g1_A, g2_B, g1_C};
    }

public:
    /// Uniform element of Fr from the operating system's CSPRNG (getrandom(2)): bitlen(r) random bits, rejection
    /// sampled below r.  No generator state: nothing links the blinders of two proofs and concurrent provers do
    /// not share anything.  (The reference: algebra::random_element<scalar_field_type>(), prover.hpp:92-93.)
    static scalar_value_type random_scalar() {
        std::uint64_t mod[4], w[4];
        adapter::scalar_modulus(mod);
        int top = 63;
        while (top > 0 && !((mod[3] >> top) & 1)) --top;
        const std::uint64_t mask = top == 63 ? ~(std::uint64_t)0 : (((std::uint64_t)1 << (top + 1)) - 1);
        for (;;) {
            detail::os_random_bytes(w, sizeof(w));
            w[3] &= mask;
            bool lt = false;
            for (int i = 3; i >= 0; --i) {
                if (w[i] != mod[i]) {
                    lt = w[i] < mod[i];
                    break;
                }
            }
            if (lt) return adapter::scalar_from_limbs(w);
        }
    }
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_R1CS_GG_PPZKSNARK_HPP
