//---------------------------------------------------------------------------//
// zkhip shim: the LPC (list polynomial commitment, FRI-based) scheme placeholder plugs in as `commitment_scheme_type`,
// with every polynomial-sized step on the MI355X.
//
// Mirrors zk/commitments/polynomial/lpc.hpp (class lpc_commitment_scheme, :50-300) on top of
// zk/commitments/batched_commitment.hpp (polys_evaluator, :58-249) and the commit phase of
// algorithms::proof_eval<FRI> (zk/commitments/detail/polynomial/basic_fri.hpp:666-742) -- same member names, same call
// order: append_to_batch / commit(batch) / append_eval_point[s] / set_batch_size / mark_batch_as_fixed /
// preprocess / setup / proof_eval(transcript), public `_z`, `is_lpc()`.
//
// What runs where
//   device  commit: the batch's inverse NTTs (the coefficient forms stay resident: the reference re-derives them per use,
//           lpc.hpp:150,173), the extension of every polynomial to D[0] and the coset-ordered leaf layout
//           (basic_fri.hpp:433-496); proof_eval: eval_polys (block-Horner), the combined quotient
//           Q = sum_points (sum_k theta^k (g_k - z_k)) / (X - point)  (lpc.hpp:139-186; one pass over the resident
//           coefficients + one synthetic division per point), its extension, and the FRI commit phase
//           (fold_polynomial rounds + per-round leaf layouts, basic_fri.hpp:705-742).
//   caller  hashing: the TreeBuilder builds the Merkle tree of a precommitment and exposes .root() (containers::merkle_tree +
//           the scheme's hash, outside this tree's scope: SURVEY 2).  Three shapes, best first (see tree_builder_kind):
//             streaming  b.begin(total_elements, elements_per_leaf); b.absorb(ptr, first_element, count) ...; b.finish()
//                        -- the leaves arrive in slices of whole leaves through two page-locked buffers: slice k + 1 crosses
//                        PCIe while the caller hashes slice k, nothing is materialised;
//             span       b(ptr, total_elements, elements_per_leaf) over a page-locked buffer the scheme keeps (one copy at link speed);
//             vector     b(const std::vector<value_type> &leaves, elements_per_leaf), as round 2 (a vector the scheme keeps).
//           transcript:
//           duck-typed on VALUES like kzg_v2.hpp (transcript(root), transcript.challenge()); the QUERY phase of
//           proof_eval<FRI> (basic_fri.hpp:747-930: lambda Merkle openings at transcript-derived indices) reads only
//           trees and a few evaluations and is left to the caller, who finds everything it needs through
//           trees() / fri_trees() / fri_round_polynomial(i) / fri_alphas().
//
// Over a device group (constructor taking a device_group): commit(batch) deals the batch's polynomials over the members -- uploads over
// their own PCIe links, extensions on their own GPUs --, then the path's one exchange: the leaf range of every LEAF OWNER (the first 2^k
// members) is made of 2^fri_step segments of every polynomial, which the members pack and push to the owners; each owner lays out ITS
// leaves and sends them to the host over ITS link while the caller hashes.  Same leaves, same order, same roots as on one device; the
// coefficient forms are gathered on member 0, where proof_eval runs as before.  See commit_group.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_LPC_HPP
#define ZKHIP_SHIM_LPC_HPP

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <deque>
#include <functional>
#include <future>
#include <iterator>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "fri.hpp"
#include "kzg_v2.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// What the scheme reads of basic_fri::params_type (basic_fri.hpp:86-140): the domains D[t] (size 2^(log_domain - t),
/// generator root_of_unity(log_domain - t)) and the step list; r = sum(step_list) folding rounds.
template <typename CurveType>
struct fri_params_hip {
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    std::size_t log_domain = 0;
    std::vector<std::size_t> step_list;
    std::function<value_type(std::size_t log_n)> root_of_unity;
    /// the roots the reference's domains D[i] = make_evaluation_domain(2^(log_domain - i)) carry (fri params, basic_fri.hpp:84-118),
    /// from the curve adapter's field constants
    static fri_params_hip standard(std::size_t log_domain, std::vector<std::size_t> step_list) {
        return {log_domain, std::move(step_list), [](std::size_t log_n) { return curve_adapter<CurveType>::root_of_unity(log_n); }};
    }
};

namespace detail {
    /// which of the three tree-builder shapes a type offers (see the header comment)
    enum class tree_builder_kind { streaming, span, vector };
    template <typename B, typename V, typename = void>
    struct is_streaming_builder : std::false_type { };
    template <typename B, typename V>
    struct is_streaming_builder<B, V,
                                std::void_t<decltype(std::declval<B &>().begin(std::size_t(), std::size_t())),
                                            decltype(std::declval<B &>().absorb(static_cast<const V *>(nullptr), std::size_t(), std::size_t())),
                                            decltype(std::declval<B &>().finish())>> : std::true_type { };
    template <typename B, typename V, typename = void>
    struct is_span_builder : std::false_type { };
    template <typename B, typename V>
    struct is_span_builder<B, V, std::void_t<decltype(std::declval<B &>()(static_cast<const V *>(nullptr), std::size_t(), std::size_t()))>> : std::true_type { };
    template <typename B, typename V, tree_builder_kind K>
    struct tree_builder_result;
    template <typename B, typename V>
    struct tree_builder_result<B, V, tree_builder_kind::streaming> {
        typedef typename std::decay<decltype(std::declval<B &>().finish())>::type type;
    };
    template <typename B, typename V>
    struct tree_builder_result<B, V, tree_builder_kind::span> {
        typedef typename std::decay<decltype(std::declval<B &>()(static_cast<const V *>(nullptr), std::size_t(), std::size_t()))>::type type;
    };
    template <typename B, typename V>
    struct tree_builder_result<B, V, tree_builder_kind::vector> {
        typedef typename std::decay<decltype(std::declval<B &>()(std::declval<const std::vector<V> &>(), std::size_t()))>::type type;
    };
}    // namespace detail

/// `PolynomialType`: as for the KZG schemes (batched_commitment.hpp:56-64) -- size() and operator[] over contiguous scalars.
template <typename CurveType, typename TranscriptType, typename TreeBuilder, typename PolynomialType = polynomial_dfs<CurveType>>
class lpc_commitment_scheme_hip {
public:
    static constexpr bool is_lpc() { return true; }

    typedef curve_adapter<CurveType> adapter;
    typedef CurveType curve_type;
    typedef typename adapter::scalar_value_type value_type;
    typedef fri_params_hip<CurveType> params_type;
    typedef TranscriptType transcript_type;
    typedef PolynomialType poly_type;
    typedef eval_storage_hip<CurveType> eval_storage_type;
    static constexpr detail::tree_builder_kind builder_kind =
        detail::is_streaming_builder<TreeBuilder, value_type>::value
            ? detail::tree_builder_kind::streaming
            : (detail::is_span_builder<TreeBuilder, value_type>::value ? detail::tree_builder_kind::span : detail::tree_builder_kind::vector);
    typedef typename detail::tree_builder_result<TreeBuilder, value_type, builder_kind>::type precommitment_type;
    typedef typename std::decay<decltype(std::declval<const precommitment_type &>().root())>::type commitment_type;
    typedef std::map<std::size_t, std::vector<value_type>> preprocessed_data_type;
    struct fri_proof_type {
        std::vector<commitment_type> fri_roots;
        std::vector<value_type> final_polynomial;    // coefficients (math::polynomial(f.coefficients()), basic_fri.hpp:736-742)
        /* query_proofs: the caller's (see the header) */
    };
    struct proof_type {
        eval_storage_type z;
        fri_proof_type fri_proof;
    };

    eval_storage_type _z;

    lpc_commitment_scheme_hip(const context &ctx, const params_type &fri_params, TreeBuilder builder) :
        _ctx(ctx), _fri_params(fri_params), _builder(std::move(builder)), _etha(value_type::zero()) {
        std::size_t r = 0;
        for (std::size_t s : fri_params.step_list) r += s;
        if (fri_params.step_list.empty() || r > fri_params.log_domain) throw std::invalid_argument("lpc: step_list does not fit the domain");
    }

    /// the scheme over a device group: commit(batch) spreads over the members (commit_group), everything else runs on member 0
    lpc_commitment_scheme_hip(const device_group &group, const params_type &fri_params, TreeBuilder builder) :
        lpc_commitment_scheme_hip(group.root(), fri_params, std::move(builder)) {
        _group = &group;
        _gs.reset(new group_scratch[group.size()]);
    }

    const params_type &get_commitment_params() const { return _fri_params; }

    /// lpc.hpp:82-96: the fixed batches' polynomials evaluated at etha
    preprocessed_data_type preprocess(transcript_type &transcript) const {
        const value_type etha = transcript.challenge();
        preprocessed_data_type result;
        for (const auto &it : _batch_fixed) {
            if (!it.second) continue;
            result[it.first] = evaluate_batch_at(it.first, etha);
        }
        return result;
    }
    void setup(transcript_type &transcript, const preprocessed_data_type &preprocessed_data) {
        _etha = transcript.challenge();
        _fixed_polys_values = preprocessed_data;
    }
    /// Should be done after commitment (lpc.hpp:109-111)
    void mark_batch_as_fixed(std::size_t index) { _batch_fixed[index] = true; }

    // ---- polys_evaluator (batched_commitment.hpp:197-247) ----
    /// as the reference: the scheme keeps a COPY of the polynomial (batched_commitment.hpp:197-206)
    void append_to_batch(std::size_t index, const poly_type &poly) { own(index, poly_type(poly)); }
    template <typename ContainerType>
    void append_to_batch(std::size_t index, const ContainerType &polys) {
        for (const auto &p : polys) append_one(index, p);
    }
    /// hand the polynomials over instead of copying them, or LEND them (std::cref: alive until commit(index) returns) -- see kzg_v2.hpp
    void append_to_batch(std::size_t index, poly_type &&poly) { own(index, std::move(poly)); }
    void append_to_batch(std::size_t index, std::vector<poly_type> &&polys) {
        for (auto &p : polys) own(index, std::move(p));
        polys.clear();
    }
    void append_to_batch(std::size_t index, std::reference_wrapper<const poly_type> poly) {
        if (_locked[index]) throw std::runtime_error("append_to_batch: batch already committed");
        _polys[index].push_back(&poly.get());
    }
    void append_eval_point(std::size_t batch_id, const value_type &point) {
        for (auto &pts : _points.at(batch_id)) pts.push_back(point);
    }
    void append_eval_point(std::size_t batch_id, std::size_t poly_id, const value_type &point) { _points.at(batch_id).at(poly_id).push_back(point); }
    void append_eval_points(std::size_t batch_id, const std::vector<value_type> &points) {
        for (auto &pts : _points.at(batch_id)) pts.insert(pts.end(), points.begin(), points.end());
    }
    void append_eval_points(std::size_t batch_id, std::size_t poly_id, const std::vector<value_type> &points) {
        auto &pts = _points.at(batch_id).at(poly_id);
        pts.insert(pts.end(), points.begin(), points.end());
    }
    void set_batch_size(std::size_t batch_id, std::size_t batch_size) {
        _points[batch_id].resize(batch_size);
        _locked[batch_id] = true;
    }

    /// commit(index) (lpc.hpp:101-106): precommit<FRI>(polys, D[0], step_list.front()) -> the tree's root
    commitment_type commit(std::size_t index) {
        if (_group && _group->size() > 1 && !_polys[index].empty()) return commit_group(index);
        ZKHIP_PROFILE_SCOPE("Basic FRI Precommit time");    // commit = precommit<FRI> + root (lpc.hpp:101-106; the scope of basic_fri.hpp:449)
        const std::vector<const poly_type *> &polys = _polys[index];
        _locked[index] = true;    // state_commited (batched_commitment.hpp:163-166)
        _points[index].resize(polys.size());
        device_batch db;
        std::size_t total = 0;
        for (const poly_type *p : polys) {
            if (p->size() == 0 || (p->size() & (p->size() - 1)) || p->size() > domain_size(0)) throw std::runtime_error("lpc commit: bad polynomial size");
            db.offset.push_back(total);
            db.len.push_back(p->size());
            total += p->size();
        }
        db.data = _ctx.alloc(std::max<std::size_t>(1, total) * 32);
        /* poly.resize(D[0]->size()) for every polynomial (basic_fri.hpp:452-455): one call per chunk of equally sized polynomials; it
           leaves the COEFFICIENTS in the source buffer, which is what proof_eval reads later.  The chunks go up on a second in-order
           stream: chunk c + 1 crosses PCIe while chunk c is transformed. */
        const std::size_t D = domain_size(0), count = polys.size();
        void *d_ext = scratch(_scratch_ext, _scratch_ext_cap, std::max<std::size_t>(1, count) * D * 32);    // kept across commits: no GB-sized hipMalloc per batch
        std::uint64_t wd[4];
        adapter::scalar_to_limbs(_fri_params.root_of_unity(_fri_params.log_domain), wd);
        const bool pipelined = upload_chunk != 0 && count > upload_chunk;
        const context &up = pipelined ? upload_context() : _ctx;
        for (std::size_t i = 0; i < count;) {
            std::size_t j = i;
            while (j < count && db.len[j] == db.len[i] && (upload_chunk == 0 || j - i < upload_chunk)) ++j;
            for (std::size_t p = i; p < j; ++p) upload_scalars<adapter>(up, db.at(p), detail::poly_data<adapter>(*polys[p]), polys[p]->size());
            if (pipelined) _ctx.wait_for(up);
            const std::size_t log_n = log2_of(db.len[i]);
            std::uint64_t wn[4];
            adapter::scalar_to_limbs(_fri_params.root_of_unity(log_n), wn);
            check(zkhip_poly_resize_dev(_ctx.get(), adapter::id, db.at(i), log_n, j - i, wn, static_cast<char *>(d_ext) + 32 * i * D,
                                        _fri_params.log_domain, wd),
                  "zkhip_poly_resize_dev", _ctx.get());
            i = j;
        }
        _trees.erase(index);
        _trees.emplace(index, build_tree(d_ext, count, _fri_params.log_domain, _fri_params.step_list.front()));
        _dev[index] = std::move(db);
        /* the host polynomials are not read again: lent ones may go; copies and handed-over ones are released with the scheme (freeing
           gigabytes of host memory here would cost more than the commit's device work) */
        _polys[index].clear();    // no pointer to a lent polynomial outlives the call
        return _trees.at(index).root();
    }

    /// commits that ran over the device group, and over how many leaf owners the last one cut its leaves (tests and logs)
    std::size_t group_commits() const { return _group_commits; }
    std::size_t last_leaf_owners() const { return _last_owners; }

    /// polynomials per upload chunk of commit() (0: the whole batch in one transfer)
    std::size_t upload_chunk = 4;
    /// elements per slice handed to a streaming tree builder (rounded up to whole leaves)
    std::size_t leaf_slice_elements = (std::size_t)1 << 21;

    /// commit(index) over the device group.  Member k takes the k-th contiguous range of the batch's polynomials: its host thread uploads them
    /// over the member's own link (chunked, a second in-order stream, as on one device) and extends them to D[0] on the member's GPU.  The
    /// leaves are then cut by RANGE over the first 2^k members (the leaf owners): leaf x reads, of every polynomial, the positions
    /// x + j D / 2^step (fri_leaf_gather, poly.hip), so an owner's L = D / 2^step / owners consecutive leaves read 2^step segments of L
    /// consecutive evaluations per polynomial -- packed side by side they ARE the evaluations over a domain of D / owners points as far as the
    /// leaf layout is concerned (segment j of the small domain is segment j of the large one), and the owner runs the ordinary leaf kernel on
    /// them.  The exchange: every member packs, per owner, the segments of its polynomials (one strided copy) and pushes the block into the
    /// owner's buffer (zkhip_group_copy: ordered after the extension on the source's stream, on the owner's stream); owners x members
    /// copies of count / members x D / owners elements each.  Each owner then sends its leaves to the host over its own link while the
    /// caller hashes.  The coefficient forms the extension leaves behind are gathered on member 0 for proof_eval.
    commitment_type commit_group(std::size_t index) {
        ZKHIP_PROFILE_SCOPE("Basic FRI Precommit time");
        const device_group &group = *_group;
        const std::vector<const poly_type *> &polys = _polys[index];
        _locked[index] = true;
        _points[index].resize(polys.size());
        device_batch db;
        std::size_t total = 0;
        for (const poly_type *p : polys) {
            if (p->size() == 0 || (p->size() & (p->size() - 1)) || p->size() > domain_size(0)) throw std::runtime_error("lpc commit: bad polynomial size");
            db.offset.push_back(total);
            db.len.push_back(p->size());
            total += p->size();
        }
        db.data = _ctx.alloc(std::max<std::size_t>(1, total) * 32);
        const std::size_t D = domain_size(0), count = polys.size(), world = group.size(), step = _fri_params.step_list.front();
        std::size_t log_owners = 0;    // owners: a power of two (the compact domain is one), every owner with at least one leaf
        while (((std::size_t)2 << log_owners) <= world && _fri_params.log_domain >= step + log_owners + 1) ++log_owners;
        const std::size_t owners = (std::size_t)1 << log_owners, Ds = D >> log_owners, seg = Ds >> step, rows_per_poly = (std::size_t)1 << step;
        struct part {
            std::size_t lo = 0, hi = 0, elems = 0;
            std::shared_ptr<void> data;    // the member's own coefficient buffer (member 0 works in the batch buffer itself)
            char *base = nullptr;
        };
        std::vector<part> parts(world);
        for (std::size_t k = 0; k < world; ++k) {
            part &pt = parts[k];
            pt.lo = count / world * k + std::min(k, count % world);
            pt.hi = count / world * (k + 1) + std::min(k + 1, count % world);
            group_scratch &gs = _gs[k];
            if (k < owners) {
                grow(group[k], gs.cmp, gs.cmp_cap, count * Ds * 32);
                grow(group[k], gs.leaves, gs.leaves_cap, count * Ds * 32);
            }
            if (pt.hi == pt.lo) continue;
            pt.elems = (pt.hi < count ? db.offset[pt.hi] : total) - db.offset[pt.lo];
            if (k == 0) pt.base = static_cast<char *>(db.at(pt.lo));
            else {
                pt.data = group[k].alloc(pt.elems * 32);
                pt.base = static_cast<char *>(pt.data.get());
            }
            grow(group[k], gs.ext, gs.ext_cap, (pt.hi - pt.lo) * D * 32);
            grow(group[k], gs.send, gs.send_cap, (pt.hi - pt.lo) * D * 32);    // owners blocks of (hi - lo) * Ds
        }
        /* the caller's root-of-unity function is called from THIS thread only: one root per distinct size, up front */
        std::map<std::size_t, std::array<std::uint64_t, 4>> root_limbs;
        adapter::scalar_to_limbs(_fri_params.root_of_unity(_fri_params.log_domain), root_limbs[_fri_params.log_domain].data());
        for (std::size_t i = 0; i < count; ++i) {
            const std::size_t log_n = log2_of(db.len[i]);
            if (!root_limbs.count(log_n)) adapter::scalar_to_limbs(_fri_params.root_of_unity(log_n), root_limbs[log_n].data());
        }
        auto member_work = [&](std::size_t k) {
            part &pt = parts[k];
            if (pt.hi == pt.lo) return;
            const context &ctx = group[k];
            group_scratch &gs = _gs[k];
            const std::size_t mine = pt.hi - pt.lo;
            auto at = [&](std::size_t p) { return pt.base + 32 * (db.offset[p] - db.offset[pt.lo]); };
            const bool pipelined = upload_chunk != 0 && mine > upload_chunk;
            if (pipelined && !gs.up) gs.up.reset(new context(ctx.device()));
            const context &up = pipelined ? *gs.up : ctx;
            char *d_ext = static_cast<char *>(gs.ext.get());
            for (std::size_t i = pt.lo; i < pt.hi;) {
                std::size_t j = i;
                while (j < pt.hi && db.len[j] == db.len[i] && (upload_chunk == 0 || j - i < upload_chunk)) ++j;
                for (std::size_t p = i; p < j; ++p) upload_scalars<adapter>(up, at(p), detail::poly_data<adapter>(*polys[p]), polys[p]->size());
                if (pipelined) ctx.wait_for(up);
                const std::size_t log_n = log2_of(db.len[i]);
                check(zkhip_poly_resize_dev(ctx.get(), adapter::id, at(i), log_n, j - i, root_limbs.at(log_n).data(), d_ext + 32 * (i - pt.lo) * D,
                                            _fri_params.log_domain, root_limbs.at(_fri_params.log_domain).data()),
                      "zkhip_poly_resize_dev", ctx.get());
                i = j;
            }
            /* per owner: the 2^step segments [d seg + j D / 2^step, + seg) of each of my polynomials, side by side (rows of seg elements at a
               pitch of D / 2^step in, seg out) -- straight into my own buffer where I am the owner */
            for (std::size_t d = 0; d < owners; ++d) {
                char *dst = d == k ? static_cast<char *>(gs.cmp.get()) + 32 * pt.lo * Ds : static_cast<char *>(gs.send.get()) + 32 * d * mine * Ds;
                check(zkhip_memcpy_2d_d2d_async(ctx.get(), dst, seg * 32, d_ext + 32 * d * seg, (D >> step) * 32, seg * 32, mine * rows_per_poly),
                      "zkhip_memcpy_2d_d2d_async", ctx.get());
            }
        };
        {
            std::vector<std::future<void>> others;
            for (std::size_t k = 1; k < world; ++k)
                if (parts[k].hi != parts[k].lo) others.push_back(std::async(std::launch::async, member_work, k));
            std::exception_ptr failed;
            try {
                member_work(0);
            } catch (...) {
                failed = std::current_exception();
            }
            for (auto &o : others) {
                try {
                    o.get();
                } catch (...) {
                    if (!failed) failed = std::current_exception();
                }
            }
            if (failed) {
                try {
                    group.sync();    // nothing of this batch may still be running on buffers that are about to go
                } catch (...) {
                }
                std::rethrow_exception(failed);
            }
        }
        /* the exchange, from this thread: blocks to their owners, coefficient forms to member 0 */
        for (std::size_t k = 0; k < world; ++k) {
            const part &pt = parts[k];
            if (pt.hi == pt.lo) continue;
            const std::size_t mine = pt.hi - pt.lo;
            for (std::size_t d = 0; d < owners; ++d)
                if (d != k)
                    group.copy(d, static_cast<char *>(_gs[d].cmp.get()) + 32 * pt.lo * Ds, k, static_cast<const char *>(_gs[k].send.get()) + 32 * d * mine * Ds,
                               mine * Ds * 32);
            if (k != 0) group.copy(0, db.at(pt.lo), k, pt.base, pt.elems * 32);
        }
        std::vector<const void *> d_leaves(owners);
        for (std::size_t d = 0; d < owners; ++d) {
            check(zkhip_fri_leaves_dev(group[d].get(), _gs[d].cmp.get(), _fri_params.log_domain - log_owners, count, step, _gs[d].leaves.get()),
                  "zkhip_fri_leaves_dev", group[d].get());
            d_leaves[d] = _gs[d].leaves.get();
        }
        _trees.erase(index);
        _trees.emplace(index, build_tree_group(d_leaves, count * Ds, count * rows_per_poly));
        group.sync();    // the members' own coefficient buffers go with `parts`: every copy out of them has finished
        ++_group_commits;
        _last_owners = owners;
        _dev[index] = std::move(db);
        _polys[index].clear();    // no pointer to a lent polynomial outlives the call
        return _trees.at(index).root();
    }

    /// proof_eval (lpc.hpp:113-200) up to and including the FRI commit phase (basic_fri.hpp:705-742)
    proof_type proof_eval(transcript_type &transcript) {
        /* ZKHIP_LPC_PHASES=1: host wall time of the phases on stderr */
        const bool phases = std::getenv("ZKHIP_LPC_PHASES") != nullptr;
        auto tp = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!phases) return;
            _ctx.sync();
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "lpc proof_eval phase %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
            tp = now;
        };
        eval_polys();
        lap("eval_polys");
        for (const auto &it : _trees) transcript(it.second.root());

        /* Prepare z-s and combined_Q (lpc.hpp:124-186), coefficient form, resident */
        const value_type theta = transcript.challenge();
        value_type theta_acc = value_type::one();
        std::size_t max_len = 1;
        for (const auto &it : _dev)
            for (std::size_t l : it.second.len) max_len = std::max(max_len, l);
        auto d_combined = _ctx.alloc(max_len * 32), d_q = _ctx.alloc(max_len * 32);
        bool have_combined = false;
        auto add_quotient = [&](const std::vector<const void *> &ptrs, const std::vector<std::size_t> &lens, const std::vector<std::uint64_t> &coeffs,
                                const value_type &constant, const value_type &point) {
            if (ptrs.empty()) return;
            /* Q_normal = sum theta_acc g - sum theta_acc z;  Q_normal /= (X - point);  combined_Q_normal += Q_normal */
            check(zkhip_poly_lincomb_dev(_ctx.get(), adapter::id, ptrs.size(), ptrs.data(), lens.data(), coeffs.data(), 1, d_q.get(), max_len, 0),
                  "zkhip_poly_lincomb_dev", _ctx.get());
            std::uint64_t c[4], z[4], rem[4];
            adapter::scalar_to_limbs(value_type::zero() - constant, c);
            auto d_c = _ctx.alloc(32);
            _ctx.h2d(d_c.get(), c, 32);
            check(zkhip_fr_vec_op_dev(_ctx.get(), adapter::id, 0, d_q.get(), d_c.get(), d_q.get(), 1), "zkhip_fr_vec_op_dev", _ctx.get());
            adapter::scalar_to_limbs(point, z);
            check(zkhip_poly_div_linear_dev(_ctx.get(), adapter::id, d_q.get(), max_len, z, d_q.get(), rem), "zkhip_poly_div_linear_dev", _ctx.get());
            if (rem[0] | rem[1] | rem[2] | rem[3]) throw std::runtime_error("lpc proof_eval: a quotient does not divide (evaluation / point mismatch)");
            /* the quotient's max_len - 1 coefficients sit at d_q + 1; slot 0 held the (zero) remainder */
            void *q = static_cast<char *>(d_q.get()) + 32;
            if (!have_combined) {
                check(zkhip_memcpy_d2d_async(_ctx.get(), d_combined.get(), q, (max_len - 1) * 32), "zkhip_memcpy_d2d_async", _ctx.get());
                have_combined = true;
            } else if (max_len > 1) {
                check(zkhip_fr_vec_op_dev(_ctx.get(), adapter::id, 0, d_combined.get(), q, d_combined.get(), max_len - 1), "zkhip_fr_vec_op_dev", _ctx.get());
            }
            _ctx.sync();    // d_c is released on return
        };
        for (const auto &point : get_unique_points()) {
            std::vector<const void *> ptrs;
            std::vector<std::size_t> lens;
            std::vector<std::uint64_t> coeffs;
            value_type constant = value_type::zero();
            for (std::size_t i : _z.get_batches()) {
                for (std::size_t j = 0; j < _z.get_batch_size(i); ++j) {
                    const auto &pts = _points.at(i)[j];
                    auto it = std::find(pts.begin(), pts.end(), point);
                    if (it == pts.end()) continue;
                    ptrs.push_back(_dev.at(i).at(j));
                    lens.push_back(_dev.at(i).len[j]);
                    coeffs.resize(coeffs.size() + 4);
                    adapter::scalar_to_limbs(theta_acc, &coeffs[coeffs.size() - 4]);
                    constant = constant + _z.get(i, j, it - pts.begin()) * theta_acc;
                    theta_acc = theta_acc * theta;
                }
            }
            add_quotient(ptrs, lens, coeffs, constant, point);
        }
        for (std::size_t i : _z.get_batches()) {    // the fixed batches at etha (lpc.hpp:164-186)
            auto fx = _batch_fixed.find(i);
            if (fx == _batch_fixed.end() || !fx->second) continue;
            std::vector<const void *> ptrs;
            std::vector<std::size_t> lens;
            std::vector<std::uint64_t> coeffs;
            value_type constant = value_type::zero();
            for (std::size_t j = 0; j < _z.get_batch_size(i); ++j) {
                ptrs.push_back(_dev.at(i).at(j));
                lens.push_back(_dev.at(i).len[j]);
                coeffs.resize(coeffs.size() + 4);
                adapter::scalar_to_limbs(theta_acc, &coeffs[coeffs.size() - 4]);
                constant = constant + _fixed_polys_values.at(i).at(j) * theta_acc;
                theta_acc = theta_acc * theta;
            }
            add_quotient(ptrs, lens, coeffs, constant, _etha);
        }
        if (!have_combined) throw std::runtime_error("lpc proof_eval: nothing to open");
        lap("combined quotient");

        /* combined_Q.from_coefficients + precommit(combined_Q, D[0], step_list.front()) (lpc.hpp:188-198): the extension to
           D[0] is the NTT of the zero-padded coefficients */
        const std::size_t D0 = domain_size(0);
        device_polynomial_dfs<CurveType> f(_ctx, D0);
        {    /* the zero-padded copy in one pass on the device: f[j] = 1 * combined[j] below max_len - 1, zero behind */
            const void *src = d_combined.get();
            const std::size_t len = max_len - 1;
            std::uint64_t one[4];
            adapter::scalar_to_limbs(value_type::one(), one);
            check(zkhip_poly_lincomb_dev(_ctx.get(), adapter::id, 1, &src, &len, one, 1, f.data(), D0, 0), "zkhip_poly_lincomb_dev", _ctx.get());
        }
        {
            std::uint64_t w[4];
            adapter::scalar_to_limbs(_fri_params.root_of_unity(_fri_params.log_domain), w);
            check(zkhip_ntt_dev(_ctx.get(), adapter::id, f.data(), _fri_params.log_domain, 1, w, 0, nullptr), "zkhip_ntt_dev", _ctx.get());
        }
        lap("extension to D[0]");
        precommitment_type precommitment = [&]() {
            ZKHIP_PROFILE_SCOPE("Basic FRI Precommit time");    // precommit(combined_Q, ...), lpc.hpp:196-198
            return build_tree(f.data(), 1, _fri_params.log_domain, _fri_params.step_list.front());
        }();

        lap("precommit (leaves of round 0)");
        /* Commit phase (basic_fri.hpp:705-742) */
        proof_type proof;
        _fri_trees.clear();
        _fs.clear();
        _alphas.clear();
        std::size_t t = 0;
        for (std::size_t i = 0; i < _fri_params.step_list.size(); ++i) {
            _fs.push_back(f);
            _fri_trees.push_back(precommitment);
            proof.fri_proof.fri_roots.push_back(precommitment.root());
            transcript(precommitment.root());
            for (std::size_t step_i = 0; step_i < _fri_params.step_list[i]; ++step_i, ++t) {
                _alphas.push_back(transcript.challenge());
                f = fold_polynomial<CurveType>(f, _alphas[t], _fri_params.root_of_unity(_fri_params.log_domain - t));
            }
            if (i != _fri_params.step_list.size() - 1)
                precommitment = build_tree(f.data(), 1, _fri_params.log_domain - t, _fri_params.step_list[i + 1]);
        }
        _fs.push_back(f);
        lap("fold rounds + their leaves");
        {
            auto d_c = f.coefficients(_fri_params.root_of_unity);
            std::vector<std::uint64_t> h(4 * f.size());
            _ctx.d2h(h.data(), d_c.get(), h.size() * 8);
            for (std::size_t k = 0; k < f.size(); ++k) proof.fri_proof.final_polynomial.push_back(adapter::scalar_from_limbs(&h[4 * k]));
        }
        proof.z = _z;
        return proof;
    }

    // ---- what the caller's query phase reads (basic_fri.hpp:747-930) ----
    const std::map<std::size_t, precommitment_type> &trees() const { return _trees; }
    const std::vector<precommitment_type> &fri_trees() const { return _fri_trees; }
    const std::vector<value_type> &fri_alphas() const { return _alphas; }
    /// fs[i] of the commit phase (i <= step_list.size()): the round polynomials, evaluations over their domains
    polynomial_dfs<CurveType> fri_round_polynomial(std::size_t i) const { return _fs.at(i).to_host(); }
    /// coefficient form of committed polynomial (batch, index): `g_coeffs` of the query phase (basic_fri.hpp:753-771)
    std::vector<value_type> coefficients(std::size_t batch, std::size_t index) const {
        const device_batch &db = _dev.at(batch);
        std::vector<value_type> out;
        download_scalars<adapter>(_ctx, db.at(index), db.len.at(index), out);
        return out;
    }

protected:
    struct device_batch {
        std::shared_ptr<void> data;    // after commit: the coefficient forms
        std::vector<std::size_t> offset, len;
        void *at(std::size_t i) const { return static_cast<char *>(data.get()) + 32 * offset[i]; }
    };
    std::size_t domain_size(std::size_t t) const { return (std::size_t)1 << (_fri_params.log_domain - t); }
    static std::size_t log2_of(std::size_t n) {
        std::size_t l = 0;
        while (((std::size_t)1 << l) < n) ++l;
        return l;
    }
    /// the leaf layout of `batch` polynomials resident as evaluations over the 2^log_domain-point domain -> the caller's tree
    precommitment_type build_tree(const void *d_evals, std::size_t batch, std::size_t log_domain, std::size_t fri_step) const {
        const std::size_t D = (std::size_t)1 << log_domain, total = batch * D, per_leaf = batch * ((std::size_t)1 << fri_step);
        void *d_leaves = scratch(_scratch_leaves, _scratch_leaves_cap, std::max<std::size_t>(1, total) * 32);
        check(zkhip_fri_leaves_dev(_ctx.get(), d_evals, log_domain, batch, fri_step, d_leaves), "zkhip_fri_leaves_dev", _ctx.get());
        if constexpr (builder_kind == detail::tree_builder_kind::streaming) {
            /* slices of whole leaves through two page-locked buffers: the copy of slice k + 1 is in flight while the caller absorbs slice k */
            const std::size_t slice = std::max<std::size_t>(1, (leaf_slice_elements + per_leaf - 1) / std::max<std::size_t>(1, per_leaf)) * std::max<std::size_t>(1, per_leaf);
            _builder.begin(total, per_leaf);
            if (total != 0) {
                void *pin[2] = {_pin[0].reserve(_ctx, std::min(slice, total) * 32), _pin[1].reserve(_ctx, std::min(slice, total) * 32)};
                const char *src = static_cast<const char *>(d_leaves);
                _ctx.d2h_async(pin[0], src, std::min(slice, total) * 32);
                _ctx.sync();
                for (std::size_t at = 0, k = 0; at < total; at += slice, ++k) {
                    const std::size_t cnt = std::min(slice, total - at), next = at + slice;
                    if (next < total) _ctx.d2h_async(pin[(k + 1) & 1], src + 32 * next, std::min(slice, total - next) * 32);
                    _builder.absorb(host_values(pin[k & 1], cnt), at, cnt);
                    _ctx.sync();
                }
            }
            return _builder.finish();
        } else if constexpr (builder_kind == detail::tree_builder_kind::span) {
            void *pin = _pin[0].reserve(_ctx, std::max<std::size_t>(1, total) * 32);
            if (total != 0) {
                _ctx.d2h_async(pin, d_leaves, total * 32);
                _ctx.sync();
            }
            return _builder(host_values(pin, total), total, per_leaf);
        } else {
            _leaf_vec.clear();    // keeps its capacity: after the first commit no page of it is touched for the first time
            download_scalars<adapter>(_ctx, d_leaves, total, _leaf_vec);    // one copy for canonical-limb scalar types (backend.hpp)
            return _builder(_leaf_vec, per_leaf);
        }
    }
    /// the same tree from leaves that lie in `d_leaves.size()` consecutive blocks of `block` elements, block d on member d of the group: every
    /// owner's first slice is requested at once and its link stays one slice ahead of the caller, who absorbs the blocks in leaf order
    precommitment_type build_tree_group(const std::vector<const void *> &d_leaves, std::size_t block, std::size_t per_leaf) const {
        const device_group &group = *_group;
        const std::size_t owners = d_leaves.size(), total = owners * block;
        if constexpr (builder_kind == detail::tree_builder_kind::streaming) {
            const std::size_t slice = std::max<std::size_t>(1, (leaf_slice_elements + per_leaf - 1) / std::max<std::size_t>(1, per_leaf)) * std::max<std::size_t>(1, per_leaf);
            const std::size_t first = std::min(slice, block);
            _builder.begin(total, per_leaf);
            for (std::size_t d = 0; d < owners; ++d) {
                void *p0 = _gs[d].pin[0].reserve(group[d], first * 32);
                _gs[d].pin[1].reserve(group[d], first * 32);
                group[d].d2h_async(p0, d_leaves[d], first * 32);
            }
            for (std::size_t d = 0; d < owners; ++d) {
                const char *src = static_cast<const char *>(d_leaves[d]);
                void *pin[2] = {_gs[d].pin[0].get(), _gs[d].pin[1].get()};
                group[d].sync();
                for (std::size_t at = 0, k = 0; at < block; at += slice, ++k) {
                    const std::size_t cnt = std::min(slice, block - at), next = at + slice;
                    if (next < block) group[d].d2h_async(pin[(k + 1) & 1], src + 32 * next, std::min(slice, block - next) * 32);
                    _builder.absorb(host_values(pin[k & 1], cnt), d * block + at, cnt);
                    group[d].sync();
                }
            }
            return _builder.finish();
        } else {
            /* one page-locked buffer for all leaves (portable: every member's copy engine may write it), the owners' downloads side by side */
            char *pin = static_cast<char *>(_pin[0].reserve(_ctx, std::max<std::size_t>(1, total) * 32));
            for (std::size_t d = 0; d < owners; ++d) group[d].d2h_async(pin + 32 * d * block, d_leaves[d], block * 32);
            for (std::size_t d = 0; d < owners; ++d) group[d].sync();
            const value_type *values = host_values(pin, total);
            if constexpr (builder_kind == detail::tree_builder_kind::span) return _builder(values, total, per_leaf);
            else {
                _leaf_vec.assign(values, values + total);
                return _builder(_leaf_vec, per_leaf);
            }
        }
    }
    /// a member's device buffer that grows on demand and is kept across commits
    static void *grow(const context &ctx, std::shared_ptr<void> &buf, std::size_t &cap, std::size_t bytes) {
        if (bytes > cap) {
            ctx.sync();
            buf.reset();
            buf = ctx.alloc(bytes);
            cap = bytes;
        }
        return buf.get();
    }
    /// `count` canonical 32-byte elements in host memory as scalar-field values: in place when the scalar type IS four canonical
    /// limbs, through a converted copy (host threads) otherwise
    const value_type *host_values(const void *limbs, std::size_t count) const {
        if constexpr (detail::canonical_scalars<adapter>::value) {
            static_assert(sizeof(value_type) == 32, "canonical-limb scalars are 4 x u64");
            return static_cast<const value_type *>(limbs);
        } else {
            _conv.resize(count);
            const std::uint64_t *h = static_cast<const std::uint64_t *>(limbs);
            const std::size_t lanes = count >= ((std::size_t)1 << 16) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1;
            std::vector<std::future<void>> work;
            for (std::size_t k = 0; k < lanes; ++k)
                work.push_back(std::async(lanes > 1 ? std::launch::async : std::launch::deferred, [&, k]() {
                    for (std::size_t i = count * k / lanes; i < count * (k + 1) / lanes; ++i) _conv[i] = adapter::scalar_from_limbs(h + 4 * i);
                }));
            for (auto &w : work) w.get();
            return _conv.data();
        }
    }
    const context &upload_context() const {
        if (!_upload_ctx) _upload_ctx.reset(new context(_ctx.device()));
        return *_upload_ctx;
    }
    /// a device scratch buffer that grows on demand and is kept (every use is ordered on the context's stream)
    void *scratch(std::shared_ptr<void> &buf, std::size_t &cap, std::size_t bytes) const {
        if (bytes > cap) {
            _ctx.sync();
            buf.reset();
            buf = _ctx.alloc(bytes);
            cap = bytes;
        }
        return buf.get();
    }
    void own(std::size_t index, poly_type &&poly) {
        if (_locked[index]) throw std::runtime_error("append_to_batch: batch already committed");
        _owned[index].push_back(std::move(poly));
        _polys[index].push_back(&_owned[index].back());
    }
    void append_one(std::size_t index, const poly_type &p) { own(index, poly_type(p)); }
    void append_one(std::size_t index, std::reference_wrapper<const poly_type> p) { append_to_batch(index, p); }
    std::vector<value_type> evaluate_batch_at(std::size_t k, const value_type &x) const {
        const device_batch &db = _dev.at(k);
        std::vector<value_type> out(db.len.size());
        std::uint64_t pt[4];
        adapter::scalar_to_limbs(x, pt);
        for (std::size_t i = 0; i < db.len.size();) {
            std::size_t j = i;
            while (j < db.len.size() && db.len[j] == db.len[i]) ++j;
            std::vector<std::uint64_t> vals(4 * (j - i));
            check(zkhip_poly_eval_dev(_ctx.get(), adapter::id, db.at(i), db.len[i], db.len[i], j - i, pt, 1, vals.data()), "zkhip_poly_eval_dev", _ctx.get());
            for (std::size_t p = i; p < j; ++p) out[p] = adapter::scalar_from_limbs(&vals[4 * (p - i)]);
            i = j;
        }
        return out;
    }
    /// eval_polys (batched_commitment.hpp:168-183)
    void eval_polys() {
        for (const auto &it : _dev) {
            const std::size_t k = it.first;
            const device_batch &db = it.second;
            const auto &point = _points.at(k);
            _z.set_batch_size(k, db.len.size());
            std::vector<value_type> uni;
            for (const auto &pl : point)
                for (const auto &x : pl)
                    if (std::find(uni.begin(), uni.end(), x) == uni.end()) uni.push_back(x);
            std::vector<std::uint64_t> pts(4 * uni.size());
            for (std::size_t j = 0; j < uni.size(); ++j) adapter::scalar_to_limbs(uni[j], &pts[4 * j]);
            for (std::size_t i = 0; i < db.len.size();) {
                std::size_t j = i;
                while (j < db.len.size() && db.len[j] == db.len[i]) ++j;
                std::vector<std::uint64_t> vals(4 * (j - i) * uni.size());
                if (!uni.empty())
                    check(zkhip_poly_eval_dev(_ctx.get(), adapter::id, db.at(i), db.len[i], db.len[i], j - i, pts.data(), uni.size(), vals.data()),
                          "zkhip_poly_eval_dev", _ctx.get());
                for (std::size_t p = i; p < j; ++p) {
                    _z.set_poly_points_number(k, p, point[p].size());
                    for (std::size_t q = 0; q < point[p].size(); ++q) {
                        const std::size_t u = std::find(uni.begin(), uni.end(), point[p][q]) - uni.begin();
                        _z.set(k, p, q, adapter::scalar_from_limbs(&vals[4 * ((p - i) * uni.size() + u)]));
                    }
                }
                i = j;
            }
        }
    }
    /// get_unique_points (batched_commitment.hpp:113-129): first-seen order over batches, polynomials, points
    std::vector<value_type> get_unique_points() const {
        std::vector<value_type> result;
        for (const auto &it : _points)
            for (const auto &point_set : it.second)
                for (const auto &point : point_set)
                    if (std::find(result.begin(), result.end(), point) == result.end()) result.push_back(point);
        return result;
    }

    /// per member of the group, kept across commits: extensions, blocks to send, the owner's compact evaluations and its leaves
    struct group_scratch {
        std::shared_ptr<void> ext, send, cmp, leaves;
        std::size_t ext_cap = 0, send_cap = 0, cmp_cap = 0, leaves_cap = 0;
        pinned_buffer pin[2];
        std::unique_ptr<context> up;    // the member's upload stream
    };
    const context &_ctx;
    const device_group *_group = nullptr;
    std::size_t _group_commits = 0, _last_owners = 0;
    mutable std::unique_ptr<group_scratch[]> _gs;    // an array: the page-locked buffers neither copy nor move
    params_type _fri_params;
    mutable TreeBuilder _builder;
    value_type _etha;
    std::map<std::size_t, std::vector<const poly_type *>> _polys;    // in append order: copies held in _owned, or the caller's (lent)
    std::map<std::size_t, std::deque<poly_type>> _owned;
    mutable std::unique_ptr<context> _upload_ctx;
    mutable pinned_buffer _pin[2];
    mutable std::vector<value_type> _leaf_vec, _conv;
    mutable std::shared_ptr<void> _scratch_ext, _scratch_leaves;
    mutable std::size_t _scratch_ext_cap = 0, _scratch_leaves_cap = 0;
    std::map<std::size_t, bool> _locked, _batch_fixed;
    std::map<std::size_t, std::vector<std::vector<value_type>>> _points;
    std::map<std::size_t, device_batch> _dev;
    std::map<std::size_t, precommitment_type> _trees;
    preprocessed_data_type _fixed_polys_values;
    std::vector<precommitment_type> _fri_trees;
    std::vector<device_polynomial_dfs<CurveType>> _fs;
    std::vector<value_type> _alphas;
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_LPC_HPP
