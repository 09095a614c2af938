//---------------------------------------------------------------------------//
// zkhip shim: the batched KZG commitment scheme placeholder plugs in as `commitment_scheme_type`, on the MI355X.
//
// Mirrors zk/commitments/polynomial/kzg_v2.hpp (class kzg_commitment_scheme_v2, :56-360) on top of
// zk/commitments/batched_commitment.hpp (class polys_evaluator, :58-249) and
// zk/commitments/detail/polynomial/eval_storage.hpp:36-95 -- same member names, same call order:
//   append_to_batch / commit(batch) / append_eval_point[s] / set_batch_size / mark_batch_as_fixed /
//   preprocess / setup / proof_eval(transcript), public `_z`.
// What changes: every committed polynomial is transformed to coefficient form ONCE (commit: one batched inverse
// NTT, kzg.hpp:431) and stays resident; proof_eval's `_polys[k][i].coefficients()` (kzg_v2.hpp:253, 287, an
// inverse NTT per polynomial per use in the reference) reads those buffers; the two accumulations, the divisions
// by V and by (X - theta_2) and the two commitments run on the device.
//
// Transcript: the reference packs commitments and scalars to bytes (nil::marshalling, outside this tree) and
// hashes them.  Here TranscriptType is duck-typed on VALUES, the caller's adapter does the packing:
//   transcript(const single_commitment_type &), transcript(const scalar_value_type &),
//   scalar_value_type transcript.challenge().
// kzg_commitment_scheme_v2_hip::commit() returns the vector of single commitments; the class placeholder takes as
// `commitment_scheme_type` is kzg_commitment_scheme_v2_placeholder_hip below: its commit() returns the reference's byte
// blob (every single commitment through the caller's Packer, concatenated: kzg_v2.hpp:208-226 with nil::marshalling
// behind the functor) and its verify_eval() hands (proof, commitments, transcript) to the caller's Verifier (the
// pairing check of kzg_v2.hpp:312-395 is CPU-side reference code outside this backend).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_KZG_V2_HPP
#define ZKHIP_SHIM_KZG_V2_HPP

#include <algorithm>
#include <array>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <iterator>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "fri.hpp"
#include "kzg.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// eval_storage (eval_storage.hpp:36-95)
template <typename CurveType>
class eval_storage_hip {
    typedef typename curve_adapter<CurveType>::scalar_value_type value_type;
    std::map<std::size_t, std::vector<std::vector<value_type>>> z;

public:
    bool operator==(const eval_storage_hip &other) const { return z == other.z; }
    std::vector<std::size_t> get_batches() const {
        std::vector<std::size_t> b;
        for (const auto &it : z) b.push_back(it.first);
        return b;
    }
    std::size_t get_batches_num() const { return z.size(); }
    std::size_t get_batch_size(std::size_t batch_id) const { return z.at(batch_id).size(); }
    std::size_t get_poly_points_number(std::size_t batch_id, std::size_t poly_id) const { return z.at(batch_id)[poly_id].size(); }
    const std::vector<std::vector<value_type>> &get(std::size_t batch_id) const { return z.at(batch_id); }
    const std::vector<value_type> &get(std::size_t batch_id, std::size_t poly_id) const { return z.at(batch_id)[poly_id]; }
    const value_type &get(std::size_t batch_id, std::size_t poly_id, std::size_t point_id) const { return z.at(batch_id)[poly_id][point_id]; }
    void set_batch_size(std::size_t batch_id, std::size_t batch_size) { z[batch_id].assign(batch_size, {}); }
    void set_poly_points_number(std::size_t batch_id, std::size_t poly_id, std::size_t n) { z[batch_id][poly_id].assign(n, value_type::zero()); }
    void set(std::size_t batch_id, std::size_t poly_id, std::size_t point_id, const value_type &v) { z[batch_id][poly_id][point_id] = v; }
};

namespace detail {
    /// dense host polynomials of a few coefficients (U, V, diffpoly): math::polynomial as far as proof_eval uses it
    template <typename S>
    struct small_poly {
        typedef std::vector<S> P;
        static P mul(const P &a, const P &b) {
            if (a.empty() || b.empty()) return {};
            P out(a.size() + b.size() - 1, S::zero());
            for (std::size_t i = 0; i < a.size(); ++i)
                for (std::size_t j = 0; j < b.size(); ++j) out[i + j] = out[i + j] + a[i] * b[j];
            return out;
        }
        static P add(const P &a, const P &b) {
            P out(std::max(a.size(), b.size()), S::zero());
            for (std::size_t i = 0; i < a.size(); ++i) out[i] = out[i] + a[i];
            for (std::size_t i = 0; i < b.size(); ++i) out[i] = out[i] + b[i];
            return out;
        }
        static P scale(const P &a, const S &c) {
            P out(a);
            for (auto &x : out) x = x * c;
            return out;
        }
        static S evaluate(const P &a, const S &x) {
            S acc = S::zero();
            for (std::size_t i = a.size(); i-- > 0;) acc = acc * x + a[i];
            return acc;
        }
        /// get_V (batched_commitment.hpp:79-87)
        static P vanishing(const std::vector<S> &points) {
            P v {S::one()};
            for (const auto &x : points) v = mul(v, P {S::zero() - x, S::one()});
            return v;
        }
        /// math::lagrange_interpolation as get_U uses it (batched_commitment.hpp:100-111)
        static P lagrange(const std::vector<S> &xs, const std::vector<S> &ys) {
            P out;
            for (std::size_t k = 0; k < xs.size(); ++k) {
                P num {S::one()};
                S den = S::one();
                for (std::size_t j = 0; j < xs.size(); ++j) {
                    if (j == k) continue;
                    num = mul(num, P {S::zero() - xs[j], S::one()});
                    den = den * (xs[k] - xs[j]);
                }
                out = add(out, scale(num, ys[k] * den.inversed()));
            }
            return out;
        }
    };
    template <typename S>
    bool limbs_less(const S &a, const S &b) {
        for (int i = 3; i >= 0; --i)
            if (a.limbs[i] != b.limbs[i]) return a.limbs[i] < b.limbs[i];
        return false;
    }
}    // namespace detail

/// algorithms::proof_eval<KZG>(params, f, z) of the basic scheme (kzg.hpp:155-172): commit((f - f(z)) / (X - z)),
/// f given by its coefficients.
template <typename CurveType>
typename curve_adapter<CurveType>::g1_value_type kzg_proof_eval(const kzg_params_hip<CurveType> &params,
                                                                const std::vector<typename curve_adapter<CurveType>::scalar_value_type> &f,
                                                                const typename curve_adapter<CurveType>::scalar_value_type &z) {
    typedef curve_adapter<CurveType> adapter;
    const context &ctx = params.ctx;
    if (f.size() > params.commitment_key.size() + 1) throw std::runtime_error("kzg_proof_eval: polynomial longer than the commitment key");
    if (f.size() <= 1) return adapter::g1_value_type::zero();
    std::vector<std::uint64_t> host(4 * f.size());
    for (std::size_t i = 0; i < f.size(); ++i) adapter::scalar_to_limbs(f[i], &host[4 * i]);
    auto d = ctx.alloc(host.size() * 8);
    ctx.h2d(d.get(), host.data(), host.size() * 8);
    std::uint64_t zl[4];
    adapter::scalar_to_limbs(z, zl);
    /* q[0] -= f(z); q /= (X - z): the synthetic division drops the remainder f(z) by itself (kzg.hpp:163-169) */
    check(zkhip_poly_div_linear_dev(ctx.get(), adapter::id, d.get(), f.size(), zl, d.get(), nullptr), "zkhip_poly_div_linear_dev", ctx.get());
    return multiexp_dev<CurveType, ZKHIP_G1>(ctx, params.commitment_key, 0, f.size() - 1, static_cast<const char *>(d.get()) + 32);
}

/// What kzg_commitment_scheme (kzg.hpp:636-873) and kzg_commitment_scheme_v2 (kzg_v2.hpp:56-360) share: the polys_evaluator
/// state (batched_commitment.hpp:58-249), commit(batch) with the coefficient forms left resident, eval_polys, get_U,
/// the transcript update and the device helpers of the opening proofs.
/// `PolynomialType` as in the reference's polys_evaluator (batched_commitment.hpp:56-64; default: the shim's own polynomial_dfs):
/// anything with size() and operator[] over contiguous scalar_value_type -- math::polynomial_dfs as placeholder_prover hands it
/// over (prover.hpp:137-138, 316) included.
template <typename CurveType, typename TranscriptType, typename PolynomialType = polynomial_dfs<CurveType>>
class kzg_polys_evaluator_hip {
protected:
    /// a committed batch on the device: the coefficient forms of its polynomials, one behind the other
    struct device_batch {
        std::shared_ptr<void> data;
        std::vector<std::size_t> offset, len;    // in elements
        void *at(std::size_t i) const { return static_cast<char *>(data.get()) + 32 * offset[i]; }
    };

public:
    typedef curve_adapter<CurveType> adapter;
    typedef CurveType curve_type;
    typedef typename adapter::scalar_value_type scalar_value_type;
    typedef typename adapter::g1_value_type single_commitment_type;
    typedef std::vector<single_commitment_type> commitment_type;
    typedef TranscriptType transcript_type;
    typedef kzg_params_hip<CurveType> params_type;
    typedef PolynomialType poly_type;
    typedef eval_storage_hip<CurveType> eval_storage_type;
    typedef bool preprocessed_data_type;
    /// primitive 2^log_n-th root of unity of the evaluation domains (math::make_evaluation_domain's choice)
    typedef std::function<scalar_value_type(std::size_t log_n)> root_of_unity_type;

    eval_storage_type _z;

    kzg_polys_evaluator_hip(const params_type &kzg_params, root_of_unity_type root_of_unity) :
        _params(kzg_params), _root_of_unity(std::move(root_of_unity)) { }
    /// the reference's constructor argument list (kzg.hpp:667, kzg_v2.hpp:94: the parameters alone): the roots of unity from the curve adapter
    explicit kzg_polys_evaluator_hip(const params_type &kzg_params) :
        _params(kzg_params), _root_of_unity([](std::size_t log_n) { return adapter::root_of_unity(log_n); }) { }
    /// Over a DEVICE GROUP: commit(batch) deals the batch's columns over the group's members (each transforms and commits its columns
    /// against its replica of the key; only commitments and -- device to device -- the coefficient forms travel to member 0), the rest
    /// of the scheme runs on member 0.  The reference's commit is a loop over the batch inside the call (kzg_v2.hpp:208-226): so is this.
    typedef kzg_params_group_hip<CurveType> group_params_type;
    kzg_polys_evaluator_hip(const group_params_type &group_params, root_of_unity_type root_of_unity) :
        _params(group_params.root()), _root_of_unity(std::move(root_of_unity)), _group_params(&group_params) { }
    explicit kzg_polys_evaluator_hip(const group_params_type &group_params) :
        _params(group_params.root()), _root_of_unity([](std::size_t log_n) { return adapter::root_of_unity(log_n); }), _group_params(&group_params) { }

    const params_type &get_commitment_params() const { return _params; }
    preprocessed_data_type preprocess(transcript_type &) const { return true; }
    void setup(transcript_type &, preprocessed_data_type = true) { }
    void mark_batch_as_fixed(std::size_t) { }

    // ---- polys_evaluator (batched_commitment.hpp:197-247) ----
    /// as the reference: the scheme keeps a COPY of the polynomial (batched_commitment.hpp:197-206)
    void append_to_batch(std::size_t index, const poly_type &poly) { own(index, poly_type(poly)); }
    template <typename ContainerType>
    void append_to_batch(std::size_t index, const ContainerType &polys) {
        for (const auto &p : polys) append_one(index, p);
    }
    /// a caller that is done with the polynomials hands them over instead (50 x 2^20 rows: 340 ms of host copying saved) ...
    void append_to_batch(std::size_t index, poly_type &&poly) { own(index, std::move(poly)); }
    void append_to_batch(std::size_t index, std::vector<poly_type> &&polys) {
        for (auto &p : polys) own(index, std::move(p));
        polys.clear();
    }
    /// ... and one that keeps them alive until commit(index) returns LENDS them: std::cref(poly), or a container of
    /// std::reference_wrapper<const poly_type> -- nothing is copied on the host, commit reads the caller's vectors where they lie
    void append_to_batch(std::size_t index, std::reference_wrapper<const poly_type> poly) {
        if (_locked[index]) throw std::runtime_error("append_to_batch: batch already committed");
        _polys[index].push_back(&poly.get());
    }
    /// A polynomial that is ALREADY resident (device_polynomial_dfs, fri.hpp: the quotient parts of placeholder_quotient.hpp, a
    /// witness column built on the device) joins the batch where it lies: commit copies it device-to-device instead of crossing PCIe.
    /// Not in the reference's interface -- an addition for callers that keep their columns on the GPU.
    void append_to_batch(std::size_t index, const device_polynomial_dfs<CurveType> &poly) {
        if (_locked[index]) throw std::runtime_error("append_to_batch: batch already committed");
        _resident[index].emplace(_polys[index].size(), poly);
        _polys[index].push_back(nullptr);
    }
    void append_to_batch(std::size_t index, const std::vector<device_polynomial_dfs<CurveType>> &polys) {
        for (const auto &p : polys) append_to_batch(index, p);
    }
    /// A resident polynomial that is ALREADY in coefficient form (placeholder's quotient parts: placeholder_quotient_hip::
    /// quotient_polynomial_split_coefficients): commit skips its inverse transform.  Same commitment, same opening proof as for the
    /// polynomial_dfs the reference would have made of it.
    void append_to_batch(std::size_t index, const device_polynomial_coefficients<CurveType> &poly) {
        if (_locked[index]) throw std::runtime_error("append_to_batch: batch already committed");
        _resident_coefficients[index].emplace(_polys[index].size(), poly);
        _polys[index].push_back(nullptr);
    }
    void append_to_batch(std::size_t index, const std::vector<device_polynomial_coefficients<CurveType>> &polys) {
        for (const auto &p : polys) append_to_batch(index, p);
    }
    void append_eval_point(std::size_t batch_id, const scalar_value_type &point) {
        for (auto &pts : _points.at(batch_id)) pts.push_back(point);
    }
    void append_eval_point(std::size_t batch_id, std::size_t poly_id, const scalar_value_type &point) { _points.at(batch_id).at(poly_id).push_back(point); }
    void append_eval_points(std::size_t batch_id, const std::vector<scalar_value_type> &points) {
        for (auto &pts : _points.at(batch_id)) pts.insert(pts.end(), points.begin(), points.end());
    }
    void append_eval_points(std::size_t batch_id, std::size_t poly_id, const std::vector<scalar_value_type> &points) {
        auto &pts = _points.at(batch_id).at(poly_id);
        pts.insert(pts.end(), points.begin(), points.end());
    }
    void set_batch_size(std::size_t batch_id, std::size_t batch_size) {
        _points[batch_id].resize(batch_size);
        _locked[batch_id] = true;
    }

    /// commit(index) (kzg_v2.hpp:208-226): one commitment per polynomial of the batch; the coefficient forms stay
    /// on the device for proof_eval.
    commitment_type commit(std::size_t index) {
        if (_group_params && _group_params->members.size() > 1) return commit_group(index);
        const context &ctx = _params.ctx;
        const std::vector<const poly_type *> &polys = _polys[index];
        const auto &resident = _resident[index];    // position -> a polynomial that is already on the device (polys[position] == nullptr)
        const auto &resident_c = _resident_coefficients[index];    // ... and already in coefficient form
        auto is_coefficients = [&resident_c](std::size_t i) { return resident_c.count(i) != 0; };
        auto is_zero = [&resident_c](std::size_t i) {    // known to be the zero polynomial: its commitment is the neutral element, no multiexp
            auto it = resident_c.find(i);
            return it != resident_c.end() && it->second.known_zero();
        };
        device_batch db;
        std::size_t total = 0;
        for (std::size_t i = 0; i < polys.size(); ++i) {
            const std::size_t sz = polys[i] ? polys[i]->size() : is_coefficients(i) ? resident_c.at(i).size() : resident.at(i).size();
            if (sz == 0 || (sz & (sz - 1))) throw std::runtime_error("commit: polynomial_dfs size must be a power of two");
            if (sz > _params.commitment_key.size()) throw std::runtime_error("commit: polynomial longer than the commitment key");
            db.offset.push_back(total);
            db.len.push_back(sz);
            total += sz;
        }
        db.data = ctx.alloc(std::max<std::size_t>(1, total) * 32);
        const std::size_t count = polys.size(), jl = 3 * adapter::g1_coord_limbs;
        auto d_res = ctx.alloc(std::max<std::size_t>(1, count) * jl * 8);
        /* The polynomials go up in chunks on a second in-order stream: while chunk c is transformed (p.coefficients(), kzg.hpp:431:
           one batched inverse NTT per chunk of equally sized polynomials) and committed (multiexp against the resident key, the
           chunk's bucket reductions sharing one launch), chunk c + 1 crosses PCIe. */
        bool any_from_host = false;    // a batch that is resident already has nothing to overlap: one chunk, one shared tail over all its multiexps
        for (const poly_type *p : polys) any_from_host = any_from_host || p != nullptr;
        const bool pipelined = upload_chunk != 0 && count > upload_chunk && any_from_host;
        const context &up = pipelined ? upload_context() : ctx;
        /* a resident polynomial may still be in the making on ctx's stream (polynomial_product, resize do not synchronise): the stream that
           copies it waits for ctx once */
        if (pipelined && (!resident.empty() || !resident_c.empty())) up.wait_for(ctx);
        for (std::size_t i = 0; i < count;) {
            std::size_t j = i;
            /* the first chunk is a short one: nothing runs on the device until it has arrived */
            const std::size_t limit = (pipelined && i == 0) ? std::max<std::size_t>(1, upload_chunk / 4) : upload_chunk;
            while (j < count && db.len[j] == db.len[i] && is_coefficients(j) == is_coefficients(i) && is_zero(j) == is_zero(i) &&
                   (!pipelined || limit == 0 || j - i < limit))
                ++j;
            for (std::size_t p = i; p < j; ++p) {
                if (polys[p]) upload_scalars<adapter>(up, db.at(p), detail::poly_data<adapter>(*polys[p]), polys[p]->size());
                else
                    check(zkhip_memcpy_d2d_async(up.get(), db.at(p), is_coefficients(p) ? resident_c.at(p).data() : resident.at(p).data(), db.len[p] * 32),
                          "zkhip_memcpy_d2d_async", up.get());
            }
            if (pipelined) ctx.wait_for(up);
            if (!is_coefficients(i)) {
                std::size_t log_n = 0;
                while (((std::size_t)1 << log_n) < db.len[i]) ++log_n;
                std::uint64_t w[4];
                adapter::scalar_to_limbs(_root_of_unity(log_n), w);
                check(zkhip_ntt_dev(ctx.get(), adapter::id, db.at(i), log_n, j - i, w, 1, nullptr), "zkhip_ntt_dev", ctx.get());
            }
            if (!is_zero(i)) commit_resident(db, i, j, d_res.get());    // the coefficients of a known-zero polynomial still go into db (proof_eval reads them)
            i = j;
        }
        commitment_type out;
        if (count) {
            std::vector<std::uint64_t> res(count * jl);
            ctx.d2h(res.data(), d_res.get(), res.size() * 8);
            for (std::size_t i = 0; i < count; ++i) out.push_back(is_zero(i) ? single_commitment_type::zero() : adapter::g1_from_jacobian(&res[i * jl]));
        }
        return commit_done(index, std::move(db), std::move(out));
    }

private:
    /// state_commited (batched_commitment.hpp:163-166).  The host polynomials are not read again: lent ones may go; copies and
    /// handed-over ones are released with the scheme (freeing gigabytes of host memory here costs ~100 ms: more than the upload)
    commitment_type commit_done(std::size_t index, device_batch &&db, commitment_type &&out) {
        const std::size_t count = _polys[index].size();
        _ind_commitments[index] = std::move(out);
        _dev[index] = std::move(db);
        _locked[index] = true;
        _points[index].resize(count);
        _polys[index].clear();    // no pointer to a lent polynomial outlives the call
        _resident[index].clear();
        _resident_coefficients[index].clear();
        return _ind_commitments[index];
    }

    /// commit(index) over the device group: member k takes the k-th contiguous range of the batch's columns.  Resident polynomials (they
    /// live on member 0's GPU) are dealt first, device to device, from this thread; then every member runs on a host thread of its own --
    /// uploads of its host columns (the members' PCIe links work side by side), one batched inverse transform per run of equal sizes, its
    /// multiexps as one batch, one small download of its commitments -- and what is left for this thread is to pull the coefficient forms
    /// of the other members into member 0's batch buffer, where proof_eval expects them.
    commitment_type commit_group(std::size_t index) {
        const group_params_type &gp = *_group_params;
        const device_group &group = gp.group;
        const std::size_t world = gp.members.size(), jl = 3 * adapter::g1_coord_limbs;
        const context &root = _params.ctx;
        const std::vector<const poly_type *> &polys = _polys[index];
        const auto &resident = _resident[index];
        const auto &resident_c = _resident_coefficients[index];
        auto is_coefficients = [&resident_c](std::size_t i) { return resident_c.count(i) != 0; };
        auto is_zero = [&resident_c](std::size_t i) {
            auto it = resident_c.find(i);
            return it != resident_c.end() && it->second.known_zero();
        };
        const std::size_t count = polys.size();
        device_batch db;
        std::size_t total = 0;
        for (std::size_t i = 0; i < count; ++i) {
            const std::size_t sz = polys[i] ? polys[i]->size() : is_coefficients(i) ? resident_c.at(i).size() : resident.at(i).size();
            if (sz == 0 || (sz & (sz - 1))) throw std::runtime_error("commit: polynomial_dfs size must be a power of two");
            if (sz > _params.commitment_key.size()) throw std::runtime_error("commit: polynomial longer than the commitment key");
            db.offset.push_back(total);
            db.len.push_back(sz);
            total += sz;
        }
        db.data = root.alloc(std::max<std::size_t>(1, total) * 32);
        /* member k's columns [lo, hi): a contiguous range, so its coefficient forms are ONE contiguous piece of the batch buffer */
        struct part {
            std::size_t lo = 0, hi = 0;
            std::shared_ptr<void> data, d_res;    // data: the member's own buffer (member 0 works in the batch buffer itself)
            char *base = nullptr;
            std::vector<std::uint64_t> res;
        };
        std::vector<part> parts(world);
        for (std::size_t k = 0; k < world; ++k) {
            part &pt = parts[k];
            pt.lo = count / world * k + std::min(k, count % world);
            pt.hi = count / world * (k + 1) + std::min(k + 1, count % world);
            if (pt.hi == pt.lo) continue;
            const std::size_t elems = (pt.hi < count ? db.offset[pt.hi] : total) - db.offset[pt.lo];
            if (k == 0) pt.base = static_cast<char *>(db.at(pt.lo));
            else {
                pt.data = group[k].alloc(elems * 32);
                pt.base = static_cast<char *>(pt.data.get());
            }
            pt.d_res = group[k].alloc((pt.hi - pt.lo) * jl * 8);
            pt.res.resize((pt.hi - pt.lo) * jl);
            /* resident polynomials: off member 0's GPU, in the order of this (the group's) thread */
            for (std::size_t p = pt.lo; p < pt.hi; ++p) {
                if (polys[p]) continue;
                const void *src = is_coefficients(p) ? resident_c.at(p).data() : resident.at(p).data();
                group.copy(k, pt.base + 32 * (db.offset[p] - db.offset[pt.lo]), 0, src, db.len[p] * 32);
            }
        }
        /* the caller's root-of-unity function is called from THIS thread only (it need not be thread-safe): one root per distinct size, up front */
        std::map<std::size_t, std::array<std::uint64_t, 4>> root_limbs;
        for (std::size_t i = 0; i < count; ++i) {
            if (is_coefficients(i)) continue;
            std::size_t log_n = 0;
            while (((std::size_t)1 << log_n) < db.len[i]) ++log_n;
            if (!root_limbs.count(log_n)) adapter::scalar_to_limbs(_root_of_unity(log_n), root_limbs[log_n].data());
        }
        auto member_work = [&](std::size_t k) {
            part &pt = parts[k];
            if (pt.hi == pt.lo) return;
            const context &ctx = group[k];
            const kzg_params_hip<CurveType> &params = *gp.members[k];
            auto at = [&](std::size_t p) { return pt.base + 32 * (db.offset[p] - db.offset[pt.lo]); };
            /* the member's own pipeline, as commit() runs it on one device: its host columns go up in chunks on a second in-order stream of
               ITS GPU while the chunk before is transformed and committed (a third of the member's columns per chunk, at most upload_chunk) */
            bool any_from_host = false;
            for (std::size_t p = pt.lo; p < pt.hi; ++p) any_from_host = any_from_host || polys[p] != nullptr;
            const std::size_t mine = pt.hi - pt.lo, chunk = upload_chunk ? std::max<std::size_t>(1, std::min(upload_chunk, (mine + 2) / 3)) : 0;
            const bool pipelined = chunk != 0 && mine > chunk && any_from_host;
            std::unique_ptr<context> up_own;
            if (pipelined) up_own.reset(new context(ctx.device()));
            const context &up = pipelined ? *up_own : ctx;
            for (std::size_t i = pt.lo; i < pt.hi;) {
                std::size_t j = i;
                while (j < pt.hi && db.len[j] == db.len[i] && is_coefficients(j) == is_coefficients(i) && is_zero(j) == is_zero(i) && (!pipelined || j - i < chunk)) ++j;
                for (std::size_t p = i; p < j; ++p)
                    if (polys[p]) upload_scalars<adapter>(up, at(p), detail::poly_data<adapter>(*polys[p]), polys[p]->size());
                if (pipelined) ctx.wait_for(up);
                if (!is_coefficients(i)) {
                    std::size_t log_n = 0;
                    while (((std::size_t)1 << log_n) < db.len[i]) ++log_n;
                    check(zkhip_ntt_dev(ctx.get(), adapter::id, at(i), log_n, j - i, root_limbs.at(log_n).data(), 1, nullptr), "zkhip_ntt_dev", ctx.get());
                }
                if (!is_zero(i)) {
                    const std::size_t cnt = j - i;
                    std::vector<const zkhip_bases *> qb(cnt, params.commitment_key.get());
                    std::vector<std::size_t> qo(cnt, 0), qn(db.len.begin() + i, db.len.begin() + j);
                    std::vector<const void *> qs(cnt);
                    std::vector<void *> qr(cnt);
                    for (std::size_t c = 0; c < cnt; ++c) {
                        qs[c] = at(i + c);
                        qr[c] = static_cast<std::uint64_t *>(pt.d_res.get()) + (i + c - pt.lo) * jl;
                    }
                    check(zkhip_msm_batch_dev(ctx.get(), cnt, qb.data(), qo.data(), qn.data(), qs.data(), qr.data()), "zkhip_msm_batch_dev", ctx.get());
                }
                i = j;
            }
            ctx.d2h(pt.res.data(), pt.d_res.get(), pt.res.size() * 8);    // synchronises this member's stream (its upload stream has drained: every upload is a synchronous copy)
        };
        {
            std::vector<std::future<void>> others;
            for (std::size_t k = 1; k < world; ++k) others.push_back(std::async(std::launch::async, member_work, k));
            std::exception_ptr failed;
            try {
                member_work(0);
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
        /* the other members' coefficient forms join the batch on member 0 (xGMI peer copies; proof_eval reads them there) */
        for (std::size_t k = 1; k < world; ++k) {
            const part &pt = parts[k];
            if (pt.hi == pt.lo) continue;
            const std::size_t elems = (pt.hi < count ? db.offset[pt.hi] : total) - db.offset[pt.lo];
            group.copy(0, db.at(pt.lo), k, pt.base, elems * 32);
        }
        root.sync();    // the copies ran on member 0's stream: the members' buffers may go
        commitment_type out;
        for (std::size_t k = 0; k < world; ++k)
            for (std::size_t p = parts[k].lo; p < parts[k].hi; ++p)
                out.push_back(is_zero(p) ? single_commitment_type::zero() : adapter::g1_from_jacobian(&parts[k].res[(p - parts[k].lo) * jl]));
        return commit_done(index, std::move(db), std::move(out));
    }

public:

    /// polynomials per upload chunk (0: the whole batch in one transfer, one shared bucket reduction)
    std::size_t upload_chunk = 10;
    /// over a device group: the quotient length from which proof_eval's multiexps are cut over the members (shorter ones stay on member 0)
    std::size_t group_commit_min = (std::size_t)1 << 15;
    /// quotient commitments that were cut over the group so far (tests and logs)
    std::size_t group_multiexps() const { return _group_multiexps; }

    const std::map<std::size_t, commitment_type> &commitments() const { return _ind_commitments; }

protected:
    /// multiexp(commitment_key[0 .. len), coefficients) for polynomials [first, last) of a resident batch, as one device batch
    /// (enqueued only); results: Jacobian points at d_res + i * 3 * coordinate limbs
    void commit_resident(const device_batch &db, std::size_t first, std::size_t last, void *d_res) const {
        const context &ctx = _params.ctx;
        const std::size_t count = last - first, jl = 3 * adapter::g1_coord_limbs;
        if (count == 0) return;
        std::vector<const zkhip_bases *> qb(count, _params.commitment_key.get());
        std::vector<std::size_t> qo(count, 0), qn(db.len.begin() + first, db.len.begin() + last);
        std::vector<const void *> qs(count);
        std::vector<void *> qr(count);
        for (std::size_t i = 0; i < count; ++i) {
            qs[i] = db.at(first + i);
            qr[i] = static_cast<std::uint64_t *>(d_res) + (first + i) * jl;
        }
        check(zkhip_msm_batch_dev(ctx.get(), count, qb.data(), qo.data(), qn.data(), qs.data(), qr.data()), "zkhip_msm_batch_dev", ctx.get());
    }
    /// the second stream the uploads of commit() ride on (created on first use, same GPU)
    const context &upload_context() const {
        if (!_upload_ctx) _upload_ctx.reset(new context(_params.ctx.device()));
        return *_upload_ctx;
    }
    void own(std::size_t index, poly_type &&poly) {
        if (_locked[index]) throw std::runtime_error("append_to_batch: batch already committed");
        _owned[index].push_back(std::move(poly));
        _polys[index].push_back(&_owned[index].back());
    }
    void append_one(std::size_t index, const poly_type &p) { own(index, poly_type(p)); }
    void append_one(std::size_t index, std::reference_wrapper<const poly_type> p) { append_to_batch(index, p); }
    single_commitment_type commit_range(const void *d_coeffs, std::size_t len) const {
        if (len == 0) return single_commitment_type::zero();
        if (len > _params.commitment_key.size()) throw std::runtime_error("proof_eval: quotient longer than the commitment key");
        if (_group_params && _group_params->members.size() > 1 && len >= std::max<std::size_t>(group_commit_min, _group_params->members.size()))
            return commit_range_group(d_coeffs, len);
        return multiexp_dev<CurveType, ZKHIP_G1>(_params.ctx, _params.commitment_key, 0, len, d_coeffs);
    }
    /// proof_eval's quotient commitments over the device group: the coefficients (on member 0) are cut by point range, every other member pulls
    /// its range device to device and runs it against ITS replica of the key at the same offset; the Jacobian partial sums meet on member 0
    /// (the group's all-gather), are folded there and downloaded once -- the reference's multiexp with `chunks` = the members (kzg.hpp:146-147)
    single_commitment_type commit_range_group(const void *d_coeffs, std::size_t len) const {
        const group_params_type &gp = *_group_params;
        const device_group &group = gp.group;
        const std::size_t world = gp.members.size(), jl = 3 * adapter::g1_coord_limbs;
        std::vector<std::shared_ptr<void>> d_sc(world), d_part(world);
        std::vector<const void *> send(world);
        std::vector<void *> recv(world, nullptr);
        for (std::size_t k = 0; k < world; ++k) {
            const std::size_t lo = len / world * k + std::min(k, len % world), hi = len / world * (k + 1) + std::min(k + 1, len % world);
            const void *sc = static_cast<const char *>(d_coeffs) + 32 * lo;
            d_part[k] = group[k].alloc(jl * 8);
            if (k != 0) {
                d_sc[k] = group[k].alloc((hi - lo) * 32);
                group.copy(k, d_sc[k].get(), 0, sc, (hi - lo) * 32);
                sc = d_sc[k].get();
            }
            check(zkhip_msm_dev(group[k].get(), gp.members[k]->commitment_key.get(), lo, hi - lo, sc, d_part[k].get()), "zkhip_msm_dev", group[k].get());
            send[k] = d_part[k].get();
        }
        const context &root = group[0];
        auto d_all = root.alloc((world + 1) * jl * 8);
        recv[0] = d_all.get();
        group.all_gather(send, recv, jl * 8);
        void *d_sum = static_cast<char *>(d_all.get()) + world * jl * 8;
        check(zkhip_jacobian_sum_dev(root.get(), adapter::id, ZKHIP_G1, d_all.get(), world, d_sum), "zkhip_jacobian_sum_dev", root.get());
        std::uint64_t jac[3 * adapter::g1_coord_limbs];
        root.d2h(jac, d_sum, sizeof(jac));
        group.sync();    // the members' scalar ranges and partial sums are released on return
        ++_group_multiexps;
        return adapter::g1_from_jacobian(jac);
    }
    /// d[0 .. small.size()) += small
    void add_low_coefficients(void *d, const std::vector<scalar_value_type> &small, std::size_t len) const {
        const context &ctx = _params.ctx;
        const std::size_t n = std::min(small.size(), len);
        if (n == 0) return;
        std::vector<std::uint64_t> h(4 * n);
        for (std::size_t i = 0; i < n; ++i) adapter::scalar_to_limbs(small[i], &h[4 * i]);
        auto tmp = ctx.alloc(n * 32);
        ctx.h2d(tmp.get(), h.data(), h.size() * 8);
        check(zkhip_fr_vec_op_dev(ctx.get(), adapter::id, 0, d, tmp.get(), d, n), "zkhip_fr_vec_op_dev", ctx.get());
        ctx.sync();
    }
    /// (ptr, len) <- quotient by (X - root); throws when the remainder is not zero
    void divide_in_place(char *&ptr, std::size_t &len, const scalar_value_type &root, const char *what) const {
        if (len == 0) return;
        std::uint64_t zl[4], rem[4];
        adapter::scalar_to_limbs(root, zl);
        check(zkhip_poly_div_linear_dev(_params.ctx.get(), adapter::id, ptr, len, zl, ptr, rem), "zkhip_poly_div_linear_dev", _params.ctx.get());
        if (rem[0] | rem[1] | rem[2] | rem[3]) throw std::runtime_error(what);
        ptr += 32;
        len -= 1;
    }

    /// eval_polys (batched_commitment.hpp:168-183): every polynomial of a batch at the union of the batch's points
    void eval_polys() {
        const context &ctx = _params.ctx;
        for (const auto &it : _dev) {
            const std::size_t k = it.first;
            const device_batch &db = it.second;
            const auto &point = _points.at(k);
            _z.set_batch_size(k, db.len.size());
            std::vector<scalar_value_type> uni;
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
                    check(zkhip_poly_eval_dev(ctx.get(), adapter::id, db.at(i), db.len[i], db.len[i], j - i, pts.data(), uni.size(), vals.data()),
                          "zkhip_poly_eval_dev", ctx.get());
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
    /// merge_eval_points (kzg_v2.hpp:121-130)
    void merge_eval_points() {
        _merged_points.clear();
        for (const auto &it : _points)
            for (const auto &pl : it.second)
                for (const auto &x : pl)
                    if (std::find(_merged_points.begin(), _merged_points.end(), x) == _merged_points.end()) _merged_points.push_back(x);
        std::sort(_merged_points.begin(), _merged_points.end(), detail::limbs_less<scalar_value_type>);
    }
    /// set_difference_polynom (kzg_v2.hpp:132-148)
    std::vector<scalar_value_type> set_difference_polynom(const std::vector<scalar_value_type> &points) const {
        std::vector<scalar_value_type> rest;
        for (const auto &x : _merged_points)
            if (std::find(points.begin(), points.end(), x) == points.end()) rest.push_back(x);
        return detail::small_poly<scalar_value_type>::vanishing(rest);
    }
    /// get_U (batched_commitment.hpp:100-111)
    std::vector<scalar_value_type> get_U(std::size_t b_ind, std::size_t poly_ind) const {
        return detail::small_poly<scalar_value_type>::lagrange(_points.at(b_ind)[poly_ind], _z.get(b_ind, poly_ind));
    }
    /// update_transcript (kzg_v2.hpp:150-190, kzg.hpp:695-738): commitments, evaluations, U polynomials
    void update_transcript(std::size_t batch_ind, transcript_type &transcript) {
        absorb_batch_commitments(batch_ind, transcript);
        for (std::size_t i = 0; i < _z.get_batch_size(batch_ind); ++i)
            for (std::size_t j = 0; j < _z.get_poly_points_number(batch_ind, i); ++j) absorb_scalar(_z.get(batch_ind, i, j), transcript);
        for (std::size_t i = 0; i < _points.at(batch_ind).size(); ++i)
            for (const auto &c : get_U(batch_ind, i)) absorb_scalar(c, transcript);
    }
    /// WHAT the transcript absorbs.  These device schemes hand it VALUES (one call per commitment / scalar); the reference hands it
    /// BYTES: the batch's commitment blob in one call, every scalar and pi_1 / pi_2 through nil::marshalling::pack (kzg_v2.hpp:153-189,
    /// 265-272, 296-304).  The placeholder-facing wrappers below override the three hooks to reproduce exactly that.
    virtual void absorb_batch_commitments(std::size_t batch_ind, transcript_type &transcript) {
        for (const auto &c : _ind_commitments.at(batch_ind)) transcript(c);
    }
    virtual void absorb_scalar(const scalar_value_type &v, transcript_type &transcript) { transcript(v); }
    virtual void absorb_point(const single_commitment_type &p, transcript_type &transcript) { transcript(p); }

public:
    virtual ~kzg_polys_evaluator_hip() = default;

protected:

    const params_type &_params;
    root_of_unity_type _root_of_unity;
    std::map<std::size_t, std::vector<const poly_type *>> _polys;    // in append order: copies held in _owned, or the caller's (lent)
    std::map<std::size_t, std::deque<poly_type>> _owned;             // a deque: references stay valid as it grows
    std::map<std::size_t, std::map<std::size_t, device_polynomial_dfs<CurveType>>> _resident;    // batch -> position -> resident polynomial
    std::map<std::size_t, std::map<std::size_t, device_polynomial_coefficients<CurveType>>> _resident_coefficients;    // ... in coefficient form
    std::map<std::size_t, bool> _locked;
    std::map<std::size_t, std::vector<std::vector<scalar_value_type>>> _points;
    std::map<std::size_t, device_batch> _dev;
    std::map<std::size_t, commitment_type> _ind_commitments;
    std::vector<scalar_value_type> _merged_points;
    mutable std::unique_ptr<context> _upload_ctx;
    const group_params_type *_group_params = nullptr;    // set: commit(batch) deals its columns over this group's members
    mutable std::size_t _group_multiexps = 0;
};

/// kzg_commitment_scheme_v2 (kzg_v2.hpp:56-360): two quotient commitments (pi_1, pi_2)
template <typename CurveType, typename TranscriptType, typename PolynomialType = polynomial_dfs<CurveType>>
class kzg_commitment_scheme_v2_hip : public kzg_polys_evaluator_hip<CurveType, TranscriptType, PolynomialType> {
    typedef kzg_polys_evaluator_hip<CurveType, TranscriptType, PolynomialType> base;

public:
    typedef typename base::adapter adapter;
    typedef typename base::curve_type curve_type;
    typedef typename base::scalar_value_type scalar_value_type;
    typedef typename base::single_commitment_type single_commitment_type;
    typedef typename base::commitment_type commitment_type;
    typedef typename base::transcript_type transcript_type;
    typedef typename base::params_type params_type;
    typedef typename base::poly_type poly_type;
    typedef typename base::eval_storage_type eval_storage_type;
    typedef typename base::preprocessed_data_type preprocessed_data_type;
    typedef typename base::root_of_unity_type root_of_unity_type;
    using base::_z;
    struct proof_type {
        eval_storage_type z;
        single_commitment_type pi_1, pi_2;
    };

    kzg_commitment_scheme_v2_hip(const params_type &kzg_params, root_of_unity_type root_of_unity) : base(kzg_params, std::move(root_of_unity)) { }
    explicit kzg_commitment_scheme_v2_hip(const params_type &kzg_params) : base(kzg_params) { }    // kzg_v2.hpp:94
    /// over a device group (kzg_params_group_hip): commit(batch) on all of its GPUs, the opening proof on member 0
    kzg_commitment_scheme_v2_hip(const kzg_params_group_hip<CurveType> &group_params, root_of_unity_type root_of_unity) : base(group_params, std::move(root_of_unity)) { }
    explicit kzg_commitment_scheme_v2_hip(const kzg_params_group_hip<CurveType> &group_params) : base(group_params) { }

    /// proof_eval (kzg_v2.hpp:236-305)
    proof_type proof_eval(transcript_type &transcript) {
        typedef detail::small_poly<scalar_value_type> SP;
        const context &ctx = _params.ctx;
        eval_polys();
        merge_eval_points();
        for (const auto &it : _ind_commitments) update_transcript(it.first, transcript);

        auto theta = transcript.challenge();
        const std::vector<scalar_value_type> V = SP::vanishing(_merged_points);

        /* every committed polynomial in the reference's iteration order (batches ascending, then index) */
        struct item {
            std::size_t k, i;
            const void *d;
            std::size_t len;
            std::vector<scalar_value_type> U, diff;
        };
        std::vector<item> items;
        std::size_t max_len = 0, taps = 1;
        for (const auto &it : _dev) {
            const std::size_t k = it.first;
            for (std::size_t i = 0; i < it.second.len.size(); ++i) {
                item e {k, i, it.second.at(i), it.second.len[i], get_U(k, i), set_difference_polynom(_points.at(k)[i])};
                max_len = std::max(max_len, e.len);
                taps = std::max(taps, e.diff.size());
                items.push_back(std::move(e));
            }
        }
        single_commitment_type pi_1 = single_commitment_type::zero(), pi_2 = single_commitment_type::zero();
        if (!items.empty()) {
            /* f = sum_i theta^i (f_i - U_i) diffpoly_i (kzg_v2.hpp:251-263): the f_i part in one pass over the resident
               coefficients, the U_i part (a few coefficients) on the host */
            const std::size_t acc_len = max_len + taps - 1;
            std::vector<const void *> ptrs;
            std::vector<std::size_t> lens;
            std::vector<std::uint64_t> coeffs(items.size() * taps * 4, 0);
            std::vector<scalar_value_type> corr;
            scalar_value_type theta_i = scalar_value_type::one();
            for (std::size_t n = 0; n < items.size(); ++n) {
                ptrs.push_back(items[n].d);
                lens.push_back(items[n].len);
                auto c = SP::scale(items[n].diff, theta_i);
                for (std::size_t t = 0; t < c.size(); ++t) adapter::scalar_to_limbs(c[t], &coeffs[4 * (n * taps + t)]);
                corr = SP::add(corr, SP::scale(SP::mul(items[n].U, c), scalar_value_type::zero() - scalar_value_type::one()));
                theta_i = theta_i * theta;
            }
            auto d_f = ctx.alloc(acc_len * 32);
            check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, items.size(), ptrs.data(), lens.data(), coeffs.data(), taps, d_f.get(), acc_len, 0),
                  "zkhip_poly_lincomb_dev", ctx.get());
            add_low_coefficients(d_f.get(), corr, acc_len);
            /* f /= V, one root at a time (kzg_v2.hpp:266-267); the remainders are the BOOST_ASSERT */
            char *f_ptr = static_cast<char *>(d_f.get());
            std::size_t f_len = acc_len;
            for (const auto &root : _merged_points) divide_in_place(f_ptr, f_len, root, "proof_eval: f is not divisible by V");
            pi_1 = commit_range(f_ptr, f_len);
            this->absorb_point(pi_1, transcript);

            auto theta_2 = transcript.challenge();
            /* L = sum_i theta^i Z_{T\S_i}(theta_2) (f_i - U_i(theta_2)) - V(theta_2) f (kzg_v2.hpp:281-289) */
            ptrs.push_back(f_ptr);
            lens.push_back(f_len);
            std::vector<std::uint64_t> c1((items.size() + 1) * 4, 0);
            scalar_value_type l0 = scalar_value_type::zero();
            theta_i = scalar_value_type::one();
            for (std::size_t n = 0; n < items.size(); ++n) {
                auto s = theta_i * SP::evaluate(items[n].diff, theta_2);
                adapter::scalar_to_limbs(s, &c1[4 * n]);
                l0 = l0 - s * SP::evaluate(items[n].U, theta_2);
                theta_i = theta_i * theta;
            }
            adapter::scalar_to_limbs(scalar_value_type::zero() - SP::evaluate(V, theta_2), &c1[4 * items.size()]);
            const std::size_t l_len = std::max(max_len, f_len);
            auto d_l = ctx.alloc(std::max<std::size_t>(1, l_len) * 32);
            check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, ptrs.size(), ptrs.data(), lens.data(), c1.data(), 1, d_l.get(), l_len, 0),
                  "zkhip_poly_lincomb_dev", ctx.get());
            add_low_coefficients(d_l.get(), {l0}, l_len);
            /* L /= (X - theta_2) (kzg_v2.hpp:290-291) */
            char *l_ptr = static_cast<char *>(d_l.get());
            std::size_t ll = l_len;
            divide_in_place(l_ptr, ll, theta_2, "proof_eval: L(theta_2) != 0");
            pi_2 = commit_range(l_ptr, ll);
            ctx.sync();    // d_f / d_l are released on return
        } else {
            this->absorb_point(pi_1, transcript);
            (void)transcript.challenge();
        }
        /* TODO in the reference: "Review the necessity of sending pi_2 to transcript" (kzg_v2.hpp:295) -- kept */
        this->absorb_point(pi_2, transcript);
        return proof_type {_z, pi_1, pi_2};
    }

protected:
    using base::_params;
    using base::_points;
    using base::_dev;
    using base::_ind_commitments;
    using base::_merged_points;
    using base::eval_polys;
    using base::merge_eval_points;
    using base::update_transcript;
    using base::get_U;
    using base::set_difference_polynom;
    using base::add_low_coefficients;
    using base::divide_in_place;
    using base::commit_range;
};

/// kzg_commitment_scheme (kzg.hpp:636-873), the first batched scheme ("Placeholder-friendly class"): ONE quotient commitment,
///   kzg_proof = commit( sum_j gamma^j (f_j - U_j) / V(S_j) )                                  (:782-807)
/// with the same polys_evaluator state and commit(batch) as v2.  The quotients are exact: polynomials that share an
/// evaluation-point set S are combined first (one pass over their resident coefficients), the few coefficients of their U_j
/// are subtracted, and the group is divided by V(S) root by root -- |S| synthetic divisions per GROUP instead of per polynomial.
/// verify_eval (:809-868) is pairings: the caller's, like v2's; commit_g2 (:659-664, 497-510) is provided for it.
template <typename CurveType, typename TranscriptType, typename PolynomialType = polynomial_dfs<CurveType>>
class kzg_commitment_scheme_hip : public kzg_polys_evaluator_hip<CurveType, TranscriptType, PolynomialType> {
    typedef kzg_polys_evaluator_hip<CurveType, TranscriptType, PolynomialType> base;

public:
    typedef typename base::adapter adapter;
    typedef typename base::curve_type curve_type;
    typedef typename base::scalar_value_type scalar_value_type;
    typedef typename base::single_commitment_type single_commitment_type;
    typedef typename base::commitment_type commitment_type;
    typedef typename base::transcript_type transcript_type;
    typedef typename base::params_type params_type;
    typedef typename base::poly_type poly_type;
    typedef typename base::eval_storage_type eval_storage_type;
    typedef typename base::preprocessed_data_type preprocessed_data_type;
    typedef typename base::root_of_unity_type root_of_unity_type;
    using base::_z;
    typedef typename adapter::g2_value_type verification_key_type;
    struct proof_type {
        eval_storage_type z;
        single_commitment_type kzg_proof;
    };

    kzg_commitment_scheme_hip(const params_type &kzg_params, root_of_unity_type root_of_unity) : base(kzg_params, std::move(root_of_unity)) { }
    explicit kzg_commitment_scheme_hip(const params_type &kzg_params) : base(kzg_params) { }    // kzg.hpp:667
    /// over a device group (kzg_params_group_hip): commit(batch) on all of its GPUs, the opening proof on member 0
    kzg_commitment_scheme_hip(const kzg_params_group_hip<CurveType> &group_params, root_of_unity_type root_of_unity) : base(group_params, std::move(root_of_unity)) { }
    explicit kzg_commitment_scheme_hip(const kzg_params_group_hip<CurveType> &group_params) : base(group_params) { }

    /// proof_eval (kzg.hpp:782-807)
    proof_type proof_eval(transcript_type &transcript) {
        typedef detail::small_poly<scalar_value_type> SP;
        const context &ctx = _params.ctx;
        eval_polys();
        merge_eval_points();
        for (const auto &it : _ind_commitments) update_transcript(it.first, transcript);
        const scalar_value_type gamma = transcript.challenge();

        /* the committed polynomials in the reference's iteration order (batches ascending, then index), grouped by point set */
        struct group {
            std::vector<scalar_value_type> pts;    // sorted
            std::vector<const void *> ptrs;
            std::vector<std::size_t> lens;
            std::vector<std::uint64_t> coeffs;     // gamma^j, canonical limbs
            std::vector<scalar_value_type> u_sum;  // sum_j gamma^j U_j
            std::size_t max_len = 0;
        };
        std::vector<group> groups;
        scalar_value_type factor = scalar_value_type::one();
        for (const auto &it : _dev) {
            const std::size_t k = it.first;
            for (std::size_t i = 0; i < it.second.len.size(); ++i) {
                std::vector<scalar_value_type> pts = _points.at(k)[i];
                std::sort(pts.begin(), pts.end(), detail::limbs_less<scalar_value_type>);
                auto g = std::find_if(groups.begin(), groups.end(), [&](const group &x) { return x.pts == pts; });
                if (g == groups.end()) {
                    groups.emplace_back();
                    g = groups.end() - 1;
                    g->pts = pts;
                }
                g->ptrs.push_back(it.second.at(i));
                g->lens.push_back(it.second.len[i]);
                g->coeffs.resize(g->coeffs.size() + 4);
                adapter::scalar_to_limbs(factor, g->coeffs.data() + g->coeffs.size() - 4);
                g->u_sum = SP::add(g->u_sum, SP::scale(get_U(k, i), factor));
                g->max_len = std::max(g->max_len, it.second.len[i]);
                factor = factor * gamma;
            }
        }
        single_commitment_type kzg_proof = single_commitment_type::zero();
        std::size_t acc_len = 0;
        for (const auto &g : groups) acc_len = std::max(acc_len, g.max_len > g.pts.size() ? g.max_len - g.pts.size() : 0);
        if (acc_len != 0) {
            auto d_acc = ctx.alloc(acc_len * 32);
            bool first = true;
            const std::uint64_t one[4] = {1, 0, 0, 0};
            for (const auto &g : groups) {
                if (g.max_len <= g.pts.size()) continue;    // deg f < |S|: f = U, the quotient is zero
                auto d_g = ctx.alloc(g.max_len * 32);
                check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, g.ptrs.size(), g.ptrs.data(), g.lens.data(), g.coeffs.data(), 1, d_g.get(), g.max_len, 0),
                      "zkhip_poly_lincomb_dev", ctx.get());
                add_low_coefficients(d_g.get(), SP::scale(g.u_sum, scalar_value_type::zero() - scalar_value_type::one()), g.max_len);
                char *q_ptr = static_cast<char *>(d_g.get());
                std::size_t q_len = g.max_len;
                for (const auto &root : g.pts) divide_in_place(q_ptr, q_len, root, "proof_eval: (f - U) is not divisible by V");
                const void *qp = q_ptr;
                check(zkhip_poly_lincomb_dev(ctx.get(), adapter::id, 1, &qp, &q_len, one, 1, d_acc.get(), acc_len, first ? 0 : 1), "zkhip_poly_lincomb_dev",
                      ctx.get());
                first = false;
                ctx.sync();    // d_g is released at the end of the iteration
            }
            if (!first) kzg_proof = commit_range(d_acc.get(), acc_len);
            ctx.sync();
        }
        return proof_type {_z, kzg_proof};
    }

    /// commit_g2 (kzg.hpp:659-664): the verifier's G2 commitments of the few-coefficient polynomials Z_{T \ S}, V(T)
    verification_key_type commit_g2(const std::vector<scalar_value_type> &poly) const { return nil::crypto3::zk::hip::commit_g2<CurveType>(_params, poly); }

protected:
    using base::_params;
    using base::_points;
    using base::_dev;
    using base::_ind_commitments;
    using base::_merged_points;
    using base::eval_polys;
    using base::merge_eval_points;
    using base::update_transcript;
    using base::get_U;
    using base::set_difference_polynom;
    using base::add_low_coefficients;
    using base::divide_in_place;
    using base::commit_range;
};


/// nil::marshalling::pack<option::big_endian>(field element) as update_transcript applies it to evaluations and U coefficients
/// (kzg_v2.hpp:163-165, 179-181): the canonical value as 32 big-endian bytes.  crypto3-marshalling is not vendored; the field
/// encoding is the one the proving-key wire format states (g16/marshalling.hpp) with the transcript's big-endian option.
template <typename CurveType>
struct scalar_be_packer {
    std::vector<std::uint8_t> operator()(const typename curve_adapter<CurveType>::scalar_value_type &v) const {
        std::uint64_t l[4];
        curve_adapter<CurveType>::scalar_to_limbs(v, l);
        std::vector<std::uint8_t> b(32);
        for (int i = 0; i < 32; ++i) b[31 - i] = (std::uint8_t)(l[i >> 3] >> (8 * (i & 7)));
        return b;
    }
};

/// The scheme object placeholder_prover / placeholder_verifier take as ParamsType::commitment_scheme_type
/// (ph/params.hpp:50-63; the consumer contract is what dummy_commitment_scheme_type implements,
/// test/systems/plonk/placeholder/placeholder.cpp:96-148): the device scheme above with the reference's commitment_type
/// (a byte blob) and a verify_eval.
///   Packer:   std::vector<std::uint8_t>(const single_commitment_type &)   -- nil::marshalling::pack<endianness>(point, status)
///   Verifier: bool(scheme &, const proof_type &, const std::map<std::size_t, commitment_type> &, transcript_type &)
///   ScalarPacker: std::vector<std::uint8_t>(const scalar_value_type &)     -- nil::marshalling::pack<endianness>(field element, status)
/// With them the TRANSCRIPT sees what the reference's sees, byte for byte: the batch's commitment blob in one call, every evaluation
/// and U coefficient packed, pi_1 and pi_2 packed (kzg_v2.hpp:150-190, 265-272, 296-304; tests/cpp/shim_test.cpp records the calls).
template <typename CurveType, typename TranscriptType, typename Packer, typename Verifier, typename PolynomialType = polynomial_dfs<CurveType>,
          typename ScalarPacker = scalar_be_packer<CurveType>>
class kzg_commitment_scheme_v2_placeholder_hip : public kzg_commitment_scheme_v2_hip<CurveType, TranscriptType, PolynomialType> {
    typedef kzg_commitment_scheme_v2_hip<CurveType, TranscriptType, PolynomialType> base;

public:
    typedef std::vector<std::uint8_t> commitment_type;
    typedef typename base::proof_type proof_type;
    typedef typename base::transcript_type transcript_type;
    typedef typename base::params_type params_type;
    typedef typename base::root_of_unity_type root_of_unity_type;

    kzg_commitment_scheme_v2_placeholder_hip(const params_type &kzg_params, root_of_unity_type root_of_unity, Packer packer, Verifier verifier,
                                             ScalarPacker scalar_packer = ScalarPacker()) :
        base(kzg_params, std::move(root_of_unity)), _packer(std::move(packer)), _verifier(std::move(verifier)), _scalar_packer(std::move(scalar_packer)) { }

    /// kzg_v2.hpp:208-226: "Differs from static, because we pack the result into byte blob."
    commitment_type commit(std::size_t index) {
        commitment_type result;
        for (const auto &single_commitment : base::commit(index)) {
            const std::vector<std::uint8_t> bytes = _packer(single_commitment);
            result.insert(result.end(), bytes.begin(), bytes.end());
        }
        _commitments[index] = result;
        return result;
    }
    /// kzg_v2.hpp:312: the caller's pairing check, with this object's evaluation points / batch layout at hand
    bool verify_eval(const proof_type &proof, const std::map<std::size_t, commitment_type> &commitments, transcript_type &transcript) {
        return _verifier(*this, proof, commitments, transcript);
    }
    const std::map<std::size_t, commitment_type> &packed_commitments() const { return _commitments; }
    /// the evaluation points of (batch, polynomial), for the verifier's U / Z_{T \ S} polynomials
    const std::vector<typename base::scalar_value_type> &eval_points(std::size_t batch, std::size_t poly) const { return this->_points.at(batch).at(poly); }

protected:
    /* the reference's transcript traffic (see the class comment) */
    void absorb_batch_commitments(std::size_t batch_ind, transcript_type &transcript) override { transcript(_commitments.at(batch_ind)); }
    void absorb_scalar(const typename base::scalar_value_type &v, transcript_type &transcript) override { transcript(_scalar_packer(v)); }
    void absorb_point(const typename base::single_commitment_type &p, transcript_type &transcript) override { transcript(_packer(p)); }

private:
    Packer _packer;
    Verifier _verifier;
    ScalarPacker _scalar_packer;
    std::map<std::size_t, commitment_type> _commitments;
};

/// The same for the FIRST batched scheme (kzg_commitment_scheme, kzg.hpp:636-873): `commit` packs the single commitments into the
/// reference's byte blob (:748-765), `verify_eval` (:809-868) goes to the caller's pairing check, which finds commit_g2, the
/// evaluation points and the per-polynomial commitments on this object.
template <typename CurveType, typename TranscriptType, typename Packer, typename Verifier, typename PolynomialType = polynomial_dfs<CurveType>,
          typename ScalarPacker = scalar_be_packer<CurveType>>
class kzg_commitment_scheme_placeholder_hip : public kzg_commitment_scheme_hip<CurveType, TranscriptType, PolynomialType> {
    typedef kzg_commitment_scheme_hip<CurveType, TranscriptType, PolynomialType> base;

public:
    typedef std::vector<std::uint8_t> commitment_type;
    typedef typename base::proof_type proof_type;
    typedef typename base::transcript_type transcript_type;
    typedef typename base::params_type params_type;
    typedef typename base::root_of_unity_type root_of_unity_type;

    kzg_commitment_scheme_placeholder_hip(const params_type &kzg_params, root_of_unity_type root_of_unity, Packer packer, Verifier verifier,
                                          ScalarPacker scalar_packer = ScalarPacker()) :
        base(kzg_params, std::move(root_of_unity)), _packer(std::move(packer)), _verifier(std::move(verifier)), _scalar_packer(std::move(scalar_packer)) { }

    commitment_type commit(std::size_t index) {
        commitment_type result;
        for (const auto &single_commitment : base::commit(index)) {
            const std::vector<std::uint8_t> bytes = _packer(single_commitment);
            result.insert(result.end(), bytes.begin(), bytes.end());
        }
        _commitments[index] = result;
        return result;
    }
    bool verify_eval(const proof_type &proof, const std::map<std::size_t, commitment_type> &commitments, transcript_type &transcript) {
        return _verifier(*this, proof, commitments, transcript);
    }
    const std::map<std::size_t, commitment_type> &packed_commitments() const { return _commitments; }
    const std::vector<typename base::scalar_value_type> &eval_points(std::size_t batch, std::size_t poly) const { return this->_points.at(batch).at(poly); }

protected:
    /* the reference's transcript traffic (kzg.hpp:695-738) */
    void absorb_batch_commitments(std::size_t batch_ind, transcript_type &transcript) override { transcript(_commitments.at(batch_ind)); }
    void absorb_scalar(const typename base::scalar_value_type &v, transcript_type &transcript) override { transcript(_scalar_packer(v)); }
    void absorb_point(const typename base::single_commitment_type &p, transcript_type &transcript) override { transcript(_packer(p)); }

private:
    Packer _packer;
    Verifier _verifier;
    ScalarPacker _scalar_packer;
    std::map<std::size_t, commitment_type> _commitments;
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_KZG_V2_HPP
