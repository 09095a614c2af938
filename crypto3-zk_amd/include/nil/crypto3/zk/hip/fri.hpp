//---------------------------------------------------------------------------//
// zkhip shim: the NTT side of the LPC / FRI commitment layer and polynomial_dfs arithmetic, on the MI355X.
//
// Mirrors (same names, same argument meaning):
//   math::polynomial_dfs::{resize, coefficients, from_coefficients, operator+=, -=, *=}, math::polynomial_product,
//   math::polynomial_shift   (crypto3-math; used at
//        zk/snark/systems/plonk/placeholder/prover.hpp:255,277, gates_argument.hpp:119-121,
//        permutation_argument.hpp:148-167, zk/commitments/detail/polynomial/basic_fri.hpp:452-455)
//   detail::fold_polynomial, DFS form        zk/commitments/detail/polynomial/fold_polynomial.hpp:68-93
//   algorithms::precommit<FRI>               zk/commitments/detail/polynomial/basic_fri.hpp:433-496, up to the
//        Merkle tree: domain extension of every polynomial + the coset-ordered leaf data; hashing is the caller's.
// The roots of unity come from the caller (math::make_evaluation_domain's choice).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_FRI_HPP
#define ZKHIP_SHIM_FRI_HPP

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <vector>

#include "scoped_profiler.hpp"
#include "kzg.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// math::polynomial_dfs with its evaluations resident on the device
template <typename CurveType>
class device_polynomial_dfs {
public:
    typedef curve_adapter<CurveType> adapter;
    typedef typename adapter::scalar_value_type value_type;
    typedef std::function<value_type(std::size_t log_n)> root_of_unity_type;

    device_polynomial_dfs(const context &ctx, std::size_t size) :
        ctx_(&ctx), size_(size), degree_(size ? size - 1 : 0), d_(ctx.alloc(std::max<std::size_t>(1, size) * 32)) { }
    /// `degree`: polynomial_dfs carries its degree next to the evaluations (polynomial_product sizes its result from
    /// it); without one the vector is taken to be full (size - 1)
    template <typename PolynomialType, typename = decltype(std::declval<const PolynomialType &>()[0])>
    device_polynomial_dfs(const context &ctx, const PolynomialType &p, std::size_t degree = (std::size_t)-1) : device_polynomial_dfs(ctx, p.size()) {
        upload_scalars<adapter>(ctx, d_.get(), detail::poly_data<adapter>(p), size_);
        if (degree != (std::size_t)-1) degree_ = degree;
    }
    std::size_t size() const { return size_; }
    std::size_t degree() const { return degree_; }
    void set_degree(std::size_t d) { degree_ = d; }
    void *data() const { return d_.get(); }
    const context &ctx() const { return *ctx_; }

    /// EXTENSION CACHE (round 5).  placeholder's arguments extend the same polynomials again and again: the preprocessed ones (q_last, q_blind,
    /// lagrange_0, the identity / permutation polynomials) in every proof, a witness column once per argument that reads it.  A polynomial whose
    /// holder calls enable_extension_cache() keeps the extensions extension() computes (size -> buffer; copies of the object share buffer AND
    /// cache) until clear_extension_cache(), an in-place change of the values, or the last copy's end.  The key holder enables it on the
    /// preprocessed polynomials for good (2^20 rows: 32 MiB per polynomial and factor of extension), the prover on the columns it hands to
    /// several arguments for the duration of a proof.  The reference extends anew at every use (polynomial_dfs::resize inside operator*).
    void enable_extension_cache() {
        if (!cache_) cache_ = std::make_shared<std::map<std::size_t, std::shared_ptr<void>>>();
    }
    bool extension_cache_enabled() const { return (bool)cache_; }
    void clear_extension_cache() {
        if (cache_) cache_->clear();
    }
    /// this polynomial on the `size`-point domain (size >= size(), a power of two): itself, a cached extension, or a fresh one (cached when the
    /// cache is on).  A result that SHARES its buffer with the cache is a read-only VIEW (is_view()): the in-place operations of this class
    /// (operator+= / -= / *=, from_coefficients) first give it a buffer of its own (copy on write: ADVICE r5 -- they used to corrupt the
    /// owner's cached extension for every later proof); code that writes through data() must call make_writable() first.
    device_polynomial_dfs extension(std::size_t size, const root_of_unity_type &root) const {
        if (size <= size_) return *this;
        if (cache_) {
            auto it = cache_->find(size);
            if (it != cache_->end()) {
                device_polynomial_dfs hit(*ctx_, size, degree_, it->second);
                hit.view_ = true;
                return hit;
            }
        }
        device_polynomial_dfs e = *this;
        e.cache_.reset();
        e.resize(size, root);
        if (cache_) {
            (*cache_)[size] = e.d_;
            e.view_ = true;
        }
        return e;
    }
    /// true for a polynomial whose buffer is a holder's cached extension (see extension())
    bool is_view() const { return view_; }
    /// a private buffer for a view (no-op otherwise), and the end of this object's own cached extensions: call before writing through data()
    void make_writable() {
        clear_extension_cache();
        if (!view_) return;
        auto d_new = ctx_->alloc(std::max<std::size_t>(1, size_) * 32);
        check(zkhip_memcpy_d2d_async(ctx_->get(), d_new.get(), d_.get(), size_ * 32), "zkhip_memcpy_d2d_async", ctx_->get());
        d_ = d_new;    // the cache keeps the old buffer alive; the copy is ordered before whatever this stream does next
        view_ = false;
    }

    polynomial_dfs<CurveType> to_host() const {
        polynomial_dfs<CurveType> out;
        download_scalars<adapter>(*ctx_, d_.get(), size_, out.values);
        return out;
    }

    /// resize(new_size): evaluations on the size()-point domain -> evaluations of the same polynomial on the
    /// new_size-point domain (both powers of two).  Growing goes through the coefficients (inverse NTT, zero padding,
    /// NTT); shrinking -- legal while degree() < new_size -- keeps every (size / new_size)-th evaluation.
    void resize(std::size_t new_size, const root_of_unity_type &root) {
        if (new_size == size_) return;
        const std::size_t lo = log2_exact(size_), ln = log2_exact(new_size);
        if (ln < lo && degree_ >= new_size) throw std::runtime_error("device_polynomial_dfs::resize: the polynomial's degree does not fit the smaller domain");
        auto d_new = ctx_->alloc(new_size * 32);
        std::uint64_t wo[4], wn[4];
        adapter::scalar_to_limbs(root(lo), wo);
        adapter::scalar_to_limbs(root(ln), wn);
        /* growing consumes its input (zkhip_poly_resize_dev leaves the coefficients there) and copies of this object
           share the buffer: work on a private copy */
        std::shared_ptr<void> src = d_;
        if (ln > lo) {
            src = ctx_->alloc(size_ * 32);
            check(zkhip_memcpy_d2d_async(ctx_->get(), src.get(), d_.get(), size_ * 32), "zkhip_memcpy_d2d_async", ctx_->get());
        }
        check(zkhip_poly_resize_dev(ctx_->get(), adapter::id, src.get(), lo, 1, wo, d_new.get(), ln, wn), "zkhip_poly_resize_dev", ctx_->get());
        ctx_->sync();    // the temporaries are released below
        d_ = d_new;
        size_ = new_size;
        cache_.reset();    // another polynomial object now: copies made before keep the old buffer and its cache
        view_ = false;     // ... with a buffer of its own
    }
    /// coefficients(): inverse NTT into a new device buffer of size() elements
    std::shared_ptr<void> coefficients(const root_of_unity_type &root) const {
        auto d_c = ctx_->alloc(std::max<std::size_t>(1, size_) * 32);
        check(zkhip_memcpy_d2d_async(ctx_->get(), d_c.get(), d_.get(), size_ * 32), "zkhip_memcpy_d2d_async", ctx_->get());
        std::uint64_t w[4];
        adapter::scalar_to_limbs(root(log2_exact(size_)), w);
        check(zkhip_ntt_dev(ctx_->get(), adapter::id, d_c.get(), log2_exact(size_), 1, w, 1, nullptr), "zkhip_ntt_dev", ctx_->get());
        return d_c;
    }
    /// from_coefficients(): the evaluations of the polynomial whose size() coefficients are at d_coeffs
    void from_coefficients(const void *d_coeffs, const root_of_unity_type &root) {
        make_writable();
        check(zkhip_memcpy_d2d_async(ctx_->get(), d_.get(), d_coeffs, size_ * 32), "zkhip_memcpy_d2d_async", ctx_->get());
        std::uint64_t w[4];
        adapter::scalar_to_limbs(root(log2_exact(size_)), w);
        check(zkhip_ntt_dev(ctx_->get(), adapter::id, d_.get(), log2_exact(size_), 1, w, 0, nullptr), "zkhip_ntt_dev", ctx_->get());
        ctx_->sync();
    }
    device_polynomial_dfs &operator+=(const device_polynomial_dfs &o) {
        degree_ = std::max(degree_, o.degree_);
        return pointwise(0, o);
    }
    device_polynomial_dfs &operator-=(const device_polynomial_dfs &o) {
        degree_ = std::max(degree_, o.degree_);
        return pointwise(1, o);
    }
    /// pointwise product on the shared domain (the caller resized first, as the reference's operator*= does internally)
    device_polynomial_dfs &operator*=(const device_polynomial_dfs &o) {
        degree_ = std::min(size_ ? size_ - 1 : 0, degree_ + o.degree_);
        return pointwise(2, o);
    }

private:
    static std::size_t log2_exact(std::size_t n) {
        std::size_t l = 0;
        while (((std::size_t)1 << l) < n) ++l;
        if (n == 0 || ((std::size_t)1 << l) != n) throw std::runtime_error("device_polynomial_dfs: size must be a power of two");
        return l;
    }
    device_polynomial_dfs &pointwise(int op, const device_polynomial_dfs &o) {
        make_writable();    // the values change in place: cached extensions go, a view gets its own buffer first
        if (o.size_ != size_) throw std::runtime_error("device_polynomial_dfs: operands must share the domain (resize first)");
        check(zkhip_fr_vec_op_dev(ctx_->get(), adapter::id, op, d_.get(), o.d_.get(), d_.get(), size_), "zkhip_fr_vec_op_dev", ctx_->get());
        return *this;
    }

    /// a view of an existing buffer (a cached extension)
    device_polynomial_dfs(const context &ctx, std::size_t size, std::size_t degree, std::shared_ptr<void> d) : ctx_(&ctx), size_(size), degree_(degree), d_(std::move(d)) { }

    const context *ctx_;
    std::size_t size_, degree_;
    std::shared_ptr<void> d_;
    std::shared_ptr<std::map<std::size_t, std::shared_ptr<void>>> cache_;    // extension cache, shared by the copies of this object (null: off)
    bool view_ = false;    // d_ is a cached extension of another polynomial: read-only (make_writable)
};

/// A polynomial resident in COEFFICIENT form: `size` (a power of two) coefficients, zero-padded.  What a KZG scheme keeps of every committed
/// polynomial anyway; placeholder's quotient parts are born in this form (chunks of T's coefficients, prover.hpp:244-249) and the reference
/// turns them into polynomial_dfs (:255) only for commit() to turn them back (kzg.hpp:431) -- 2 n transform points per part that
/// append_to_batch(batch, device_polynomial_coefficients) saves.
template <typename CurveType>
class device_polynomial_coefficients {
public:
    device_polynomial_coefficients(const context &ctx, std::size_t size) : ctx_(&ctx), size_(size), d_(ctx.alloc(std::max<std::size_t>(1, size) * 32)) { }
    std::size_t size() const { return size_; }
    void *data() const { return d_.get(); }
    const context &ctx() const { return *ctx_; }
    /// the producer KNOWS every coefficient is zero (a quotient part beyond the quotient's length: the reference commits those too,
    /// prover.hpp:251-257, and gets the point at infinity): a commitment scheme may skip the multiexp and put down the neutral element
    bool known_zero() const { return zero_; }
    void set_known_zero(bool z) { zero_ = z; }

private:
    const context *ctx_;
    std::size_t size_;
    std::shared_ptr<void> d_;
    bool zero_ = false;
};

/// math::polynomial_product(multipliers) (ph/permutation_argument.hpp:148, gates_argument.hpp:117): the product of k
/// DFS polynomials on the smallest power-of-two domain that holds its degree (sum of the degrees): every factor is
/// resized to it, then ONE k-way pointwise pass multiplies them.  A factor may arrive on a LARGER domain than the product's (a polynomial
/// a caller extended once for several products): it is subsampled, which costs no transform.  `min_size`: a floor for the product's
/// domain, for a result that is combined with polynomials living there afterwards.
template <typename CurveType>
device_polynomial_dfs<CurveType> polynomial_product(std::vector<device_polynomial_dfs<CurveType>> multipliers,
                                                    const typename device_polynomial_dfs<CurveType>::root_of_unity_type &root, std::size_t min_size = 0) {
    typedef curve_adapter<CurveType> adapter;
    if (multipliers.empty()) throw std::invalid_argument("polynomial_product: no factors");
    const context &ctx = multipliers[0].ctx();
    std::size_t degree = 0, size = 1;
    for (const auto &m : multipliers) degree += m.degree();
    while (size < degree + 1 || size < min_size) size <<= 1;
    std::vector<const void *> ptrs;
    for (auto &m : multipliers) {
        m.resize(size, root);
        ptrs.push_back(m.data());
    }
    device_polynomial_dfs<CurveType> out(ctx, size);
    out.set_degree(degree);
    check(zkhip_fr_vec_prod_dev(ctx.get(), adapter::id, ptrs.size(), ptrs.data(), out.data(), size), "zkhip_fr_vec_prod_dev", ctx.get());
    ctx.sync();    // the resized copies are released on return
    return out;
}

/// math::polynomial_shift(f, shift, domain_size) (ph/permutation_argument.hpp:148, lookup_argument.hpp:232,315,360):
/// the evaluations of f(omega^shift X), omega the generator of the domain_size-point domain f is an extension over
/// (domain_size = 0: f's own domain): entry i reads entry i + shift * (size / domain_size), cyclically.
template <typename CurveType>
device_polynomial_dfs<CurveType> polynomial_shift(const device_polynomial_dfs<CurveType> &f, int shift, std::size_t domain_size = 0) {
    if (domain_size == 0) domain_size = f.size();
    if (domain_size == 0 || f.size() % domain_size) throw std::invalid_argument("polynomial_shift: the vector is not an extension of the domain");
    std::size_t log_size = 0;
    while (((std::size_t)1 << log_size) < f.size()) ++log_size;
    device_polynomial_dfs<CurveType> out(f.ctx(), f.size());
    out.set_degree(f.degree());
    check(zkhip_poly_shift_dev(f.ctx().get(), f.data(), log_size, (std::int64_t)shift * (std::int64_t)(f.size() / domain_size), out.data()),
          "zkhip_poly_shift_dev", f.ctx().get());
    return out;
}

/// detail::fold_polynomial, DFS form (fold_polynomial.hpp:68-93): f over the size()-point domain with generator
/// `omega` -> the folded polynomial over the half-size domain.
template <typename CurveType>
device_polynomial_dfs<CurveType> fold_polynomial(const device_polynomial_dfs<CurveType> &f, const typename curve_adapter<CurveType>::scalar_value_type &alpha,
                                                 const typename curve_adapter<CurveType>::scalar_value_type &omega) {
    typedef curve_adapter<CurveType> adapter;
    std::size_t log_size = 0;
    while (((std::size_t)1 << log_size) < f.size()) ++log_size;
    device_polynomial_dfs<CurveType> out(f.ctx(), f.size() / 2);
    std::uint64_t a[4], w[4];
    adapter::scalar_to_limbs(alpha, a);
    adapter::scalar_to_limbs(omega, w);
    check(zkhip_fri_fold_dev(f.ctx().get(), adapter::id, f.data(), log_size, a, w, out.data()), "zkhip_fri_fold_dev", f.ctx().get());
    return out;
}

/// The device part of algorithms::precommit<FRI>(poly, D, fri_step) (basic_fri.hpp:433-496): every polynomial is
/// extended to the 2^log_domain-point domain D (`poly[i].resize(D->size())`, :452-455) and the leaves are laid out
/// in the reference's coset order; returns the leaf data (2^log_domain / 2^fri_step leaves of
/// polys.size() * 2^fri_step canonical elements each) on the host for the caller's Merkle tree.
template <typename CurveType, typename PolynomialType = polynomial_dfs<CurveType>>
std::vector<typename curve_adapter<CurveType>::scalar_value_type>
    precommit_leaves(const context &ctx, const std::vector<PolynomialType> &polys, std::size_t log_domain, std::size_t fri_step,
                     const typename device_polynomial_dfs<CurveType>::root_of_unity_type &root) {
    typedef curve_adapter<CurveType> adapter;
    ZKHIP_PROFILE_SCOPE("Basic FRI Precommit time");    // basic_fri.hpp:449
    const std::size_t D = (std::size_t)1 << log_domain, batch = polys.size();
    std::vector<typename adapter::scalar_value_type> out;
    if (batch == 0) return out;
    auto d_ext = ctx.alloc(batch * D * 32);
    std::uint64_t wd[4];
    adapter::scalar_to_limbs(root(log_domain), wd);
    /* runs of equally sized polynomials are extended by one batched call */
    for (std::size_t i = 0; i < batch;) {
        std::size_t j = i;
        while (j < batch && polys[j].size() == polys[i].size()) ++j;
        const std::size_t n = polys[i].size();
        std::size_t log_n = 0;
        while (((std::size_t)1 << log_n) < n) ++log_n;
        if (n == 0 || ((std::size_t)1 << log_n) != n || log_n > log_domain) throw std::runtime_error("precommit: bad polynomial size");
        char *dst = static_cast<char *>(d_ext.get()) + 32 * i * D;
        if (log_n == log_domain) {
            for (std::size_t p = i; p < j; ++p) upload_scalars<adapter>(ctx, dst + 32 * (p - i) * n, detail::poly_data<adapter>(polys[p]), n);
        } else {
            auto d_in = ctx.alloc(n * (j - i) * 32);
            for (std::size_t p = i; p < j; ++p) upload_scalars<adapter>(ctx, static_cast<char *>(d_in.get()) + 32 * (p - i) * n, detail::poly_data<adapter>(polys[p]), n);
            std::uint64_t wn[4];
            adapter::scalar_to_limbs(root(log_n), wn);
            check(zkhip_poly_resize_dev(ctx.get(), adapter::id, d_in.get(), log_n, j - i, wn, dst, log_domain, wd), "zkhip_poly_resize_dev", ctx.get());
            ctx.sync();
        }
        i = j;
    }
    auto d_leaves = ctx.alloc(batch * D * 32);
    check(zkhip_fri_leaves_dev(ctx.get(), d_ext.get(), log_domain, batch, fri_step, d_leaves.get()), "zkhip_fri_leaves_dev", ctx.get());
    download_scalars<adapter>(ctx, d_leaves.get(), batch * D, out);
    return out;
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_FRI_HPP
