//---------------------------------------------------------------------------//
// zkhip shim: RAII view of the C ABI (include/zkhip.h) for the header-only C++ classes that mirror the
// reference's call sites.  Errors become std::runtime_error here, on the C++ side of the boundary (the ABI
// itself never throws); the reference's own failure mode at these call sites is BOOST_ASSERT / one
// std::runtime_error (kzg.hpp:167).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_BACKEND_HPP
#define ZKHIP_SHIM_BACKEND_HPP

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <future>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <sys/random.h>

#include "algebra.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

namespace detail {
    /// `bytes` bytes from the kernel's CSPRNG (getrandom(2), blocking until the pool is initialised); throws on failure
    inline void os_random_bytes(void *out, std::size_t bytes) {
        unsigned char *p = static_cast<unsigned char *>(out);
        while (bytes) {
            const ssize_t k = getrandom(p, bytes, 0);
            if (k < 0) {
                if (errno == EINTR) continue;
                throw std::runtime_error("getrandom failed: no entropy source for the prover's blinding factors");
            }
            p += k;
            bytes -= (std::size_t)k;
        }
    }
}    // namespace detail

inline void check(int rc, const char *what, const zkhip_ctx *ctx = nullptr) {
    if (rc != ZKHIP_OK) {
        std::string msg = std::string(what) + ": " + zkhip_strerror(rc);
        if (ctx) msg += std::string(" [") + zkhip_last_error(ctx) + "]";
        throw std::runtime_error(msg);
    }
}

/// One context per GPU per process.  There is no CPU fallback: construction throws without a GPU.
class context {
public:
    explicit context(int device = 0) {
        check(zkhip_init(device, &ctx_), "zkhip_init");
    }
    /// a view of a context somebody else owns and outlives this object with (a member of a device_group)
    struct borrowed_tag { };
    context(borrowed_tag, zkhip_ctx *borrowed) : ctx_(borrowed), owns_(false) {
        if (!borrowed) throw std::invalid_argument("context: null handle");
    }
    ~context() {
        if (owns_) zkhip_destroy(ctx_);
    }
    context(const context &) = delete;
    context &operator=(const context &) = delete;
    zkhip_ctx *get() const { return ctx_; }
    int device() const { return zkhip_device(ctx_); }
    /// work enqueued here from now on runs after what `other` has enqueued so far (both on the same GPU)
    void wait_for(const context &other) const { check(zkhip_stream_wait(ctx_, other.ctx_), "zkhip_stream_wait", ctx_); }
    void sync() const { check(zkhip_sync(ctx_), "zkhip_sync", ctx_); }
    /// the sticky flags kernels raised since the last call (zkhip.h: zkhip_device_status), read and cleared; synchronises; 0 = none
    std::uint32_t device_status() const {
        std::uint32_t flags = 0;
        const int rc = zkhip_device_status(ctx_, &flags);
        if (rc != ZKHIP_OK && rc != ZKHIP_ERR_RANGE) check(rc, "zkhip_device_status", ctx_);
        return flags;
    }
    void set_option(const char *name, std::int64_t value) const { check(zkhip_set_option(ctx_, name, value), "zkhip_set_option", ctx_); }
    std::int64_t get_option(const char *name) const {
        std::int64_t v = 0;
        check(zkhip_get_option(ctx_, name, &v), "zkhip_get_option", ctx_);
        return v;
    }

    /// device buffer of `bytes` bytes, freed with the returned handle
    std::shared_ptr<void> alloc(std::size_t bytes) const {
        void *p = nullptr;
        check(zkhip_malloc(ctx_, bytes, &p), "zkhip_malloc", ctx_);
        zkhip_ctx *c = ctx_;
        return std::shared_ptr<void>(p, [c](void *q) { zkhip_free(c, q); });
    }
    void h2d(void *dst, const void *src, std::size_t bytes) const { check(zkhip_memcpy_h2d(ctx_, dst, src, bytes), "zkhip_memcpy_h2d", ctx_); }
    void d2h(void *dst, const void *src, std::size_t bytes) const { check(zkhip_memcpy_d2h(ctx_, dst, src, bytes), "zkhip_memcpy_d2h", ctx_); }
    /// enqueue only; `dst` page-locked (pinned_buffer), valid after the next sync()
    void d2h_async(void *dst, const void *src, std::size_t bytes) const {
        check(zkhip_memcpy_d2h_async(ctx_, dst, src, bytes), "zkhip_memcpy_d2h_async", ctx_);
    }

private:
    zkhip_ctx *ctx_ = nullptr;
    bool owns_ = true;
};

/// N GPUs behind ONE caller (include/zkhip.h, "device group").  The reference hides its parallelism inside the call -- `chunks =
/// omp_get_max_threads()` inside r1cs_gg_ppzksnark_prover::process (prover.hpp:94-99), the loop over the batch inside
/// kzg_commitment_scheme_v2::commit (kzg_v2.hpp:208-226) -- and so do the shim classes that take a device_group: one context per device,
/// one host thread, the partial sums exchanged INSIDE the library (RCCL all-gather over xGMI, peer copies, or a staged copy).
///     device_group gpus({0, 1, 2, 3, 4, 5, 6, 7});      // or device_group::from_env(): ZKHIP_DEVICES=0,1,2,3,4,5,6,7
/// A device may be named more than once (several members on one GPU).
class device_group {
public:
    explicit device_group(const std::vector<int> &devices) {
        if (devices.empty()) throw std::invalid_argument("device_group: no devices");
        check(zkhip_group_init(devices.data(), (int)devices.size(), &g_), "zkhip_group_init");
        try {
            for (std::size_t k = 0; k < devices.size(); ++k)
                members_.emplace_back(new context(context::borrowed_tag(), zkhip_group_ctx(g_, (int)k)));
        } catch (...) {
            members_.clear();
            zkhip_group_destroy(g_);
            throw;
        }
    }
    ~device_group() {
        members_.clear();
        zkhip_group_destroy(g_);
    }
    device_group(const device_group &) = delete;
    device_group &operator=(const device_group &) = delete;

    /// the devices ZKHIP_DEVICES names ("0,1,2,3"; empty when the variable is unset or malformed)
    static std::vector<int> devices_from_env() {
        std::vector<int> out;
        const char *e = std::getenv("ZKHIP_DEVICES");
        if (!e) return out;
        const std::string all(e);
        std::size_t at = 0;
        while (at <= all.size()) {
            const std::size_t end = all.find(',', at);
            const std::string item = all.substr(at, end == std::string::npos ? std::string::npos : end - at);
            if (item.empty() || item.find_first_not_of("0123456789") != std::string::npos) return {};
            out.push_back(std::atoi(item.c_str()));
            if (end == std::string::npos) break;
            at = end + 1;
        }
        return out;
    }

    std::size_t size() const { return members_.size(); }
    const context &operator[](std::size_t k) const { return *members_.at(k); }
    /// member 0: where gathered partial sums are folded and what one-device objects of a group-aware caller live on
    const context &root() const { return *members_.front(); }
    zkhip_device_group *get() const { return g_; }
    void check_group(int rc, const char *what) const {
        if (rc != ZKHIP_OK) throw std::runtime_error(std::string(what) + ": " + zkhip_strerror(rc) + " [" + zkhip_group_last_error(g_) + "]");
    }
    /// ZKHIP_GROUP_AUTO / _RCCL / _PEER / _STAGED (zkhip.h)
    void set_transport(int kind) const { check_group(zkhip_group_set_transport(g_, kind), "zkhip_group_set_transport"); }
    int transport() const { return zkhip_group_transport(g_); }
    /// member k's `bytes` bytes at d_send[k] -> d_recv[j] + k * bytes on every member j with a non-null d_recv[j]; in stream order
    void all_gather(const std::vector<const void *> &d_send, const std::vector<void *> &d_recv, std::size_t bytes) const {
        if (d_send.size() != size() || d_recv.size() != size()) throw std::invalid_argument("device_group::all_gather: one pointer per member");
        check_group(zkhip_group_all_gather(g_, d_send.data(), d_recv.data(), bytes), "zkhip_group_all_gather");
    }
    /// d_dst on member dst <- d_src on member src, after src's stream so far, on dst's stream
    void copy(std::size_t dst, void *d_dst, std::size_t src, const void *d_src, std::size_t bytes) const {
        check_group(zkhip_group_copy(g_, (int)dst, d_dst, (int)src, d_src, bytes), "zkhip_group_copy");
    }
    void sync() const { check_group(zkhip_group_sync(g_), "zkhip_group_sync"); }

private:
    zkhip_device_group *g_ = nullptr;
    std::vector<std::unique_ptr<context>> members_;
};

/// Page-locked host memory that grows on demand and is reused: transfers at link speed, no first-touch page faults on reuse.
class pinned_buffer {
public:
    pinned_buffer() = default;
    ~pinned_buffer() { release(); }
    pinned_buffer(const pinned_buffer &) = delete;
    pinned_buffer &operator=(const pinned_buffer &) = delete;
    /// at least `bytes` bytes (contents are not preserved when it grows)
    void *reserve(const context &ctx, std::size_t bytes) {
        if (bytes > cap_) {
            release();
            check(zkhip_host_alloc(ctx.get(), bytes, &p_), "zkhip_host_alloc", ctx.get());
            ctx_ = ctx.get();
            cap_ = bytes;
        }
        return p_;
    }
    void *get() const { return p_; }
    std::size_t capacity() const { return cap_; }

private:
    void release() {
        if (p_) zkhip_host_free(ctx_, p_);
        p_ = nullptr;
        cap_ = 0;
    }
    void *p_ = nullptr;
    std::size_t cap_ = 0;
    zkhip_ctx *ctx_ = nullptr;
};

/// `count` scalar-field values -> canonical limbs in device memory at d_dst (32 bytes each).
/// A curve adapter whose scalar type IS four canonical little-endian u64 limbs says so (`scalars_are_canonical_limbs`): the
/// values then go out as they lie in the caller's vector, one copy at link speed.  Any other representation (crypto3-algebra's
/// Montgomery form) is converted slice by slice on a few host threads, each slice sent as soon as it is ready.
namespace detail {
    template <typename A, typename = void>
    struct canonical_scalars : std::false_type { };
    template <typename A>
    struct canonical_scalars<A, std::void_t<decltype(A::scalars_are_canonical_limbs)>> : std::integral_constant<bool, A::scalars_are_canonical_limbs> { };
}    // namespace detail
template <typename Adapter>
void upload_scalars(const context &ctx, void *d_dst, const typename Adapter::scalar_value_type *values, std::size_t count) {
    if (count == 0) return;
    if constexpr (detail::canonical_scalars<Adapter>::value) {
        static_assert(sizeof(typename Adapter::scalar_value_type) == 32, "canonical-limb scalars are 4 x u64");
        ctx.h2d(d_dst, values, count * 32);
    } else {
        const std::size_t slice = (std::size_t)1 << 18, nslices = (count + slice - 1) / slice;
        const std::size_t lanes = std::min<std::size_t>(nslices, std::max(2u, std::min(8u, std::thread::hardware_concurrency())));
        std::vector<std::vector<std::uint64_t>> stage(lanes);    // one staging slice per thread in flight
        for (std::size_t base = 0; base < nslices; base += lanes) {
            std::vector<std::future<void>> ready;
            for (std::size_t k = 0; k < lanes && base + k < nslices; ++k)
                ready.push_back(std::async(std::launch::async, [&, k]() {
                    const std::size_t lo = (base + k) * slice, hi = std::min(count, lo + slice);
                    stage[k].resize(4 * (hi - lo));
                    for (std::size_t i = lo; i < hi; ++i) Adapter::scalar_to_limbs(values[i], &stage[k][4 * (i - lo)]);
                }));
            for (std::size_t k = 0; k < ready.size(); ++k) {
                ready[k].get();
                const std::size_t lo = (base + k) * slice;
                ctx.h2d(static_cast<char *>(d_dst) + 32 * lo, stage[k].data(), stage[k].size() * 8);    // synchronous: the slice may be reused
            }
        }
    }
}

/// The way back: `count` canonical 32-byte elements at d_src -> scalar-field values appended to `out` (same two paths).
template <typename Adapter>
void download_scalars(const context &ctx, const void *d_src, std::size_t count, std::vector<typename Adapter::scalar_value_type> &out) {
    if (count == 0) return;
    const std::size_t at = out.size();
    out.resize(at + count);
    if constexpr (detail::canonical_scalars<Adapter>::value) {
        static_assert(sizeof(typename Adapter::scalar_value_type) == 32, "canonical-limb scalars are 4 x u64");
        ctx.d2h(out.data() + at, d_src, count * 32);
    } else {
        std::vector<std::uint64_t> h(4 * count);
        ctx.d2h(h.data(), d_src, h.size() * 8);
        const std::size_t lanes = count >= ((std::size_t)1 << 16) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1;
        std::vector<std::future<void>> work;
        for (std::size_t k = 0; k < lanes; ++k)
            work.push_back(std::async(lanes > 1 ? std::launch::async : std::launch::deferred, [&, k]() {
                for (std::size_t i = count * k / lanes; i < count * (k + 1) / lanes; ++i) out[at + i] = Adapter::scalar_from_limbs(&h[4 * i]);
            }));
        for (auto &w : work) w.get();
    }
}

/// Resident bases (a proving-key query or an SRS): uploaded once, reused for every proof / commitment.
template <typename CurveType, int Group>
class device_bases {
public:
    typedef curve_adapter<CurveType> adapter;
    device_bases() = default;
    /// from a range of group values (G::value_type of the reference; here adapter point types)
    template <typename InputIt>
    device_bases(const context &ctx, InputIt first, InputIt last) : ctx_(&ctx) {
        const std::size_t cl = Group == ZKHIP_G1 ? adapter::g1_coord_limbs : adapter::g2_coord_limbs;
        std::vector<std::uint64_t> xy;
        std::vector<std::uint8_t> inf;
        if constexpr (std::is_base_of<std::random_access_iterator_tag, typename std::iterator_traits<InputIt>::iterator_category>::value) {
            /* a query of a large key: the conversions (the adapter's to-affine + to-canonical per point) on a few host threads */
            const std::size_t n = (std::size_t)(last - first);
            xy.resize(n * 2 * cl);
            inf.resize(n);
            const std::size_t lanes = n >= ((std::size_t)1 << 14) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1;
            std::vector<std::future<void>> work;
            for (std::size_t k = 0; k < lanes; ++k)
                work.push_back(std::async(lanes > 1 ? std::launch::async : std::launch::deferred, [&, k]() {
                    for (std::size_t i = n * k / lanes; i < n * (k + 1) / lanes; ++i)
                        inf[i] = adapter::point_to_affine_limbs(first[i], xy.data() + i * 2 * cl) ? 0 : 1;
                }));
            for (auto &w : work) w.get();
        } else {
            for (InputIt it = first; it != last; ++it) {
                xy.resize(xy.size() + 2 * cl);
                inf.push_back(adapter::point_to_affine_limbs(*it, xy.data() + xy.size() - 2 * cl) ? 0 : 1);
            }
        }
        size_ = inf.size();
        check(zkhip_bases_upload(ctx.get(), adapter::id, Group, xy.data(), inf.data(), size_, &b_), "zkhip_bases_upload", ctx.get());
    }
    /// bases[i] = scalars[i] * G computed on the device (fixed-base batch exponentiation, generator.hpp:187-214)
    template <typename ScalarIt>
    static device_bases from_scalars(const context &ctx, ScalarIt first, ScalarIt last) {
        std::vector<std::uint64_t> s;
        for (ScalarIt it = first; it != last; ++it) {
            s.resize(s.size() + 4);
            adapter::scalar_to_limbs(*it, s.data() + s.size() - 4);
        }
        device_bases r;
        r.ctx_ = &ctx;
        r.size_ = s.size() / 4;
        check(zkhip_bases_from_scalars(ctx.get(), adapter::id, Group, nullptr, s.data(), r.size_, &r.b_), "zkhip_bases_from_scalars", ctx.get());
        return r;
    }
    /// from `n` compressed wire encodings (48 bytes per G1 point, 96 per G2 point), decoded on the device
    static device_bases from_compressed(const context &ctx, const std::uint8_t *octets, std::size_t n) {
        device_bases r;
        r.ctx_ = &ctx;
        r.size_ = n;
        check(zkhip_bases_upload_compressed(ctx.get(), adapter::id, Group, octets, n, &r.b_), "zkhip_bases_upload_compressed", ctx.get());
        return r;
    }
    /// these points at the rows d_rows[j] (device array of size() u32; nullptr: the consecutive rows first + j) of an object of
    /// n_total points, the point at infinity everywhere else (zkhip_bases_spread: queries laid out over the same rows share a sort)
    device_bases spread(const void *d_rows, std::size_t first, std::size_t n_total) const {
        device_bases r;
        r.ctx_ = ctx_;
        r.size_ = n_total;
        check(zkhip_bases_spread(ctx_->get(), b_, static_cast<const std::uint32_t *>(d_rows), first, n_total, &r.b_), "zkhip_bases_spread", ctx_->get());
        return r;
    }
    /// host copy of entry i as a group value
    typename std::conditional<Group == ZKHIP_G1, typename adapter::g1_value_type, typename adapter::g2_value_type>::type at(std::size_t i) const {
        typedef typename std::conditional<Group == ZKHIP_G1, typename adapter::g1_value_type, typename adapter::g2_value_type>::type G;
        const std::size_t cl = Group == ZKHIP_G1 ? adapter::g1_coord_limbs : adapter::g2_coord_limbs;
        std::vector<std::uint64_t> xy(2 * cl);
        std::uint8_t inf = 0;
        check(zkhip_bases_download(ctx_->get(), b_, i, 1, xy.data(), &inf), "zkhip_bases_download", ctx_->get());
        return G::from_affine(xy.data(), inf != 0);
    }
    /// a copy of these bases on ANOTHER context -- another GPU of a device group (an SRS replicated over the members): through the
    /// canonical affine form (one download, one upload; no conversion of group values on the host)
    device_bases replicate(const context &other) const {
        const std::size_t cl = Group == ZKHIP_G1 ? adapter::g1_coord_limbs : adapter::g2_coord_limbs;
        std::vector<std::uint64_t> xy(std::max<std::size_t>(1, size_) * 2 * cl);
        std::vector<std::uint8_t> inf(std::max<std::size_t>(1, size_));
        if (size_) check(zkhip_bases_download(ctx_->get(), b_, 0, size_, xy.data(), inf.data()), "zkhip_bases_download", ctx_->get());
        device_bases r;
        r.ctx_ = &other;
        r.size_ = size_;
        check(zkhip_bases_upload(other.get(), adapter::id, Group, xy.data(), inf.data(), size_, &r.b_), "zkhip_bases_upload", other.get());
        return r;
    }
    /// a second handle on the SAME resident points (they are read-only once built): for another prover lane on the same GPU
    /// (its own context / stream); `o` keeps the ownership and must outlive the alias
    static device_bases alias(const device_bases &o) {
        device_bases r;
        r.ctx_ = o.ctx_;
        r.b_ = o.b_;
        r.size_ = o.size_;
        r.owner_ = false;
        return r;
    }
    ~device_bases() {
        if (b_ && owner_) zkhip_bases_free(ctx_ ? ctx_->get() : nullptr, b_);
    }
    device_bases(device_bases &&o) noexcept : ctx_(o.ctx_), b_(o.b_), size_(o.size_), owner_(o.owner_) { o.b_ = nullptr; }
    device_bases &operator=(device_bases &&o) noexcept {
        if (this != &o) {
            if (b_ && owner_) zkhip_bases_free(ctx_ ? ctx_->get() : nullptr, b_);
            ctx_ = o.ctx_;
            b_ = o.b_;
            size_ = o.size_;
            owner_ = o.owner_;
            o.b_ = nullptr;
        }
        return *this;
    }
    device_bases(const device_bases &) = delete;
    device_bases &operator=(const device_bases &) = delete;
    const zkhip_bases *get() const { return b_; }
    std::size_t size() const { return size_; }

private:
    const context *ctx_ = nullptr;
    zkhip_bases *b_ = nullptr;
    std::size_t size_ = 0;
    bool owner_ = true;
};

/// Resident bases cut by point range over a device group's members (zkhip_group_bases_upload): the `bases_begin, bases_end` of a
/// multiexp whose chunks are GPUs.
template <typename CurveType, int Group>
class device_group_bases {
public:
    typedef curve_adapter<CurveType> adapter;
    template <typename InputIt>
    device_group_bases(const device_group &group, InputIt first, InputIt last) : group_(&group) {
        const std::size_t cl = Group == ZKHIP_G1 ? adapter::g1_coord_limbs : adapter::g2_coord_limbs;
        std::vector<std::uint64_t> xy;
        std::vector<std::uint8_t> inf;
        for (InputIt it = first; it != last; ++it) {
            xy.resize(xy.size() + 2 * cl);
            inf.push_back(adapter::point_to_affine_limbs(*it, xy.data() + xy.size() - 2 * cl) ? 0 : 1);
        }
        size_ = inf.size();
        group.check_group(zkhip_group_bases_upload(group.get(), adapter::id, Group, xy.data(), inf.data(), size_, &b_), "zkhip_group_bases_upload");
    }
    ~device_group_bases() {
        if (b_) zkhip_group_bases_free(group_->get(), b_);
    }
    device_group_bases(const device_group_bases &) = delete;
    device_group_bases &operator=(const device_group_bases &) = delete;
    const zkhip_group_bases *get() const { return b_; }
    std::size_t size() const { return size_; }
    const device_group &group() const { return *group_; }

private:
    const device_group *group_ = nullptr;
    zkhip_group_bases *b_ = nullptr;
    std::size_t size_ = 0;
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_BACKEND_HPP
