//---------------------------------------------------------------------------//
// zkhip shim: kc_multiexp_with_mixed_addition over a sparse vector of (G2, G1) pairs, on the MI355X.
//
// Mirrors zk/commitments/polynomial/knowledge_commitment_multiexp.hpp:57-108 (the B-query evaluation of the
// Groth16 prover, r1cs_gg_ppzksnark/prover.hpp:116-123):
//   walk vec.indices in [min_idx, max_idx); scalar = scalar_start[index - min_idx];
//   skip zeros, add ones directly, multiexp over the rest; return (sum g, sum h).
// On the device the scalars are gathered by index (zkhip_fr_gather_dev) and the two dense MSMs run over the
// selected sub-range of the resident bases; zeros cost nothing and ones land in one bucket, so no peeling pass.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_KNOWLEDGE_COMMITMENT_MULTIEXP_HPP
#define ZKHIP_SHIM_KNOWLEDGE_COMMITMENT_MULTIEXP_HPP

#include <algorithm>
#include <vector>

#include "r1cs_gg_ppzksnark.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// knowledge_commitment_vector (container::sparse_vector of element_kc) resident on the device
template <typename CurveType>
class device_kc_vector {
public:
    typedef curve_adapter<CurveType> adapter;
    device_kc_vector(const context &ctx, const knowledge_commitment_vector<CurveType> &vec) : indices(vec.indices), domain_size_(vec.domain_size_), ctx_(&ctx) {
        std::vector<typename adapter::g2_value_type> g;
        std::vector<typename adapter::g1_value_type> h;
        std::vector<std::uint32_t> idx;
        for (std::size_t i = 0; i < vec.values.size(); ++i) {
            g.push_back(vec.values[i].g);
            h.push_back(vec.values[i].h);
            idx.push_back((std::uint32_t)vec.indices[i]);
        }
        if (!std::is_sorted(indices.begin(), indices.end())) throw std::runtime_error("device_kc_vector: indices must be sorted");
        g_bases = device_bases<CurveType, ZKHIP_G2>(ctx, g.begin(), g.end());
        h_bases = device_bases<CurveType, ZKHIP_G1>(ctx, h.begin(), h.end());
        d_indices = ctx.alloc(std::max<std::size_t>(1, idx.size()) * 4);
        if (!idx.empty()) ctx.h2d(d_indices.get(), idx.data(), idx.size() * 4);
    }
    const context &ctx() const { return *ctx_; }
    std::vector<std::size_t> indices;    // host copy (sorted), for the [min_idx, max_idx) search
    std::size_t domain_size_;
    device_bases<CurveType, ZKHIP_G2> g_bases;
    device_bases<CurveType, ZKHIP_G1> h_bases;
    std::shared_ptr<void> d_indices;

private:
    const context *ctx_;
};

/// kc_multiexp_with_mixed_addition<MultiexpMethod>(vec, min_idx, max_idx, scalar_start, scalar_end, chunks)
template <typename MultiexpMethod, typename CurveType, typename InputFieldIterator>
element_kc<CurveType> kc_multiexp_with_mixed_addition(const device_kc_vector<CurveType> &vec, const std::size_t min_idx, const std::size_t max_idx,
                                                      InputFieldIterator scalar_start, InputFieldIterator scalar_end, const std::size_t /*chunks*/) {
    typedef curve_adapter<CurveType> adapter;
    const context &ctx = vec.ctx();
    const std::size_t scalar_length = std::distance(scalar_start, scalar_end);
    if (scalar_length > vec.domain_size_) throw std::runtime_error("kc_multiexp_with_mixed_addition: more scalars than the vector's domain");
    const std::size_t lo = std::lower_bound(vec.indices.begin(), vec.indices.end(), min_idx) - vec.indices.begin();
    const std::size_t hi = std::lower_bound(vec.indices.begin(), vec.indices.end(), max_idx) - vec.indices.begin();
    element_kc<CurveType> acc {adapter::g2_value_type::zero(), adapter::g1_value_type::zero()};
    if (hi <= lo) return acc;
    if (vec.indices[hi - 1] - min_idx >= scalar_length) throw std::runtime_error("kc_multiexp_with_mixed_addition: index beyond the scalar range");
    std::vector<std::uint64_t> s = detail::pack_scalars<CurveType>(scalar_start, scalar_end);
    auto d_s = ctx.alloc(std::max<std::size_t>(1, s.size()) * 8);
    ctx.h2d(d_s.get(), s.data(), s.size() * 8);
    /* selected[k] = scalars[indices[lo + k] - min_idx]: the gather reads d_src[index], so shift the source */
    const std::size_t count = hi - lo;
    auto d_sel = ctx.alloc(count * 32);
    const char *src = static_cast<const char *>(d_s.get()) - 32 * min_idx;
    const char *idx = static_cast<const char *>(vec.d_indices.get()) + 4 * lo;
    check(zkhip_fr_gather_dev(ctx.get(), src, min_idx + scalar_length, idx, count, d_sel.get()), "zkhip_fr_gather_dev", ctx.get());
    acc.g = multiexp_dev<CurveType, ZKHIP_G2>(ctx, vec.g_bases, lo, count, d_sel.get());
    acc.h = multiexp_dev<CurveType, ZKHIP_G1>(ctx, vec.h_bases, lo, count, d_sel.get());
    return acc;
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_KNOWLEDGE_COMMITMENT_MULTIEXP_HPP
