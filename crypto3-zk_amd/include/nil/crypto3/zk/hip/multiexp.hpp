//---------------------------------------------------------------------------//
// zkhip shim: drop-in for algebra::multiexp<Method>(bases_begin, bases_end, scalars_begin, scalars_end, chunks)
// and algebra::multiexp_with_mixed_addition<Method>(...) as the reference calls them at
//   zk/snark/systems/ppzksnark/r1cs_gg_ppzksnark/prover.hpp:108-139
//   zk/commitments/polynomial/kzg.hpp:143-148, 409-435, 505-508
//   zk/commitments/polynomial/knowledge_commitment_multiexp.hpp:107
// Same argument order and meaning; `chunks` is accepted and ignored (the device splits the work itself).
// `multiexp_method_hip` is the POLICY a KZG parameter struct shadows `multiexp_method` with (kzg.hpp:82, 231 declare
// `using multiexp_method = ...BDLO12`).  The reference's call sites are QUALIFIED calls of crypto3-algebra's dispatcher,
//   algebra::multiexp<typename KZG::multiexp_method>(b0, b1, s0, s1, 1)          (kzg.hpp:146-147, 414-418, 433-434, 505-508, 661-662)
// which hands the range to the policy as `MultiexpMethod::process(b0, b1, s0, s1)` (one call when chunks == 1, one per chunk
// otherwise).  So the seam is the policy's static `process`: no context argument, the group read off the
// bases' value type, run on the calling thread's default context (default_context()).  tests/cpp/shim_test.cpp drives it
// through a dispatcher declared in ANOTHER namespace and called qualified, as kzg.hpp does.  The free functions
// multiexp<multiexp_method_hip>(b0, b1, s0, s1, chunks) at the end of this file are the same call for code that lives inside
// zk::hip (the shim's own schemes).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_MULTIEXP_HPP
#define ZKHIP_SHIM_MULTIEXP_HPP

#include <cstdlib>
#include <iterator>
#include <memory>
#include <type_traits>
#include <vector>

#include "backend.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// The device multiexp as a crypto3-algebra multiexp policy: `process(bases_begin, bases_end, scalars_begin, scalars_end)`
/// returns the sum as the bases' value type (defined below, after group_traits / default_context).
struct multiexp_method_hip {
    template <typename BaseIt, typename ScalarIt>
    static typename std::iterator_traits<BaseIt>::value_type process(BaseIt bases_begin, BaseIt bases_end, ScalarIt scalars_begin,
                                                                     ScalarIt scalars_end);
};

namespace detail {
    template <typename CurveType, typename ScalarIt>
    std::vector<std::uint64_t> pack_scalars(ScalarIt first, ScalarIt last) {
        std::vector<std::uint64_t> out;
        for (ScalarIt it = first; it != last; ++it) {
            out.resize(out.size() + 4);
            curve_adapter<CurveType>::scalar_to_limbs(*it, out.data() + out.size() - 4);
        }
        return out;
    }
    template <typename CurveType, int Group>
    struct jac_result;
    template <typename CurveType>
    struct jac_result<CurveType, ZKHIP_G1> {
        typedef typename curve_adapter<CurveType>::g1_value_type type;
        static constexpr std::size_t limbs = 3 * curve_adapter<CurveType>::g1_coord_limbs;
        static type make(const std::uint64_t *j) { return curve_adapter<CurveType>::g1_from_jacobian(j); }
    };
    template <typename CurveType>
    struct jac_result<CurveType, ZKHIP_G2> {
        typedef typename curve_adapter<CurveType>::g2_value_type type;
        static constexpr std::size_t limbs = 3 * curve_adapter<CurveType>::g2_coord_limbs;
        static type make(const std::uint64_t *j) { return curve_adapter<CurveType>::g2_from_jacobian(j); }
    };
}    // namespace detail

/// sum scalars[i] * bases[offset + i] over resident bases, scalars on the host.
template <typename CurveType, int Group, typename ScalarIt>
typename detail::jac_result<CurveType, Group>::type multiexp(const context &ctx, const device_bases<CurveType, Group> &bases, std::size_t offset,
                                                             ScalarIt scalars_begin, ScalarIt scalars_end, std::size_t /*chunks*/ = 1) {
    typedef detail::jac_result<CurveType, Group> R;
    std::vector<std::uint64_t> s = detail::pack_scalars<CurveType>(scalars_begin, scalars_end);
    std::uint64_t jac[R::limbs];
    check(zkhip_msm(ctx.get(), bases.get(), offset, s.size() / 4, s.data(), jac), "zkhip_msm", ctx.get());
    return R::make(jac);
}

/// Same over scalars already resident on the device (n elements at d_scalars).
template <typename CurveType, int Group>
typename detail::jac_result<CurveType, Group>::type multiexp_dev(const context &ctx, const device_bases<CurveType, Group> &bases, std::size_t offset,
                                                                 std::size_t n, const void *d_scalars) {
    typedef detail::jac_result<CurveType, Group> R;
    auto d_out = ctx.alloc(R::limbs * 8);
    check(zkhip_msm_dev(ctx.get(), bases.get(), offset, n, d_scalars, d_out.get()), "zkhip_msm_dev", ctx.get());
    std::uint64_t jac[R::limbs];
    ctx.d2h(jac, d_out.get(), sizeof(jac));
    return R::make(jac);
}

/// The reference's one-shot signature: bases given as an iterator range of group values (uploaded for this call).
template <typename Method, typename CurveType, int Group, typename BaseIt, typename ScalarIt>
typename detail::jac_result<CurveType, Group>::type multiexp(const context &ctx, BaseIt bases_begin, BaseIt bases_end, ScalarIt scalars_begin,
                                                             ScalarIt scalars_end, std::size_t chunks = 1) {
    device_bases<CurveType, Group> b(ctx, bases_begin, bases_end);
    return multiexp<CurveType, Group>(ctx, b, 0, scalars_begin, scalars_end, chunks);
}

/// multiexp_with_mixed_addition peels scalars 0 and 1 before the bucket method (prover.hpp:108-114); the device
/// path needs no peeling (0 -> no digit, 1 -> one bucket hit), so this is the same call.
template <typename Method, typename CurveType, int Group, typename BaseIt, typename ScalarIt>
typename detail::jac_result<CurveType, Group>::type multiexp_with_mixed_addition(const context &ctx, BaseIt bases_begin, BaseIt bases_end,
                                                                                 ScalarIt scalars_begin, ScalarIt scalars_end,
                                                                                 std::size_t chunks = 1) {
    return multiexp<Method, CurveType, Group>(ctx, bases_begin, bases_end, scalars_begin, scalars_end, chunks);
}

// ---- the reference's arity: no context argument ---------------------------------------------------------------------
/// Which curve / group a group value type belongs to (what algebra::multiexp reads off its iterators' value_type).  A
/// crypto3 maintainer specialises it for `typename curve_type::template g1_type<>::value_type` / g2 next to curve_adapter.
template <typename G>
struct group_traits;
template <int Curve, int Group>
struct group_traits<group_value<Curve, Group>> {
    typedef native_curve<Curve> curve_type;
    static constexpr int group = Group;
};

namespace detail {
    inline context *&default_context_override() {
        thread_local context *p = nullptr;
        return p;
    }
}    // namespace detail
namespace detail {
    inline device_group *&default_group_override() {
        thread_local device_group *p = nullptr;
        return p;
    }
}    // namespace detail
/// The device group the context-less entry points spread over -- the reference's static `process(proving_key, primary_input,
/// auxiliary_input)` (prover.hpp:73-75) among them --, or nullptr when the caller runs on one GPU: the group the caller installed
/// (set_default_group), else one per host thread over the devices ZKHIP_DEVICES names when it names more than one ("0,1,2,3").
inline const device_group *default_group() {
    if (device_group *p = detail::default_group_override()) return p;
    thread_local std::unique_ptr<device_group> own;
    thread_local bool looked = false;
    if (!looked) {
        const std::vector<int> devices = device_group::devices_from_env();
        if (devices.size() > 1) own.reset(new device_group(devices));    // throws for a device that does not exist -- at EVERY call, not only the first:
        looked = true;                                                      // a misconfigured ZKHIP_DEVICES must not quietly become one GPU
    }
    return own.get();
}
/// make `group` the calling thread's default group (nullptr: back to ZKHIP_DEVICES); the caller keeps it alive
inline void set_default_group(device_group *group) { detail::default_group_override() = group; }

/// The context the context-less overloads run on: one per host thread, created on first use on device ZKHIP_DEVICE (default 0)
/// -- "one context per GPU per process, a context is not thread-safe" (include/zkhip.h) -- unless the caller installed its own;
/// with a default group (above) and no ZKHIP_DEVICE it is the group's member 0.
inline const context &default_context() {
    if (context *p = detail::default_context_override()) return *p;
    if (!std::getenv("ZKHIP_DEVICE"))
        if (const device_group *g = default_group()) return g->root();
    thread_local std::unique_ptr<context> own;
    if (!own) {
        const char *e = std::getenv("ZKHIP_DEVICE");
        own.reset(new context(e ? std::atoi(e) : 0));
    }
    return *own;
}
/// make `ctx` the calling thread's default context (nullptr: back to the thread's own); the caller keeps it alive
inline void set_default_context(context *ctx) { detail::default_context_override() = ctx; }

/// sum scalars[i] * bases[offset + i] over bases cut over a device group: every member multiplies its point range, the partial sums
/// meet on member 0 inside the library (zkhip_group_msm).  `chunks` of the reference's call = the group's members.
template <typename CurveType, int Group, typename ScalarIt>
typename detail::jac_result<CurveType, Group>::type multiexp(const device_group_bases<CurveType, Group> &bases, std::size_t offset, ScalarIt scalars_begin,
                                                             ScalarIt scalars_end, std::size_t /*chunks*/ = 1) {
    typedef detail::jac_result<CurveType, Group> R;
    std::vector<std::uint64_t> s = detail::pack_scalars<CurveType>(scalars_begin, scalars_end);
    std::uint64_t jac[R::limbs];
    bases.group().check_group(zkhip_group_msm(bases.group().get(), bases.get(), offset, s.size() / 4, s.data(), jac), "zkhip_group_msm");
    return R::make(jac);
}

template <typename BaseIt, typename ScalarIt>
typename std::iterator_traits<BaseIt>::value_type multiexp_method_hip::process(BaseIt bases_begin, BaseIt bases_end, ScalarIt scalars_begin,
                                                                               ScalarIt scalars_end) {
    typedef group_traits<typename std::iterator_traits<BaseIt>::value_type> T;
    /* with a default device group (ZKHIP_DEVICES / set_default_group) the one-shot multiexp is cut over its GPUs */
    if (const device_group *g = default_group()) {
        device_group_bases<typename T::curve_type, T::group> b(*g, bases_begin, bases_end);
        return multiexp<typename T::curve_type, T::group>(b, 0, scalars_begin, scalars_end, 1);
    }
    return multiexp<multiexp_method_hip, typename T::curve_type, T::group>(default_context(), bases_begin, bases_end, scalars_begin, scalars_end, 1);
}

/// algebra::multiexp<Method>(bases_begin, bases_end, scalars_begin, scalars_end, chunks) with Method = multiexp_method_hip, for
/// callers inside zk::hip; `chunks` is ignored (the device splits the work itself), so this is ONE `process` call.
template <typename Method, typename BaseIt, typename ScalarIt, typename std::enable_if<std::is_same<Method, multiexp_method_hip>::value, bool>::type = true>
typename std::iterator_traits<BaseIt>::value_type multiexp(BaseIt bases_begin, BaseIt bases_end, ScalarIt scalars_begin, ScalarIt scalars_end,
                                                           std::size_t /*chunks*/) {
    return Method::process(bases_begin, bases_end, scalars_begin, scalars_end);
}
template <typename Method, typename BaseIt, typename ScalarIt, typename std::enable_if<std::is_same<Method, multiexp_method_hip>::value, bool>::type = true>
typename std::iterator_traits<BaseIt>::value_type multiexp_with_mixed_addition(BaseIt bases_begin, BaseIt bases_end, ScalarIt scalars_begin,
                                                                               ScalarIt scalars_end, std::size_t chunks) {
    return multiexp<Method>(bases_begin, bases_end, scalars_begin, scalars_end, chunks);
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_MULTIEXP_HPP
