//---------------------------------------------------------------------------//
// zkhip shim: the evaluation domain math::make_evaluation_domain<Fr>(min_size) returns, as far as the QAP reduction
// reads it (zk/snark/reductions/r1cs_to_qap.hpp:138-139, 150-153, 229-230, 250-315):
//   m, get_domain_element, compute_vanishing_polynomial, evaluate_all_lagrange_polynomials, add_poly_z   (host, key side)
//   fft / inverse_fft                                                                                   (device vectors)
// crypto3-math is not part of the reference tree; the family (basic, extended, step radix-2) and the selection order are
// those of its libfqfft lineage (include/zkhip.h, "evaluation domains").  The roots of unity are the CALLER's
// (arithmetic_params<F> lives in crypto3-algebra): `domain_params` carries them, as it always carried omega.
//
// Which omega: basic -- the primitive m-th root; extended -- the primitive (m/2)-th root; step (m = big + small) -- the
// primitive (2 big)-th root, i.e. the `omega` member of the reference's domain object in all three cases.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_EVALUATION_DOMAIN_HPP
#define ZKHIP_SHIM_EVALUATION_DOMAIN_HPP

#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "backend.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

namespace detail {
    /// fn(lo, hi) over [0, n) in contiguous chunks on up to 32 host threads (the QAP evaluation is embarrassingly parallel)
    template <typename Fn>
    void parallel_chunks(std::size_t n, Fn fn) {
        const std::size_t hw = std::max<std::size_t>(1, std::min<std::size_t>(32, std::thread::hardware_concurrency()));
        const std::size_t parts = n < (std::size_t)1 << 14 ? 1 : hw, per = (n + parts - 1) / parts;
        std::vector<std::thread> th;
        for (std::size_t k = 1; k < parts; ++k)
            if (k * per < n) th.emplace_back([=]() { fn(k * per, std::min(n, (k + 1) * per)); });
        fn(0, std::min(n, per));
        for (auto &t : th) t.join();
    }
    template <typename Fr>
    Fr pow_u64(Fr b, std::uint64_t e) {
        Fr r = Fr::one();
        for (; e; e >>= 1) {
            if (e & 1) r = r * b;
            b = b * b;
        }
        return r;
    }
    inline std::size_t ceil_log2(std::size_t n) {
        std::size_t r = 0;
        while (((std::size_t)1 << r) < n) ++r;
        return r;
    }
}    // namespace detail

/// The evaluation-domain constants the reference takes from crypto3-algebra / crypto3-math.
///   kind < 0 ("auto"): the domain make_evaluation_domain(num_constraints + num_inputs + 1) picks -- or, for a key whose
///   H_query says so, the basic domain of the next power of two (basic and step radix-2 use the SAME omega: the primitive
///   2^ceil(log2)-th root; the extended domain's is the primitive 2^s-th root, s the field's two-adicity);
///   kind >= 0: exactly this domain (m points).
template <typename CurveType>
struct domain_params {
    typename curve_adapter<CurveType>::scalar_value_type omega;             // see the header comment
    typename curve_adapter<CurveType>::scalar_value_type coset_generator;    // arithmetic_params<F>::multiplicative_generator
    int kind = -1;                                                           // zkhip_domain_kind, or -1 = auto
    std::size_t m = 0;                                                       // number of points (kind >= 0)
    typename curve_adapter<CurveType>::scalar_value_type shift;              // extended radix-2: detail::coset_shift<F>()
};

/// The constants of the domain make_evaluation_domain(min_size) returns, from the adapter's field constants (curve_adapter::
/// multiplicative_generator / root_of_unity: for the reference's curve types they forward to arithmetic_params<F> and math::unity_root,
/// INTEGRATION.md): kind "auto", omega the primitive 2^ceil(log2 min_size)-th root, the extended domain's shift generator^2
/// (detail::coset_shift<F>()).  What lets the entry points below take the reference's own argument lists.
template <typename CurveType>
domain_params<CurveType> standard_domain_params(std::size_t min_size) {
    typedef curve_adapter<CurveType> adapter;
    static_assert(adapter::has_field_constants, "this curve adapter does not provide multiplicative_generator() / root_of_unity()");
    domain_params<CurveType> d;
    /* basic and step radix-2 use the primitive 2^ceil(log2 min_size)-th root; the EXTENDED domain (2^(s + 1) points for a field of
       two-adicity s: two cosets of the largest radix-2 subgroup) uses the primitive 2^s-th root -- a 2^(s + 1)-th one does not
       exist.  Ask the selection first. */
    int kind = ZKHIP_DOMAIN_BASIC_RADIX2;
    std::size_t m = 0;
    check(zkhip_domain_choice(adapter::id, min_size, &kind, &m), "zkhip_domain_choice");
    d.omega = adapter::root_of_unity(kind == ZKHIP_DOMAIN_EXTENDED_RADIX2 ? detail::ceil_log2(m) - 1 : detail::ceil_log2(min_size));
    d.coset_generator = adapter::multiplicative_generator();
    d.shift = d.coset_generator * d.coset_generator;
    return d;
}

template <typename CurveType>
class evaluation_domain_hip {
public:
    typedef curve_adapter<CurveType> adapter;
    typedef typename adapter::scalar_value_type value_type;

    int kind = ZKHIP_DOMAIN_BASIC_RADIX2;
    std::size_t m = 0, big_m = 0, small_m = 0;    // step: m = big_m + small_m; extended: small_m = m / 2
    value_type omega, shift, big_omega, small_omega;

    evaluation_domain_hip() = default;
    evaluation_domain_hip(int kind_, std::size_t m_, const value_type &omega_, const value_type &shift_ = value_type::zero()) :
        kind(kind_), m(m_), omega(omega_), shift(shift_) {
        if (m <= 1) throw std::invalid_argument("evaluation domain: expected m > 1");
        if (kind == ZKHIP_DOMAIN_EXTENDED_RADIX2) {
            small_m = m / 2;
        } else if (kind == ZKHIP_DOMAIN_STEP_RADIX2) {
            big_m = (std::size_t)1 << (detail::ceil_log2(m) - 1);
            small_m = m - big_m;
            if (small_m != (std::size_t)1 << detail::ceil_log2(small_m)) throw std::invalid_argument("step_radix2(): expected small_m == 1ul<<log2(small_m)");
            big_omega = omega * omega;
            small_omega = detail::pow_u64(omega, 2 * (big_m / small_m));
        } else if (kind != ZKHIP_DOMAIN_BASIC_RADIX2) {
            throw std::invalid_argument("evaluation domain: unknown kind");
        }
    }

    /// (kind, m) of make_evaluation_domain(min_size) over this curve's scalar field
    static std::pair<int, std::size_t> choice(std::size_t min_size) {
        int k = 0;
        std::size_t mm = 0;
        check(zkhip_domain_choice(adapter::id, min_size, &k, &mm), "zkhip_domain_choice");
        return {k, mm};
    }
    /// The domain `dom` describes for an instance that needs `min_size` points (kind < 0: make_evaluation_domain's choice).
    static evaluation_domain_hip make(const domain_params<CurveType> &dom, std::size_t min_size) {
        if (dom.kind >= 0) {
            if (dom.m < min_size) throw std::invalid_argument("evaluation domain: " + std::to_string(dom.m) + " points do not hold " + std::to_string(min_size));
            return evaluation_domain_hip(dom.kind, dom.m, dom.omega, dom.shift);
        }
        const auto c = choice(min_size);
        return evaluation_domain_hip(c.first, c.second, dom.omega, dom.shift);
    }

    std::size_t size() const { return m; }
    zkhip_domain c_desc() const {
        zkhip_domain d;
        d.kind = kind;
        d.reserved = 0;
        d.m = m;
        adapter::scalar_to_limbs(omega, d.omega);
        adapter::scalar_to_limbs(shift, d.shift);
        return d;
    }

    value_type get_domain_element(std::size_t idx) const {
        if (kind == ZKHIP_DOMAIN_BASIC_RADIX2) return detail::pow_u64(omega, idx);
        if (kind == ZKHIP_DOMAIN_EXTENDED_RADIX2) return idx < small_m ? detail::pow_u64(omega, idx) : shift * detail::pow_u64(omega, idx - small_m);
        return idx < big_m ? detail::pow_u64(big_omega, idx) : omega * detail::pow_u64(small_omega, idx - big_m);
    }

    value_type compute_vanishing_polynomial(const value_type &t) const {
        const value_type one = value_type::one();
        if (kind == ZKHIP_DOMAIN_BASIC_RADIX2) return detail::pow_u64(t, m) - one;
        if (kind == ZKHIP_DOMAIN_EXTENDED_RADIX2) {
            const value_type tm = detail::pow_u64(t, small_m);
            return (tm - one) * (tm - detail::pow_u64(shift, small_m));
        }
        return (detail::pow_u64(t, big_m) - one) * (detail::pow_u64(t, small_m) - detail::pow_u64(omega, small_m));
    }

    /// L_i(t) for every point of the domain, in get_domain_element order (r1cs_to_qap.hpp:152-153).  t on the domain itself
    /// (probability m / r) is refused: the closed forms do not cover it and a trapdoor must not lie there anyway.
    std::vector<value_type> evaluate_all_lagrange_polynomials(const value_type &t) const {
        if (compute_vanishing_polynomial(t).is_zero()) throw std::invalid_argument("evaluate_all_lagrange_polynomials: t lies in the evaluation domain");
        const value_type one = value_type::one();
        std::vector<value_type> u(m);
        if (kind == ZKHIP_DOMAIN_BASIC_RADIX2) {
            basic_lagrange(m, omega, t, one, u.data());
        } else if (kind == ZKHIP_DOMAIN_EXTENDED_RADIX2) {
            const value_type t_sm = detail::pow_u64(t, small_m), s_sm = detail::pow_u64(shift, small_m), ood = (s_sm - one).inversed();
            basic_lagrange(small_m, omega, t, (s_sm - t_sm) * ood, u.data());
            basic_lagrange(small_m, omega, t * shift.inversed(), (t_sm - one) * ood, u.data() + small_m);
        } else {
            /* big part: l_i(t) over <omega^2> times (t^small - omega^small) / (x_i^small - omega^small); the denominator takes
               big / small distinct values.  small part: l_i(t / omega) over <small_omega> times (t^big - 1) / (omega^big - 1) */
            const std::size_t compr = big_m / small_m;
            const value_type w_sm = detail::pow_u64(omega, small_m), L0 = detail::pow_u64(t, small_m) - w_sm, step = detail::pow_u64(big_omega, small_m);
            std::vector<value_type> dinv(compr);
            value_type elt = one;
            for (std::size_t j = 0; j < compr; ++j) {
                dinv[j] = elt - w_sm;
                elt = elt * step;
            }
            batch_invert(dinv);
            basic_lagrange(big_m, big_omega, t, L0, u.data());
            detail::parallel_chunks(big_m, [&](std::size_t lo, std::size_t hi) {
                for (std::size_t i = lo; i < hi; ++i) u[i] = u[i] * dinv[i % compr];
            });
            const value_type L1 = (detail::pow_u64(t, big_m) - one) * (detail::pow_u64(omega, big_m) - one).inversed();
            basic_lagrange(small_m, small_omega, t * omega.inversed(), L1, u.data() + big_m);
        }
        return u;
    }

    /// The same values computed ON THE DEVICE (zkhip_domain_lagrange_dev: one batched inversion per 16 points) and downloaded: what
    /// the key generator uses -- the host form above takes about a second at 2^20 points, this one milliseconds.
    std::vector<value_type> evaluate_all_lagrange_polynomials(const context &ctx, const value_type &t) const {
        if (compute_vanishing_polynomial(t).is_zero()) throw std::invalid_argument("evaluate_all_lagrange_polynomials: t lies in the evaluation domain");
        const zkhip_domain d = c_desc();
        std::uint64_t tl[4];
        adapter::scalar_to_limbs(t, tl);
        auto d_u = ctx.alloc(m * 32);
        check(zkhip_domain_lagrange_dev(ctx.get(), adapter::id, &d, tl, d_u.get()), "zkhip_domain_lagrange_dev", ctx.get());
        std::vector<value_type> u;
        download_scalars<adapter>(ctx, d_u.get(), m, u);
        return u;
    }

    /// H += coeff * Z (r1cs_to_qap.hpp:261; H has m + 1 coefficients)
    void add_poly_z(const value_type &coeff, std::vector<value_type> &H) const {
        if (H.size() != m + 1) throw std::invalid_argument("add_poly_z: expected H.size() == m + 1");
        if (kind == ZKHIP_DOMAIN_BASIC_RADIX2) {
            H[m] = H[m] + coeff;
            H[0] = H[0] - coeff;
        } else if (kind == ZKHIP_DOMAIN_EXTENDED_RADIX2) {
            const value_type s_sm = detail::pow_u64(shift, small_m);
            H[2 * small_m] = H[2 * small_m] + coeff;
            H[small_m] = H[small_m] - coeff * (s_sm + value_type::one());
            H[0] = H[0] + coeff * s_sm;
        } else {
            const value_type w_sm = detail::pow_u64(omega, small_m);
            H[m] = H[m] + coeff;
            H[big_m] = H[big_m] - coeff * w_sm;
            H[small_m] = H[small_m] - coeff;
            H[0] = H[0] + coeff * w_sm;
        }
    }

    /// evaluation_domain::fft / inverse_fft on `batch` resident vectors of m elements each (in place); `coset`: multiply_by_coset
    /// fused in (before a forward, after an inverse transform -- by its inverse)
    void fft(const context &ctx, void *d_data, std::size_t batch = 1, const value_type *coset = nullptr) const { run(ctx, d_data, batch, 0, coset); }
    void inverse_fft(const context &ctx, void *d_data, std::size_t batch = 1, const value_type *coset = nullptr) const { run(ctx, d_data, batch, 1, coset); }

private:
    void run(const context &ctx, void *d_data, std::size_t batch, int inverse, const value_type *coset) const {
        const zkhip_domain d = c_desc();
        std::uint64_t g[4];
        if (coset) adapter::scalar_to_limbs(*coset, g);
        check(zkhip_domain_fft_dev(ctx.get(), adapter::id, &d, d_data, batch, inverse, coset ? g : nullptr), "zkhip_domain_fft_dev", ctx.get());
    }
    static void batch_invert(std::vector<value_type> &v) {
        std::vector<value_type> pre(v.size());
        value_type acc = value_type::one();
        for (std::size_t i = 0; i < v.size(); ++i) {
            pre[i] = acc;
            acc = acc * v[i];
        }
        value_type inv = acc.inversed();
        for (std::size_t i = v.size(); i-- > 0;) {
            const value_type x = v[i];
            v[i] = inv * pre[i];
            inv = inv * x;
        }
    }
    /// out[i] = scale * l_i(t) over {w^i, i < n}: l_i(t) = (t^n - 1) w^i / (n (t - w^i)); one inversion per chunk (Montgomery's trick)
    static void basic_lagrange(std::size_t n, const value_type &w, const value_type &t, const value_type &scale, value_type *out) {
        const value_type z_over_n = (detail::pow_u64(t, n) - value_type::one()) * value_type((std::uint64_t)n).inversed() * scale;
        detail::parallel_chunks(n, [&](std::size_t lo, std::size_t hi) {
            std::vector<value_type> pre(hi - lo);
            value_type x = detail::pow_u64(w, lo), acc = value_type::one();
            for (std::size_t i = lo; i < hi; ++i) {
                out[i] = x;    // w^i for now
                pre[i - lo] = acc;
                acc = acc * (t - x);
                x = x * w;
            }
            value_type inv = acc.inversed();
            for (std::size_t i = hi; i-- > lo;) {
                const value_type den = t - out[i];
                out[i] = out[i] * z_over_n * (inv * pre[i - lo]);
                inv = inv * den;
            }
        });
    }
};

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_EVALUATION_DOMAIN_HPP
