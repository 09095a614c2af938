//---------------------------------------------------------------------------//
// zkhip shim: plonk columns -> coefficient form --
//   detail::column_polynomial / column_range_polynomials   zk/snark/arithmetization/plonk/detail/column_polynomial.hpp:43-72
// The reference copies every column and runs `domain->inverse_fft` on it, one column after the other; here a range of columns is
// uploaded side by side and inverted by ONE batched transform over the caller's evaluation domain (basic, step or extended radix-2).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_COLUMN_POLYNOMIAL_HPP
#define ZKHIP_SHIM_COLUMN_POLYNOMIAL_HPP

#include <memory>
#include <stdexcept>
#include <vector>

#include "evaluation_domain.hpp"
#include "kzg.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// the coefficient forms, RESIDENT: column i's domain.m coefficients at element offset i * domain.m of the returned buffer.
/// ColumnType: anything with size() and operator[] over the scalar field (plonk_column is a std::vector of field elements).
template <typename CurveType, typename ColumnType>
std::shared_ptr<void> column_range_polynomials_dev(const context &ctx, const std::vector<ColumnType> &column_range_assignment,
                                                   const evaluation_domain_hip<CurveType> &domain) {
    typedef curve_adapter<CurveType> adapter;
    const std::size_t m = domain.m, count = column_range_assignment.size();
    auto d = ctx.alloc(std::max<std::size_t>(1, count * m) * 32);
    for (std::size_t c = 0; c < count; ++c) {
        if (column_range_assignment[c].size() != m) throw std::invalid_argument("column_polynomial: a column's size differs from the domain's");
        upload_scalars<adapter>(ctx, static_cast<char *>(d.get()) + 32 * c * m, detail::poly_data<adapter>(column_range_assignment[c]), m);
    }
    if (count) domain.inverse_fft(ctx, d.get(), count);
    return d;
}

/// column_range_polynomials (:58-72) with the reference's host result: one coefficient vector per column
template <typename CurveType, typename ColumnType>
std::vector<std::vector<typename curve_adapter<CurveType>::scalar_value_type>>
column_range_polynomials(const context &ctx, const std::vector<ColumnType> &column_range_assignment, const evaluation_domain_hip<CurveType> &domain) {
    typedef curve_adapter<CurveType> adapter;
    auto d = column_range_polynomials_dev<CurveType>(ctx, column_range_assignment, domain);
    std::vector<std::vector<typename adapter::scalar_value_type>> out(column_range_assignment.size());
    for (std::size_t c = 0; c < out.size(); ++c) download_scalars<adapter>(ctx, static_cast<const char *>(d.get()) + 32 * c * domain.m, domain.m, out[c]);
    return out;
}

/// column_polynomial (:43-56)
template <typename CurveType, typename ColumnType>
std::vector<typename curve_adapter<CurveType>::scalar_value_type> column_polynomial(const context &ctx, const ColumnType &column_assignment,
                                                                                    const evaluation_domain_hip<CurveType> &domain) {
    return column_range_polynomials<CurveType>(ctx, std::vector<ColumnType> {column_assignment}, domain)[0];
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_COLUMN_POLYNOMIAL_HPP
