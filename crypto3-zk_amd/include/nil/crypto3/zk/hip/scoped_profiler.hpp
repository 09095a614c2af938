//---------------------------------------------------------------------------//
// zkhip shim: the reference's region profiler hook, for the regions that live INSIDE code this backend replaces.
//
// Mirrors zk/snark/systems/plonk/placeholder/detail/placeholder_scoped_profiler.hpp:39-127: the same compile-time switch
// (ZK_PLACEHOLDER_PROFILING_ENABLED), the same behaviour (wall time of the enclosing scope, printed as "<name>: <ms> ms" on
// std::cout when the scope is left), the same region NAMES.  Of the reference's scopes only one sits inside replaced code --
// "Basic FRI Precommit time" (commitments/detail/polynomial/basic_fri.hpp:449, the body of precommit<FRI>) --; the others
// ("variable_values_precommit_time", "T_splitted_precommit_time", "commitment scheme proof eval time", ph/prover.hpp:140, 315,
// 212, ...) wrap CALLS of the commitment scheme from placeholder_prover, which stays the reference's code and keeps timing them
// with a zkhip scheme plugged in.  Per-kernel device times come from zkhip_profile_* (include/zkhip.h).
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_SCOPED_PROFILER_HPP
#define ZKHIP_SHIM_SCOPED_PROFILER_HPP

#include <chrono>
#include <iomanip>
#include <iostream>
#include <string>

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {
namespace detail {
    class scoped_profiler {
    public:
        explicit scoped_profiler(std::string name) : start(std::chrono::high_resolution_clock::now()), name(std::move(name)) { }
        ~scoped_profiler() {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - start).count();
            std::cout << name << ": " << std::fixed << std::setprecision(3) << ms << " ms" << std::endl;
        }

    private:
        std::chrono::time_point<std::chrono::high_resolution_clock> start;
        std::string name;
    };
}    // namespace detail
}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#ifdef ZK_PLACEHOLDER_PROFILING_ENABLED
#define ZKHIP_PROFILE_SCOPE(name) nil::crypto3::zk::hip::detail::scoped_profiler zkhip_scope_profiler_(name);
#else
#define ZKHIP_PROFILE_SCOPE(name)
#endif

#endif    // ZKHIP_SHIM_SCOPED_PROFILER_HPP
