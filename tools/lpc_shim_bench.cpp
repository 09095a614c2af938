// LPC commit THROUGH THE SHIM CLASS at size: `cols` polynomial_dfs of 2^log_n rows on the host -> append_to_batch(&&) -> commit:
// upload, inverse NTTs, extension to D[0] = 2^(log_n + expand), coset-ordered leaf layout, download of the leaves, and a tree builder
// that only touches every leaf element (XOR fold) in place of the caller's hash.
#include <chrono>
#include <cstdio>
#include <nil/crypto3/zk/hip/lpc.hpp>
using namespace nil::crypto3::zk::hip;
typedef bls12_381 C;
typedef curve_adapter<C> A;
typedef A::scalar_value_type Fr;
struct transcript {
    template <typename T>
    void operator()(const T &) { }
    Fr challenge() { return Fr(12345); }
};
struct fold_tree {
    uint64_t r = 0;
    uint64_t root() const { return r; }
};
struct fold_builder {
    fold_tree operator()(const std::vector<Fr> &leaves, std::size_t) const {
        fold_tree t;
        for (const auto &v : leaves) t.r ^= v.limbs[0];
        return t;
    }
};
static uint64_t sm(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
int main(int argc, char **argv) {
    const size_t log_n = argc > 1 ? atoi(argv[1]) : 20, cols = argc > 2 ? atoi(argv[2]) : 16, expand = 1, n = (size_t)1 << log_n;
    context ctx(0);
    const uint64_t rm1[4] = {0xffffffff00000000ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    auto root = [&](std::size_t l) {
        uint64_t e[4] = {rm1[0], rm1[1], rm1[2], rm1[3]};
        for (std::size_t k = 0; k < l; ++k) {
            for (int i = 0; i < 3; ++i) e[i] = (e[i] >> 1) | (e[i + 1] << 63);
            e[3] >>= 1;
        }
        Fr r = Fr::one(), b(7);
        for (int i = 255; i >= 0; --i) {
            r = r * r;
            if ((e[i >> 6] >> (i & 63)) & 1) r = r * b;
        }
        return r;
    };
    fri_params_hip<C> params;
    params.log_domain = log_n + expand;
    params.step_list.assign(log_n + expand - 4, 1);    // fold down to 16 points, one step per round
    params.root_of_unity = root;
    uint64_t seed = 5;
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<polynomial_dfs<C>> polys(cols);
        for (auto &p : polys) {
            p.values.resize(n);
            for (auto &v : p.values) {
                uint64_t w[4] = {sm(seed), sm(seed), sm(seed), sm(seed) & 0x0fffffffffffffffull};
                v = A::scalar_from_limbs(w);
            }
        }
        lpc_commitment_scheme_hip<C, transcript, fold_builder> scheme(ctx, params, fold_builder());
        auto t0 = std::chrono::steady_clock::now();
        scheme.append_to_batch(0, std::move(polys));
        auto root0 = scheme.commit(0);
        auto t1 = std::chrono::steady_clock::now();
        printf("LPC commit of %zu x 2^%zu (domain 2^%zu) through the shim: %.1f ms (leaves: %.2f GB to the host)  [%016llx]\n", cols, log_n, log_n + expand,
               std::chrono::duration<double, std::milli>(t1 - t0).count(), (double)cols * (n << expand) * 32 / 1e9, (unsigned long long)root0);
    }
    return 0;
}
