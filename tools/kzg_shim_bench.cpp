// BASELINE config 5's commitment leg THROUGH THE SHIM CLASS, columns starting on the host as placeholder hands them over:
// kzg_commitment_scheme_v2_hip::append_to_batch + commit(batch) of 50 polynomial_dfs of 2^20 rows (upload included).
// Build: g++ -std=c++17 -O2 -pthread -I crypto3-zk_amd/include -I include tools/kzg_shim_bench.cpp -L crypto3-zk_amd -lzkhip -Wl,-rpath,$PWD/crypto3-zk_amd -o /tmp/kzg_shim_bench
#include <chrono>
#include <cstdio>
#include <nil/crypto3/zk/hip/kzg_v2.hpp>
using namespace nil::crypto3::zk::hip;
typedef bls12_381 C;
typedef curve_adapter<C> A;
typedef A::scalar_value_type Fr;
struct transcript {
    void operator()(const A::g1_value_type &) { }
    void operator()(const Fr &) { }
    Fr challenge() { return Fr(12345); }
};
static uint64_t sm(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
int main(int argc, char **argv) {
    const size_t log_n = argc > 1 ? atoi(argv[1]) : 20, cols = argc > 2 ? atoi(argv[2]) : 50, n = (size_t)1 << log_n;
    context ctx(0);
    std::vector<Fr> pw(n);
    Fr x = Fr::one(), alpha(7);
    for (size_t i = 0; i < n; ++i) pw[i] = x, x = x * alpha;
    kzg_params_hip<C> params(ctx, device_bases<C, ZKHIP_G1>::from_scalars(ctx, pw.begin(), pw.end()));
    // omega = 7^((r - 1) / n) by square-and-multiply on the host
    const uint64_t rm1[4] = {0xffffffff00000000ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    auto root = [&](std::size_t l) {
        uint64_t e[4] = {rm1[0], rm1[1], rm1[2], rm1[3]};
        for (std::size_t k = 0; k < l; ++k) {  // e >>= 1
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
    uint64_t seed = 5;
    std::vector<polynomial_dfs<C>> polys(cols);
    for (auto &p : polys) {
        p.values.resize(n);
        for (auto &v : p.values) {
            uint64_t w[4] = {sm(seed), sm(seed), sm(seed), sm(seed) & 0x0fffffffffffffffull};
            v = A::scalar_from_limbs(w);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        kzg_commitment_scheme_v2_hip<C, transcript> scheme(params, root);
        auto t0 = std::chrono::steady_clock::now();
        scheme.append_to_batch(0, polys);
        auto t1 = std::chrono::steady_clock::now();
        auto commits = scheme.commit(0);
        auto t2 = std::chrono::steady_clock::now();
        scheme.append_eval_point(0, Fr(1234567));
        scheme.append_eval_point(0, Fr(7654321));
        transcript tr;
        auto t3 = std::chrono::steady_clock::now();
        auto proof = scheme.proof_eval(tr);
        auto t4 = std::chrono::steady_clock::now();
        printf("proof_eval of the same %zu polynomials at 2 points through the shim: %.1f ms\n", cols, std::chrono::duration<double, std::milli>(t4 - t3).count());
        uint64_t xy[12];
        proof.pi_1.to_affine(xy);
        commits[0].to_affine(xy);
        printf("commit of %zu x 2^%zu through the shim: append_to_batch (host copy) %.1f ms, commit (upload + iNTT + MSM + download) %.1f ms  [%016llx]\n", cols,
               log_n, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
               (unsigned long long)xy[0]);
    }
    return 0;
}
