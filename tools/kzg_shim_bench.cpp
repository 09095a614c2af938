// BASELINE config 5's commitment leg THROUGH THE SHIM CLASS, columns starting on the host as placeholder hands them over:
// kzg_commitment_scheme_v2_hip::append_to_batch + commit(batch) + proof_eval of 50 polynomial_dfs of 2^20 rows (upload included),
// for the three ways of handing the columns over.  The driver lives in crypto3-zk_amd/bench/scheme_bench.cpp (bench.py runs it too).
// Build: g++ -std=c++17 -O2 tools/kzg_shim_bench.cpp -L crypto3-zk_amd -lzkhip_bench -lzkhip -Wl,-rpath,$PWD/crypto3-zk_amd -o /tmp/kzg_shim_bench
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" int zkhip_bench_kzg_scheme(int device, size_t log_n, size_t cols, int steps, int mode, size_t upload_chunk, const uint64_t *evals, double *ms,
                                      uint64_t *out_commitments);
int main(int argc, char **argv) {
    const size_t log_n = argc > 1 ? atoi(argv[1]) : 20, cols = argc > 2 ? atoi(argv[2]) : 50, chunk = argc > 3 ? atoi(argv[3]) : 10, n = (size_t)1 << log_n;
    std::vector<uint64_t> evals(4 * n * cols);
    uint64_t x = 5;
    for (size_t i = 0; i < evals.size(); ++i) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        evals[i] = (z ^ (z >> 31)) & ((i & 3) == 3 ? 0x0fffffffffffffffull : ~0ull);
    }
    const char *names[3] = {"append_to_batch(const &): the reference's host copy", "append_to_batch(&&): handed over", "append_to_batch(std::cref): lent"};
    for (int mode = 0; mode < 3; ++mode) {
        double ms[9];
        if (zkhip_bench_kzg_scheme(0, log_n, cols, 3, mode, chunk, evals.data(), ms, nullptr)) return 1;
        for (int rep = 0; rep < 3; ++rep)
            printf("%zu x 2^%zu through the shim, %s: append %.1f ms, commit (upload + iNTT + MSM + download) %.1f ms, proof_eval at 2 points %.1f ms\n", cols, log_n,
                   names[mode], ms[3 * rep], ms[3 * rep + 1], ms[3 * rep + 2]);
    }
    return 0;
}
