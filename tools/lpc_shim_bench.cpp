// LPC commit THROUGH THE SHIM CLASS at size: `cols` polynomial_dfs of 2^log_n rows on the host -> append_to_batch (lent) -> commit:
// upload, inverse NTTs, extension to D[0] = 2^(log_n + expand), coset-ordered leaf layout, and the leaves to a tree builder that only
// touches every leaf element (XOR fold) in place of the caller's hash -- in the streaming shape (slices absorbed while the next one
// crosses PCIe) and in round 2's std::vector shape.  The driver lives in crypto3-zk_amd/bench/scheme_bench.cpp.
// Build: g++ -std=c++17 -O2 tools/lpc_shim_bench.cpp -L crypto3-zk_amd -lzkhip_bench -lzkhip -Wl,-rpath,$PWD/crypto3-zk_amd -o /tmp/lpc_shim_bench
#include <cstdint>
#include <cstdio>
#include <cstdlib>
extern "C" int zkhip_bench_lpc_scheme(int device, size_t log_n, size_t cols, size_t expand, int steps, int streaming, unsigned threads, double *ms,
                                      uint64_t *root);
int main(int argc, char **argv) {
    const size_t log_n = argc > 1 ? atoi(argv[1]) : 20, cols = argc > 2 ? atoi(argv[2]) : 16;
    const unsigned threads = argc > 3 ? atoi(argv[3]) : 8;
    for (int streaming = 1; streaming >= 0; --streaming) {
        double ms[4];
        uint64_t root = 0;
        if (zkhip_bench_lpc_scheme(0, log_n, cols, 1, 4, streaming, threads, ms, &root)) return 1;
        printf("LPC commit of %zu x 2^%zu (domain 2^%zu, %.2f GB of leaves), %s: %.1f %.1f %.1f %.1f ms  [%016llx]\n", cols, log_n, log_n + 1,
               (double)cols * ((size_t)2 << log_n) * 32 / 1e9, streaming ? "streaming builder" : "std::vector builder", ms[0], ms[1], ms[2], ms[3],
               (unsigned long long)root);
    }
    return 0;
}
