// Groth16 prover throughput driver for bench.py / tools/bench_groth16.py: runs the header-only shim
// (r1cs_gg_ppzksnark_prover_hip::process) on a synthetic instance and reports wall time per proof.
// Host compiler only; links libzkhip.so.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <nil/crypto3/zk/hip/r1cs_gg_ppzksnark.hpp>

using namespace nil::crypto3::zk::hip;

namespace {

struct SplitMix {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
};

// Groth16 prover throughput on a synthetic instance of the reference's example family
// (generate_r1cs_example_with_field_input, r1cs_examples.hpp:77-140): the constraint system and a satisfying
// assignment are real; the proving-key queries are random multiples of the generators computed on the device
// (the prover's work does not depend on the key being a valid setup).  times[]: wall ms per proof.
// rank / world > 1: ONE proof sharded over `world` GPUs (one process each): this rank generates and holds only its
// slice of every query, runs the replicated witness map and its five partial MSMs, and `all_gather` (supplied by the
// caller: RCCL through torch.distributed in bench.py) exchanges the 864-byte partial sums.
typedef void (*all_gather_fn)(const uint64_t *mine, size_t words, uint64_t *all);

template <typename Curve>
int groth16_bench_t(int device, size_t rank, size_t world, all_gather_fn all_gather, size_t M, size_t n, uint64_t seed, int steps, const uint64_t *omega,
                    const uint64_t *coset, double *times, double *setup_ms, char *prof, size_t prof_cap) {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    auto t0 = std::chrono::steady_clock::now();
    SplitMix rng {seed};
    auto rnd = [&]() {
        uint64_t w[4] = {rng.next(), rng.next(), rng.next(), rng.next() & 0x0fffffffffffffffULL};
        return A::scalar_from_limbs(w);
    };
    r1cs_gg_ppzksnark_proving_key<Curve> pk;
    auto &cs = pk.constraint_system;
    cs.primary_input_size = n;
    cs.auxiliary_input_size = 2 + M - n;
    std::vector<Fr> full;
    Fr a = rnd(), b = rnd();
    full.push_back(a);
    full.push_back(b);
    for (size_t i = 0; i + 1 < M; ++i) {
        r1cs_constraint<Curve> c;
        Fr tmp;
        if (i % 2) {
            c.a.add_term(i + 1, 1);
            c.b.add_term(i + 2, 1);
            tmp = a * b;
        } else {
            c.b.add_term(0, 1);
            c.a.add_term(i + 1, 1);
            c.a.add_term(i + 2, 1);
            tmp = a + b;
        }
        c.c.add_term(i + 3, 1);
        full.push_back(tmp);
        a = b;
        b = tmp;
        cs.add_constraint(c);
    }
    {
        r1cs_constraint<Curve> c;
        Fr fin = Fr::zero();
        for (size_t i = 1; i < cs.num_variables(); ++i) {
            c.a.add_term(i, 1);
            c.b.add_term(i, 1);
            fin = fin + full[i - 1];
        }
        c.c.add_term(cs.num_variables(), 1);
        cs.add_constraint(c);
        full.push_back(fin * fin);
    }
    const size_t N = cs.num_variables();
    std::vector<Fr> primary(full.begin(), full.begin() + n), auxiliary(full.begin() + n, full.end());

    context ctx(device);
    size_t m = 1;
    while (m < M + n + 1) m <<= 1;
    const query_shard slice = query_shard::make(rank, world, N + 1, N + 1, m - 1, N - n);
    SplitMix key_rng {seed * 1000003 + 17 * rank + 1};    // key material differs per rank (each rank owns other points)
    auto rnd_key = [&]() {
        uint64_t w[4] = {key_rng.next(), key_rng.next(), key_rng.next(), key_rng.next() & 0x0fffffffffffffffULL};
        return A::scalar_from_limbs(w);
    };
    auto rand_bases_g1 = [&](size_t cnt) {
        std::vector<Fr> s(cnt);
        for (auto &x : s) x = rnd_key();
        return device_bases<Curve, ZKHIP_G1>::from_scalars(ctx, s.begin(), s.end());
    };
    std::vector<Fr> sb(slice.B_n);
    for (auto &x : sb) x = rnd_key();
    std::vector<uint32_t> bidx(slice.B_n);
    for (size_t i = 0; i < slice.B_n; ++i) bidx[i] = (uint32_t)(slice.B_lo + i);
    std::vector<Fr> fx = {rnd(), rnd(), rnd()};
    auto f1 = device_bases<Curve, ZKHIP_G1>::from_scalars(ctx, fx.begin(), fx.end());
    auto f2 = device_bases<Curve, ZKHIP_G2>::from_scalars(ctx, fx.begin(), fx.end());
    pk.alpha_g1 = f1.at(0);
    pk.beta_g1 = f1.at(1);
    pk.delta_g1 = f1.at(2);
    pk.beta_g2 = f2.at(1);
    pk.delta_g2 = f2.at(2);
    domain_params<Curve> dom {A::scalar_from_limbs(omega), A::scalar_from_limbs(coset)};
    r1cs_gg_ppzksnark_proving_key_hip<Curve> dpk(ctx, pk, dom, rand_bases_g1(slice.A_n), device_bases<Curve, ZKHIP_G2>::from_scalars(ctx, sb.begin(), sb.end()),
                                                 device_bases<Curve, ZKHIP_G1>::from_scalars(ctx, sb.begin(), sb.end()), bidx, rand_bases_g1(slice.H_n),
                                                 rand_bases_g1(slice.L_n), &slice);
    ctx.sync();
    *setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    Fr r = rnd(), s = rnd();
    for (int k = 0; k < steps; ++k) {
        if (k == steps - 1) zkhip_profile_enable(ctx.get(), 1);    // per-kernel HIP-event times of the last proof
        auto t1 = std::chrono::steady_clock::now();
        auto proof = all_gather ? r1cs_gg_ppzksnark_prover_hip<Curve>::process(dpk, primary, auxiliary, r, s, all_gather)
                                : r1cs_gg_ppzksnark_prover_hip<Curve>::process(dpk, primary, auxiliary, r, s);
        times[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        if (proof.g_A.is_zero()) return -2;
    }
    if (prof && prof_cap) zkhip_profile_dump(ctx.get(), prof, prof_cap);    // per-kernel HIP-event ms of the last proof
    return 0;
}

}    // namespace

extern "C" {

int zkhip_bench_groth16(int curve, size_t M, size_t n, uint64_t seed, int steps, const uint64_t *omega, const uint64_t *coset, double *times,
                        double *setup_ms, char *prof, size_t prof_cap) {
    try {
        if (curve == ZKHIP_BLS12_381) return groth16_bench_t<bls12_381>(0, 0, 1, nullptr, M, n, seed, steps, omega, coset, times, setup_ms, prof, prof_cap);
        return groth16_bench_t<alt_bn128_254>(0, 0, 1, nullptr, M, n, seed, steps, omega, coset, times, setup_ms, prof, prof_cap);
    } catch (const std::exception &e) {
        fprintf(stderr, "zkhip_bench_groth16: %s\n", e.what());
        return -1;
    }
}

/* one proof sharded over `world` processes / GPUs; every rank calls this with the same M, n, seed and steps */
int zkhip_bench_groth16_sharded(int device, size_t rank, size_t world, all_gather_fn all_gather, int curve, size_t M, size_t n, uint64_t seed, int steps,
                                const uint64_t *omega, const uint64_t *coset, double *times, double *setup_ms) {
    try {
        if (curve == ZKHIP_BLS12_381)
            return groth16_bench_t<bls12_381>(device, rank, world, all_gather, M, n, seed, steps, omega, coset, times, setup_ms, nullptr, 0);
        return groth16_bench_t<alt_bn128_254>(device, rank, world, all_gather, M, n, seed, steps, omega, coset, times, setup_ms, nullptr, 0);
    } catch (const std::exception &e) {
        fprintf(stderr, "zkhip_bench_groth16_sharded: %s\n", e.what());
        return -1;
    }
}

}    // extern "C"
