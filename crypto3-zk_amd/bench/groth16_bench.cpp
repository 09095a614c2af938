// Groth16 prover throughput driver for bench.py / tools/bench_groth16.py: runs the header-only shim
// (r1cs_gg_ppzksnark_prover_hip::process) on a synthetic instance and reports wall time per proof.
// Host compiler only; links libzkhip.so.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>
#include <memory>
#include <vector>

#include <nil/crypto3/zk/hip/r1cs_gg_ppzksnark_generator.hpp>

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

// Groth16 prover throughput on an instance of the reference's example family
// (generate_r1cs_example_with_field_input, r1cs_examples.hpp:77-140): constraint system, satisfying assignment and a
// VALID proving key generated on the device from a fixed trapdoor (r1cs_gg_ppzksnark_generator_hip).  After the timed
// proofs the last proof is held against the trapdoor identity (A = a G1, B = b G2, C = c G1, prover.hpp:141-153; the
// exponents come from O(nnz + m) host field arithmetic, the three scalar multiplications run on the host), and every timed proof
// must equal it (r and s are fixed across the steps): *verified.
// times[]: wall ms per proof.
// rank / world > 1: ONE proof sharded over `world` GPUs (one process each): this rank generates and holds only its
// slice of every query, runs the replicated witness map and its five partial MSMs, and `all_gather` (supplied by the
// caller: RCCL through torch.distributed in bench.py) exchanges the 864-byte partial sums.
typedef void (*all_gather_fn)(const uint64_t *mine, size_t words, uint64_t *all);
// the exchange on DEVICE buffers the caller owns (d_mine: this rank's 108 u64, d_all: world x 108 u64): returns when d_all is complete
typedef void (*all_gather_dev_fn)();
all_gather_dev_fn g_gather_dev = nullptr;
void *g_d_mine = nullptr, *g_d_all = nullptr;
// world > 1 without any exchange: time this rank's share only (process_partial) -- the per-rank emulation of tools/shard_emulation.py
bool g_partial_only = false;
// called between key generation and the timed proofs (tools/groth16_two_provers.py lines its threads up there)
typedef void (*after_setup_fn)();
after_setup_fn g_after_setup = nullptr;
int g_lanes = 1;
double g_lanes_info[4] = {0, 0, 0, 0};
// the evaluation domain the next runs name (kind < 0: make_evaluation_domain's choice, the default) and what the last run used
int g_dom_kind = -1;
size_t g_dom_m = 0;
uint64_t g_last_info[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // domain kind, domain points, A / B / H / L query sizes of this rank, N, n
double g_last_instance_ms = 0;    // building the synthetic constraint system + assignment of the last run (not part of *setup_ms)

/// the synthetic instance of both bench entry points: the reference's generate_r1cs_example_with_field_input family
/// (r1cs_examples.hpp:77-140), M constraints, n public inputs, N = M + 2 variables; returns the generator state for (r, s)
template <typename Curve>
SplitMix build_instance(size_t M, size_t n, uint64_t seed, r1cs_constraint_system<Curve> &cs, std::vector<typename curve_adapter<Curve>::scalar_value_type> &full) {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    SplitMix rng {seed};
    auto rnd = [&]() {    // < 2^252 < r for both curves: canonical
        uint64_t w[4] = {rng.next(), rng.next(), rng.next(), rng.next() & 0x0fffffffffffffffULL};
        return A::scalar_from_limbs(w);
    };
    cs.primary_input_size = n;
    cs.auxiliary_input_size = 2 + M - n;
    Fr a = rnd(), b = rnd();
    full.push_back(a);
    full.push_back(b);
    cs.constraints.reserve(M);
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
    return rng;
}

template <typename Curve>
int groth16_bench_t(int device, size_t rank, size_t world, all_gather_fn all_gather, size_t M, size_t n, uint64_t seed, int steps, const uint64_t *omega,
                    const uint64_t *coset, double *times, double *setup_ms, int *verified, char *prof, size_t prof_cap) {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    auto t0 = std::chrono::steady_clock::now();
    r1cs_constraint_system<Curve> cs;
    std::vector<Fr> full;
    SplitMix rng = build_instance<Curve>(M, n, seed, cs, full);
    auto rnd = [&]() {    // < 2^252 < r for both curves: canonical
        uint64_t w[4] = {rng.next(), rng.next(), rng.next(), rng.next() & 0x0fffffffffffffffULL};
        return A::scalar_from_limbs(w);
    };
    std::vector<Fr> primary(full.begin(), full.begin() + n), auxiliary(full.begin() + n, full.end());

    context ctx(device);
    if (const char *e = getenv("ZKHIP_G16_SEGMENT_LOG")) ctx.set_option("msm_segment_log", atoi(e));    // experiments: buckets per lane of the shared reduction
    if (const char *e = getenv("ZKHIP_G16_MAIN_PRIORITY")) ctx.set_option("stream_priority", atoi(e));    // experiments
    domain_params<Curve> dom {A::scalar_from_limbs(omega), A::scalar_from_limbs(coset)};
    dom.kind = g_dom_kind;
    dom.m = g_dom_m;
    /* the toxic waste: fixed by the seed (identical on every rank of a sharded proof) */
    SplitMix key_rng {seed * 1000003 + 1};
    auto rnd_key = [&]() {
        uint64_t w[4] = {key_rng.next(), key_rng.next(), key_rng.next(), key_rng.next() & 0x0fffffffffffffffULL};
        return A::scalar_from_limbs(w);
    };
    const Fr t = rnd_key(), alpha = rnd_key(), beta = rnd_key(), gamma = rnd_key(), delta = rnd_key();
    ctx.sync();
    g_last_instance_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();    // the synthetic circuit + context
    t0 = std::chrono::steady_clock::now();    // from here: key generation proper (generator.hpp:240-377)
    auto key = r1cs_gg_ppzksnark_generator_hip<Curve>::deterministic_basic_process(ctx, cs, dom, t, alpha, beta, gamma, delta, rank, world);
    auto &dpk = *key->device;
    {
        const uint64_t info[8] = {(uint64_t)dpk.evaluation_domain.kind, dpk.evaluation_domain.m, dpk.A_query.size(), dpk.B_count, dpk.H_query.size(),
                                  dpk.shard.L_n, cs.num_variables(), cs.num_inputs()};
        memcpy(g_last_info, info, sizeof(info));
    }
    if (const char *e = getenv("ZKHIP_G16_OVERLAP")) dpk.overlap_g2 = atoi(e) != 0;    // experiments: G2 multiexp on the main stream
    if (const char *e = getenv("ZKHIP_G16_SIDE_PRIORITY")) dpk.side_stream_priority = atoi(e);
    if (const char *e = getenv("ZKHIP_G16_SHARE_SORTS")) dpk.share_sorts = atoi(e) != 0;    // experiments: 0 = every query sorts the assignment's digits itself
    if (const char *e = getenv("ZKHIP_G16_SKIP_G2")) dpk.experiment_skip_g2 = atoi(e) != 0;    // ceiling experiment: the proof is then WRONG
    ctx.sync();
    *setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (g_after_setup) g_after_setup();
    Fr r = rnd(), s = rnd();
    typedef r1cs_gg_ppzksnark_prover_hip<Curve> prover;
    typename prover::proof_type proof, first;
    int differing = 0;
    for (int k = 0; k < steps; ++k) {
        auto t1 = std::chrono::steady_clock::now();
        if (g_partial_only && world > 1) (void)prover::process_partial(dpk, primary, auxiliary);
        else if (g_gather_dev) proof = prover::process_device_gather(dpk, primary, auxiliary, r, s, g_d_mine, g_d_all, g_gather_dev);
        else proof = all_gather ? prover::process(dpk, primary, auxiliary, r, s, all_gather) : prover::process(dpk, primary, auxiliary, r, s);
        times[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        if (k == 0) first = proof;
        else if (!(proof.g_A == first.g_A && proof.g_B == first.g_B && proof.g_C == first.g_C)) ++differing;    // same (r, s): every proof is THE proof
    }
    /* Per-kernel durations for the roofline objects: ONE more, untimed proof with the G2 multiexp on the MAIN stream.  HIP events
       around the launches of a single in-order stream bracket exactly each kernel's execution; with the G2 multiexp on its own
       stream (the timed arrangement) an event pair also counts the time a launch queues behind the other stream's workgroups
       (msm_bucket_large 2.98 ms for empty launches), which distorted every share computed from it. */
    if (prof && prof_cap && !(g_partial_only && world > 1) && !g_gather_dev && !all_gather) {
        const bool was = dpk.overlap_g2;
        auto side = std::move(dpk.side);    // without its second context the key's G2 multiexp runs on the main stream (process() looks at `side`)
        dpk.overlap_g2 = false;
        (void)prover::process(dpk, primary, auxiliary, r, s);    // settle the serial arrangement, unprofiled
        zkhip_profile_enable(ctx.get(), 1);
        auto pv = prover::process(dpk, primary, auxiliary, r, s);
        zkhip_profile_dump(ctx.get(), prof, prof_cap);
        zkhip_profile_enable(ctx.get(), 0);
        dpk.overlap_g2 = was;
        dpk.side = std::move(side);
        if (!(pv.g_A == first.g_A && pv.g_B == first.g_B && pv.g_C == first.g_C)) ++differing;
    }
    /* the throughput arrangement: g_lanes provers at once over the SAME resident key (lane keys alias its queries), one host thread each */
    g_lanes_info[0] = g_lanes_info[1] = g_lanes_info[2] = g_lanes_info[3] = 0;
    if (g_lanes > 1 && world == 1 && !g_gather_dev) {
        std::vector<std::unique_ptr<context>> lane_ctx;
        std::vector<std::unique_ptr<typename prover::proving_key_type>> lane_key;
        for (int l = 1; l < g_lanes; ++l) {
            lane_ctx.emplace_back(new context(device));
            lane_key.emplace_back(new typename prover::proving_key_type(*lane_ctx.back(), dpk));
        }
        std::atomic<int> lane_differing {0};
        auto run = [&](const typename prover::proving_key_type &key, int count) {
            for (int k = 0; k < count; ++k) {
                auto pv = prover::process(key, primary, auxiliary, r, s);
                if (!(pv.g_A == first.g_A && pv.g_B == first.g_B && pv.g_C == first.g_C)) ++lane_differing;
            }
        };
        for (auto &k : lane_key) run(*k, 1);    // every lane allocates its work buffers outside the timed region
        const auto w0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (auto &k : lane_key) th.emplace_back([&, kp = k.get()]() { run(*kp, steps); });
        run(dpk, steps);
        for (auto &t : th) t.join();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        differing += lane_differing;
        g_lanes_info[0] = g_lanes;
        g_lanes_info[1] = (double)g_lanes * steps / wall;    // proofs per second, all lanes
        g_lanes_info[2] = wall / steps * 1e3;                // ms per proof seen by one lane
        g_lanes_info[3] = lane_differing == 0 ? 1 : 0;
    }
    if (getenv("ZKHIP_G16_PHASES"))
        fprintf(stderr, "proof phases (host ms): stage+launch %.3f | host products %.3f | wait %.3f | assemble %.3f\n", dpk.last_phase_ms[0],
                dpk.last_phase_ms[1], dpk.last_phase_ms[2], dpk.last_phase_ms[3]);
    /* the check, outside the timed region: the proof must be the one the trapdoor dictates */
    if (verified && !(g_partial_only && world > 1)) {
        const auto e = groth16_proof_exponents<Curve>(key->host.constraint_system, dom, primary, auxiliary, t, alpha, beta, delta, r, s);
        std::vector<Fr> one = {Fr::one()};
        const auto g1 = device_bases<Curve, ZKHIP_G1>::from_scalars(ctx, one.begin(), one.end()).at(0);    // the standard generators
        const auto g2 = device_bases<Curve, ZKHIP_G2>::from_scalars(ctx, one.begin(), one.end()).at(0);
        *verified = (differing == 0 && proof.g_A == e[0] * g1 && proof.g_B == e[1] * g2 && proof.g_C == e[2] * g1) ? 1 : 0;
    }
    return 0;
}

// ONE proof over a DEVICE GROUP (r1cs_gg_ppzksnark_proving_key_group_hip): `n_dev` contexts in THIS process, one host thread, the exchange
// inside the library -- BASELINE cfg 4's arrangement as a C++ caller of the drop-in class reaches it.  The key is generated slice by
// slice on the members' GPUs; every timed proof must equal the first (same r, s) and the proof the trapdoor dictates.
// info[0..3]: host ms of the last proof's phases (launches | host products | exchange + wait | assembly), info[4]: the transport used.
template <typename Curve>
int groth16_group_bench_t(const int *devices, int n_dev, int transport, size_t M, size_t n, uint64_t seed, int steps, const uint64_t *omega, const uint64_t *coset,
                          double *times, double *setup_ms, int *verified, double *info) {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    r1cs_constraint_system<Curve> cs;
    std::vector<Fr> full;
    SplitMix rng = build_instance<Curve>(M, n, seed, cs, full);
    auto rnd = [&]() {
        uint64_t w[4] = {rng.next(), rng.next(), rng.next(), rng.next() & 0x0fffffffffffffffULL};
        return A::scalar_from_limbs(w);
    };
    std::vector<Fr> primary(full.begin(), full.begin() + n), auxiliary(full.begin() + n, full.end());
    device_group grp(std::vector<int>(devices, devices + n_dev));
    grp.set_transport(transport);
    domain_params<Curve> dom {A::scalar_from_limbs(omega), A::scalar_from_limbs(coset)};
    dom.kind = g_dom_kind;
    dom.m = g_dom_m;
    SplitMix key_rng {seed * 1000003 + 1};
    auto rnd_key = [&]() {
        uint64_t w[4] = {key_rng.next(), key_rng.next(), key_rng.next(), key_rng.next() & 0x0fffffffffffffffULL};
        return A::scalar_from_limbs(w);
    };
    const Fr t = rnd_key(), alpha = rnd_key(), beta = rnd_key(), gamma = rnd_key(), delta = rnd_key();
    auto t0 = std::chrono::steady_clock::now();
    auto key = r1cs_gg_ppzksnark_generator_hip<Curve>::deterministic_basic_process(grp, cs, dom, t, alpha, beta, gamma, delta);
    grp.sync();
    *setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const Fr r = rnd(), s = rnd();
    typedef r1cs_gg_ppzksnark_prover_hip<Curve> prover;
    typename prover::proof_type proof, first;
    int differing = 0;
    (void)prover::process(*key->device, primary, auxiliary, r, s);    // work buffers, side streams, the communicator: outside the timed proofs
    for (int k = 0; k < steps; ++k) {
        auto t1 = std::chrono::steady_clock::now();
        proof = prover::process(*key->device, primary, auxiliary, r, s);
        times[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        if (k == 0) first = proof;
        else if (!(proof.g_A == first.g_A && proof.g_B == first.g_B && proof.g_C == first.g_C)) ++differing;
    }
    for (int i = 0; i < 4; ++i) info[i] = key->device->last_phase_ms[i];
    info[4] = grp.transport();
    if (verified) {
        const auto e = groth16_proof_exponents<Curve>(key->parts[0]->host.constraint_system, dom, primary, auxiliary, t, alpha, beta, delta, r, s);
        std::vector<Fr> one = {Fr::one()};
        const auto g1 = device_bases<Curve, ZKHIP_G1>::from_scalars(grp[0], one.begin(), one.end()).at(0);
        const auto g2 = device_bases<Curve, ZKHIP_G2>::from_scalars(grp[0], one.begin(), one.end()).at(0);
        *verified = (differing == 0 && proof.g_A == e[0] * g1 && proof.g_B == e[1] * g2 && proof.g_C == e[2] * g1) ? 1 : 0;
    }
    return 0;
}

}    // namespace

extern "C" {

/* one proof over a device group of THIS process (see groth16_group_bench_t); info: 5 doubles */
int zkhip_bench_groth16_group(const int *devices, int n_dev, int transport, int curve, size_t M, size_t n, uint64_t seed, int steps, const uint64_t *omega,
                              const uint64_t *coset, double *times, double *setup_ms, int *verified, double *info) {
    try {
        if (curve == ZKHIP_BLS12_381) return groth16_group_bench_t<bls12_381>(devices, n_dev, transport, M, n, seed, steps, omega, coset, times, setup_ms, verified, info);
        return groth16_group_bench_t<alt_bn128_254>(devices, n_dev, transport, M, n, seed, steps, omega, coset, times, setup_ms, verified, info);
    } catch (const std::exception &e) {
        fprintf(stderr, "zkhip_bench_groth16_group: %s\n", e.what());
        return -1;
    }
}

void zkhip_bench_set_after_setup(void (*fn)()) { g_after_setup = fn; }
/* lanes > 1: after the one-at-a-time proofs, zkhip_bench_groth16 runs `lanes` provers at once over the same key (steps proofs each);
   zkhip_bench_last_lanes: {lanes, proofs per second over all lanes, wall ms per proof of one lane, 1 if every proof equalled the verified one} */
void zkhip_bench_set_lanes(int lanes) { g_lanes = lanes < 1 ? 1 : lanes; }
void zkhip_bench_last_lanes(double *out4) { memcpy(out4, g_lanes_info, sizeof(g_lanes_info)); }
/* kind < 0: make_evaluation_domain(M + n + 1)'s choice (r1cs_to_qap.hpp:229-230); 0 with m = 2^k: the basic domain of that size */
void zkhip_bench_set_domain(int kind, size_t m) {
    g_dom_kind = kind;
    g_dom_m = m;
}
void zkhip_bench_last_info(uint64_t *out) { memcpy(out, g_last_info, sizeof(g_last_info)); }
double zkhip_bench_last_instance_ms() { return g_last_instance_ms; }
/* the sharded proof's exchange on device buffers (see all_gather_dev_fn); fn == NULL: back to the host-buffer callback */
void zkhip_bench_set_device_gather(all_gather_dev_fn fn, void *d_mine, void *d_all) {
    g_gather_dev = fn;
    g_d_mine = d_mine;
    g_d_all = d_all;
}
/* on: zkhip_bench_groth16_sharded times process_partial only (no exchange, no assembly, nothing verified) */
void zkhip_bench_set_partial_only(int on) { g_partial_only = on != 0; }

int zkhip_bench_groth16(int device, int curve, size_t M, size_t n, uint64_t seed, int steps, const uint64_t *omega, const uint64_t *coset, double *times,
                        double *setup_ms, int *verified, char *prof, size_t prof_cap) {
    try {
        if (curve == ZKHIP_BLS12_381)
            return groth16_bench_t<bls12_381>(device, 0, 1, nullptr, M, n, seed, steps, omega, coset, times, setup_ms, verified, prof, prof_cap);
        return groth16_bench_t<alt_bn128_254>(device, 0, 1, nullptr, M, n, seed, steps, omega, coset, times, setup_ms, verified, prof, prof_cap);
    } catch (const std::exception &e) {
        fprintf(stderr, "zkhip_bench_groth16: %s\n", e.what());
        return -1;
    }
}

/* one proof sharded over `world` processes / GPUs; every rank calls this with the same M, n, seed and steps */
int zkhip_bench_groth16_sharded(int device, size_t rank, size_t world, all_gather_fn all_gather, int curve, size_t M, size_t n, uint64_t seed, int steps,
                                const uint64_t *omega, const uint64_t *coset, double *times, double *setup_ms, int *verified) {
    try {
        if (curve == ZKHIP_BLS12_381)
            return groth16_bench_t<bls12_381>(device, rank, world, all_gather, M, n, seed, steps, omega, coset, times, setup_ms, verified, nullptr, 0);
        return groth16_bench_t<alt_bn128_254>(device, rank, world, all_gather, M, n, seed, steps, omega, coset, times, setup_ms, verified, nullptr, 0);
    } catch (const std::exception &e) {
        fprintf(stderr, "zkhip_bench_groth16_sharded: %s\n", e.what());
        return -1;
    }
}

}    // extern "C"
