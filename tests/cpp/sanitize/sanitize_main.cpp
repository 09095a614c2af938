// TEST INFRASTRUCTURE: the host side of the header-only shim under AddressSanitizer + UBSan / ThreadSanitizer, against the
// stub backend (stub_backend.cpp: no GPU, nothing computed).  What runs here is everything the shim does on host threads --
// std::async conversions and uploads (backend.hpp), the prover's regrouped host products and two-thread assembly
// (r1cs_gg_ppzksnark.hpp), the generator's thread pool and the evaluation domains' chunked Lagrange evaluation, the chunked /
// lent uploads of the KZG and LPC schemes and the LPC leaf streaming -- and a mutation loop over proving_key_from_bytes
// (marshalling.hpp), which parses untrusted blobs: truncations at every framing boundary, oversized and zero counts,
// non-increasing / out-of-range B indices, random byte flips.  The parser must throw or succeed; the sanitizers decide the rest.
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <nil/crypto3/zk/hip/column_polynomial.hpp>
#include <nil/crypto3/zk/hip/lpc.hpp>
#include <nil/crypto3/zk/hip/marshalling.hpp>
#include <nil/crypto3/zk/hip/placeholder_arguments.hpp>
#include <nil/crypto3/zk/hip/placeholder_lookup.hpp>
#include <nil/crypto3/zk/hip/placeholder_quotient.hpp>
#include <nil/crypto3/zk/hip/r1cs_gg_ppzksnark_generator.hpp>

using namespace nil::crypto3::zk::hip;

// a curve whose scalar type is NOT declared to be canonical limbs in memory: every bulk transfer takes the converting,
// multi-threaded path (what a crypto3-algebra adapter gets)
struct converting_curve { };
namespace nil { namespace crypto3 { namespace zk { namespace hip {
template <>
struct curve_adapter<converting_curve> : curve_adapter<bls12_381> {
    static constexpr bool scalars_are_canonical_limbs = false;
};
}}}}

namespace {
int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

template <typename Curve>
r1cs_constraint_system<Curve> example_cs(std::size_t M, std::size_t n, std::vector<typename curve_adapter<Curve>::scalar_value_type> &full) {
    typedef typename curve_adapter<Curve>::scalar_value_type Fr;
    r1cs_constraint_system<Curve> cs;
    cs.primary_input_size = n;
    cs.auxiliary_input_size = 2 + M - n;
    Fr a(3), b(5);
    full = {a, b};
    for (std::size_t i = 0; i + 1 < M; ++i) {
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
    r1cs_constraint<Curve> c;
    Fr fin = Fr::zero();
    for (std::size_t i = 1; i < cs.num_variables(); ++i) {
        c.a.add_term(i, 1);
        c.b.add_term(i, 1);
        fin = fin + full[i - 1];
    }
    c.c.add_term(cs.num_variables(), 1);
    cs.add_constraint(c);
    full.push_back(fin * fin);
    return cs;
}

template <typename Curve>
void groth16_host_paths(std::size_t M) {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    const std::size_t n = 10;
    std::vector<Fr> full;
    auto cs = example_cs<Curve>(M, n, full);
    std::vector<Fr> primary(full.begin(), full.begin() + n), auxiliary(full.begin() + n, full.end());
    EXPECT(cs.is_satisfied(primary, auxiliary));
    context ctx(0);
    domain_params<Curve> dom {Fr(7), Fr(5)};    // the stub does not look at the roots
    /* the generator: parallel_chunks over the Lagrange basis and the query scalars, five from_scalars */
    auto key = r1cs_gg_ppzksnark_generator_hip<Curve>::deterministic_basic_process(ctx, cs, dom, Fr(1234567), Fr(11), Fr(13), Fr(17), Fr(19));
    typedef r1cs_gg_ppzksnark_prover_hip<Curve> prover;
    for (int rep = 0; rep < 2; ++rep) (void)prover::process(*key->device, primary, auxiliary);    // CSPRNG blinders, async host products
    /* a host-resident key uploaded whole and in two rank slices: threaded point conversions, process_partial + finish */
    r1cs_gg_ppzksnark_proving_key<Curve> pk;
    pk.constraint_system = key->host.constraint_system;
    const std::size_t N = cs.num_variables(), m = key->device->evaluation_domain.m;
    pk.A_query.assign(N + 1, A::g1_value_type::zero());
    pk.H_query.assign(m - 1, A::g1_value_type::zero());
    pk.L_query.assign(N - n, A::g1_value_type::zero());
    for (std::size_t i = 0; i <= N; i += 2) {
        pk.B_query.indices.push_back(i);
        pk.B_query.values.push_back({A::g2_value_type::zero(), A::g1_value_type::zero()});
    }
    pk.B_query.domain_size_ = N + 1;
    r1cs_gg_ppzksnark_proving_key_hip<Curve> whole(ctx, pk, dom);
    (void)prover::process(whole, primary, auxiliary, Fr(3), Fr(4));
    {
        /* two prover lanes over one key, proving at once on two threads (TSan: what the lanes share is read-only) */
        context ctx2(0);
        r1cs_gg_ppzksnark_proving_key_hip<Curve> lane(ctx2, whole);
        std::thread other([&]() {
            for (int k = 0; k < 2; ++k) (void)prover::process(lane, primary, auxiliary, Fr(3), Fr(4));
        });
        for (int k = 0; k < 2; ++k) (void)prover::process(whole, primary, auxiliary);
        other.join();
    }
    std::vector<std::uint64_t> gathered;
    for (std::size_t rank = 0; rank < 2; ++rank) {
        r1cs_gg_ppzksnark_proving_key_hip<Curve> part(ctx, pk, dom, rank, 2);
        auto mine = prover::process_partial(part, primary, auxiliary);
        gathered.insert(gathered.end(), mine.begin(), mine.end());
        if (rank == 1) (void)prover::finish(part, gathered, Fr(3), Fr(4));
    }
    {
        /* the device group: member keys over slices of every query, all members enqueued from this thread, the exchange, one assembly;
           then the reference's static signature over the default group (its per-thread key cache) */
        device_group grp({0, 0, 0});
        r1cs_gg_ppzksnark_proving_key_group_hip<Curve> gk(grp, pk, dom);
        for (int transport : {ZKHIP_GROUP_AUTO, ZKHIP_GROUP_STAGED}) {
            grp.set_transport(transport);
            (void)prover::process(gk, primary, auxiliary, Fr(3), Fr(4));
        }
        (void)prover::process(gk, primary, auxiliary);
        EXPECT(gk.members.size() == 3 && gk.members[2]->shard.rank == 2 && gk.members[0]->shard.A_n + gk.members[1]->shard.A_n + gk.members[2]->shard.A_n == N + 1);
    }
    /* malformed keys are refused, not read past their end */
    auto bad = pk;
    bad.H_query.pop_back();
    try {
        r1cs_gg_ppzksnark_proving_key_hip<Curve> k(ctx, bad, dom);
        EXPECT(!"a key with a short H query must be refused");
    } catch (const std::invalid_argument &) {
    }
    bad = pk;
    std::swap(bad.B_query.indices[1], bad.B_query.indices[2]);
    try {
        r1cs_gg_ppzksnark_proving_key_hip<Curve> k(ctx, bad, dom);
        EXPECT(!"non-increasing B indices must be refused");
    } catch (const std::invalid_argument &) {
    }
}

struct any_transcript {
    template <typename T>
    void operator()(const T &) { }
    fr_value<ZKHIP_BLS12_381> challenge() { return fr_value<ZKHIP_BLS12_381>(c++); }
    std::uint64_t c = 12345;
};
struct toy_tree {
    std::uint64_t r = 0;
    std::uint64_t root() const { return r; }
};
template <typename Fr>
struct vec_builder {
    toy_tree operator()(const std::vector<Fr> &leaves, std::size_t) const {
        toy_tree t;
        for (const auto &v : leaves) t.r ^= v.limbs[0];
        return t;
    }
};
template <typename Fr>
struct span_builder {
    toy_tree operator()(const Fr *leaves, std::size_t count, std::size_t) const {
        toy_tree t;
        for (std::size_t i = 0; i < count; ++i) t.r ^= leaves[i].limbs[0];
        return t;
    }
};
template <typename Fr>
struct stream_builder {
    toy_tree t;
    void begin(std::size_t, std::size_t) { t = toy_tree(); }
    void absorb(const Fr *leaves, std::size_t, std::size_t count) {
        for (std::size_t i = 0; i < count; ++i) t.r ^= leaves[i].limbs[0];
    }
    toy_tree finish() { return t; }
};

template <typename Curve>
void scheme_host_paths() {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    context ctx(0);
    auto root = [](std::size_t l) { return Fr((std::uint64_t)(l + 2)); };
    std::vector<polynomial_dfs<Curve>> polys(7);
    for (std::size_t p = 0; p < polys.size(); ++p) {
        polys[p].values.resize(p < 5 ? 256 : 512);
        for (std::size_t i = 0; i < polys[p].values.size(); ++i) polys[p].values[i] = Fr((std::uint64_t)(i * 7 + p));
    }
    {   /* KZG v2 and v1: copied, handed over, lent; chunked upload on the second context */
        std::vector<typename A::g1_value_type> ck(600, A::g1_value_type::zero());
        std::vector<typename A::g2_value_type> vk(4, A::g2_value_type::zero());
        kzg_params_hip<Curve> params(ctx, ck.begin(), ck.end(), vk.begin(), vk.end());
        kzg_commitment_scheme_v2_hip<Curve, any_transcript> v2(params, root);
        v2.upload_chunk = 2;
        v2.append_to_batch(0, polys[0]);
        v2.append_to_batch(0, std::vector<polynomial_dfs<Curve>>(polys.begin() + 1, polys.begin() + 3));
        std::vector<std::reference_wrapper<const polynomial_dfs<Curve>>> lent(polys.begin() + 3, polys.end());
        v2.append_to_batch(0, lent);
        auto moved = polys[0];
        v2.append_to_batch(0, std::move(moved));
        EXPECT(v2.commit(0).size() == 8);
        v2.append_eval_point(0, Fr(77));
        v2.append_eval_point(0, 2, Fr(78));
        any_transcript tr;
        (void)v2.proof_eval(tr);
        kzg_commitment_scheme_hip<Curve, any_transcript> v1(params, root);
        v1.append_to_batch(3, lent);
        (void)v1.commit(3);
        v1.append_eval_point(3, Fr(5));
        v1.append_eval_point(3, 1, Fr(6));
        (void)v1.proof_eval(tr);
        (void)v1.commit_g2({Fr(1), Fr(2), Fr(3)});
        (void)commit_one<Curve>(params, std::vector<Fr> {Fr(1), Fr(2)});
    }
    {   /* KZG over a device group: the key replicated, commit(batch) dealt over the members on a host thread each (TSan: the members share
           nothing but read-only inputs), host and resident columns mixed, more members than columns */
        std::vector<typename A::g1_value_type> ck(600, A::g1_value_type::zero());
        for (std::size_t world : {(std::size_t)2, (std::size_t)3, (std::size_t)9}) {
            device_group grp(std::vector<int>(world, 0));
            kzg_params_group_hip<Curve> gparams(grp, ck.begin(), ck.end());
            EXPECT(gparams.members.size() == world);
            kzg_commitment_scheme_v2_hip<Curve, any_transcript> v2(gparams, root);
            v2.group_commit_min = 1;    // the quotient commitments cut over the members too
            std::vector<std::reference_wrapper<const polynomial_dfs<Curve>>> lent(polys.begin(), polys.end());
            v2.append_to_batch(0, lent);
            device_polynomial_dfs<Curve> resident(grp[0], polys[1]);
            v2.append_to_batch(0, resident);
            EXPECT(v2.commit(0).size() == 8);
            v2.append_eval_point(0, Fr(77));
            any_transcript tr;
            (void)v2.proof_eval(tr);
            EXPECT(v2.group_multiexps() == 2);
        }
    }
    fri_params_hip<Curve> fp;
    fp.log_domain = 10;
    fp.step_list = {1, 2, 1};
    fp.root_of_unity = root;
    auto run_lpc = [&](auto builder) {
        lpc_commitment_scheme_hip<Curve, any_transcript, decltype(builder)> s(ctx, fp, builder);
        s.upload_chunk = 2;
        s.leaf_slice_elements = 700;    // several slices, rounded up to whole leaves
        std::vector<std::reference_wrapper<const polynomial_dfs<Curve>>> lent(polys.begin(), polys.end());
        s.append_to_batch(0, lent);
        (void)s.commit(0);
        s.append_to_batch(1, polys[5]);
        (void)s.commit(1);
        s.append_eval_point(0, Fr(9));
        s.append_eval_point(1, Fr(9));
        any_transcript tr;
        auto proof = s.proof_eval(tr);
        EXPECT(proof.fri_proof.fri_roots.size() == 3);
        (void)s.coefficients(0, 1);
        (void)s.fri_round_polynomial(0);
    };
    run_lpc(vec_builder<Fr>());
    run_lpc(span_builder<Fr>());
    run_lpc(stream_builder<Fr>());
    /* the same over a device group: one host thread per member with polynomials, the exchange, leaves by owner */
    auto run_lpc_group = [&](auto builder, std::size_t world) {
        device_group grp(std::vector<int>(world, 0));
        lpc_commitment_scheme_hip<Curve, any_transcript, decltype(builder)> s(grp, fp, builder);
        s.upload_chunk = 1;
        s.leaf_slice_elements = 700;
        std::vector<std::reference_wrapper<const polynomial_dfs<Curve>>> lent(polys.begin(), polys.end());
        s.append_to_batch(0, lent);
        (void)s.commit(0);
        s.append_to_batch(1, polys[5]);
        (void)s.commit(1);
        EXPECT(s.group_commits() == 2);
        s.append_eval_point(0, Fr(9));
        s.append_eval_point(1, Fr(9));
        any_transcript tr;
        auto proof = s.proof_eval(tr);
        EXPECT(proof.fri_proof.fri_roots.size() == 3);
        (void)s.coefficients(0, 1);
    };
    run_lpc_group(vec_builder<Fr>(), 2);
    run_lpc_group(span_builder<Fr>(), 3);
    run_lpc_group(stream_builder<Fr>(), 9);
}

/// the host logic of placeholder's arguments and quotient chain over the stub backend (device buffers are host memory the stub only sizes and
/// touches): buffer sizing, the grouping of gates and parts, slot indexing, the multi-part forms, argument checks
template <typename Curve>
void placeholder_host_paths() {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    typedef device_polynomial_dfs<Curve> dfs;
    context ctx(0);
    auto root = [](std::size_t l) { return Fr((std::uint64_t)(l + 2)); };
    const std::size_t n = 64, usable = 60;
    auto make = [&](std::size_t size, std::uint64_t seed) {
        polynomial_dfs<Curve> h;
        for (std::size_t i = 0; i < size; ++i) h.values.push_back(Fr(i * 3 + seed));
        return dfs(ctx, h, size - 1);
    };
    std::vector<dfs> cols, sid, ssig;
    for (std::uint64_t i = 0; i < 5; ++i) cols.push_back(make(n, i)), sid.push_back(make(n, 10 + i)), ssig.push_back(make(n, 20 + i));
    dfs q_last = make(n, 31), q_blind = make(n, 32), l0 = make(n, 33);
    typedef placeholder_permutation_hip<Curve> PA;
    typedef placeholder_lookup_hip<Curve> LA;
    typedef placeholder_quotient_hip<Curve> Q;
    auto one = PA::prove_eval(ctx, cols, sid, ssig, q_last, q_blind, l0, Fr(3), Fr(5), root);
    EXPECT(one.parts_dfs.empty() && one.F_dfs[1].size() >= 4 * n);
    auto multi = PA::prove_eval(ctx, cols, sid, ssig, q_last, q_blind, l0, Fr(3), Fr(5), root, 3, {Fr(7), Fr(8)}, usable);    // 5 factors in groups of 2: 3 parts
    EXPECT(multi.parts_dfs.size() == 2);
    bool threw = false;
    try {
        (void)PA::prove_eval(ctx, cols, sid, ssig, q_last, q_blind, l0, Fr(3), Fr(5), root, 3, {Fr(7)}, usable);    // one alpha short
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    EXPECT(threw);
    std::vector<dfs> l_in = {make(2 * n, 40), make(n, 41)}, l_val = {make(n, 42)}, sorted = {make(n, 43), make(n, 44), make(n, 45)};
    auto look = LA::prove_eval(ctx, l_in, l_val, sorted, q_last, q_blind, l0, Fr(3), Fr(5), {Fr(1), Fr(2)}, usable, root);
    EXPECT(look.parts_dfs.empty());
    auto look2 = LA::prove_eval(ctx, l_in, l_val, sorted, q_last, q_blind, l0, Fr(3), Fr(5), {Fr(1), Fr(2)}, usable, root, {1, 2}, {Fr(9)});
    EXPECT(look2.parts_dfs.size() == 1);
    threw = false;
    try {
        (void)LA::prove_eval(ctx, l_in, l_val, sorted, q_last, q_blind, l0, Fr(3), Fr(5), {Fr(1), Fr(2)}, usable, root, {1, 1}, {Fr(9)});    // parts do not cover
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    EXPECT(threw);
    /* gates sharing columns, with and without a budget that splits them into groups */
    gate_product_hip<Curve> g1, g2, g3;
    g1.factors = {&cols[0], &cols[1], &cols[2]};
    g1.rotations = {0, 1, 0};
    g1.coefficient = Fr(2);
    g2.factors = {&cols[0], &cols[1]};
    g2.rotations = {0, 0};
    g2.coefficient = Fr(3);
    g3.factors = {&cols[3], &cols[3], &cols[1]};
    g3.rotations = {0, -1, 1};
    g3.coefficient = Fr(4);
    dfs mask = LA::affine(q_last, &q_blind, Fr(1), Fr(1), Fr(1));
    dfs G = Q::gate_argument(ctx, {g1, g2, g3}, mask, 4 * n, root);
    dfs Gs = Q::gate_argument(ctx, {g1, g2, g3}, mask, 4 * n, root, 1);
    EXPECT(G.size() == 4 * n && Gs.size() == 4 * n && G.degree() == Gs.degree());
    std::vector<dfs> F = {one.F_dfs[0], one.F_dfs[1], one.F_dfs[2], look.F_dfs[0], look.F_dfs[1], look.F_dfs[2], look.F_dfs[3], G};
    std::vector<Fr> alphas(F.size(), Fr(6));
    auto T = Q::quotient_polynomial(ctx, F, alphas, n, root);    // the stub reports no remainder
    std::size_t largest = 0;
    for (const auto &f : F) largest = std::max(largest, f.size());
    EXPECT(T.size == largest - n);
    auto parts = Q::quotient_polynomial_split_dfs(ctx, T, n, T.size / n + 1, n, root);
    EXPECT(parts.size() == T.size / n + 1);
    evaluation_domain_hip<Curve> dom(ZKHIP_DOMAIN_BASIC_RADIX2, n, Fr(2));
    std::vector<std::vector<Fr>> columns(3, std::vector<Fr>(n, Fr(4)));
    EXPECT(column_range_polynomials<Curve>(ctx, columns, dom).size() == 3);
    /* round 5: extension caches (shared by copies, dropped by in-place changes), the factors-from-columns path, coefficient-form quotient parts */
    for (auto *v : {&sid, &ssig})
        for (auto &p : *v) p.enable_extension_cache();
    for (dfs *p : {&q_last, &q_blind, &l0}) p->enable_extension_cache();
    dfs copy_of_l0 = l0;
    const dfs e1 = l0.extension(4 * n, root), e2 = copy_of_l0.extension(4 * n, root);
    EXPECT(e1.data() == e2.data() && e1.size() == 4 * n);    // the copy hit the cache the original filled
    copy_of_l0 += q_last;                                     // an in-place change drops the (shared) cache's entries
    EXPECT(l0.extension(4 * n, root).data() != e1.data());
    auto cached = PA::prove_eval(ctx, cols, sid, ssig, q_last, q_blind, l0, Fr(3), Fr(5), root, 3, {Fr(7), Fr(8)}, usable);
    auto cached2 = PA::prove_eval(ctx, cols, sid, ssig, q_last, q_blind, l0, Fr(3), Fr(5), root, 3, {Fr(7), Fr(8)}, usable);
    EXPECT(cached.parts_dfs.size() == 2 && cached2.F_dfs[1].size() == cached.F_dfs[1].size());
    auto cparts = Q::quotient_polynomial_split_coefficients(ctx, T, n, T.size / n + 1, n);
    EXPECT(cparts.size() == T.size / n + 1 && cparts[0].size() == n);
}

/// the reference-shaped entry points of placeholder_arguments.hpp over the stub: the duck-typed reads, the hooks' lifetimes, the lookup prover class
namespace ref_like_ph {
    struct variable {
        std::size_t index;
        bool operator<(const variable &o) const { return index < o.index; }
    };
    struct table_description {
        std::size_t usable_rows_amount;
        std::size_t global_index(const variable &v) const { return v.index; }
    };
    struct lookup_table {
        std::size_t tag_index, columns_number;
        std::vector<std::vector<variable>> lookup_options;
    };
    struct constraint_system {
        std::vector<variable> _permuted;
        std::vector<lookup_table> _tables;
        const std::vector<variable> &permuted_columns() const { return _permuted; }
        const std::vector<lookup_table> &lookup_tables() const { return _tables; }
        std::vector<std::size_t> lookup_parts(std::size_t) const { return {2, 1}; }
    };
    template <typename Poly>
    struct preprocessed_data {
        std::vector<Poly> permutation_polynomials, identity_polynomials;
        Poly q_last, q_blind;
        struct {
            Poly lagrange_0;
            std::size_t max_quotient_chunks, permutation_parts;
            table_description desc;
        } common_data;
    };
    template <typename Poly>
    struct polynomial_table {
        std::vector<Poly> cols;
        const Poly &operator[](std::size_t i) const { return cols[i]; }
        const Poly &selector(std::size_t i) const { return cols[i]; }
        const Poly &constant(std::size_t i) const { return cols[i]; }
    };
    template <typename Fr>
    struct counting_transcript {
        std::size_t challenges = 0, absorbed = 0;
        Fr challenge() { return Fr((std::uint64_t)(100 + challenges++)); }
        template <typename T>
        void operator()(const T &) {
            ++absorbed;
        }
    };
    template <typename Curve>
    struct counting_scheme {
        typedef std::vector<int> commitment_type;
        std::map<std::size_t, std::size_t> appended;
        void append_to_batch(std::size_t batch, const device_polynomial_dfs<Curve> &) { ++appended[batch]; }
        commitment_type commit(std::size_t batch) { return commitment_type(appended[batch], 1); }
    };
}    // namespace ref_like_ph
template <typename Curve>
void placeholder_reference_entry_points() {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    typedef polynomial_dfs<Curve> poly;
    context ctx(0);
    set_default_context(&ctx);
    const std::size_t n = 32, usable = 28;
    auto make = [&](std::size_t size, std::uint64_t seed) {
        poly h;
        for (std::size_t i = 0; i < size; ++i) h.values.push_back(Fr(i * 5 + seed));
        return h;
    };
    ref_like_ph::preprocessed_data<poly> pd;
    ref_like_ph::polynomial_table<poly> table;
    ref_like_ph::constraint_system cs;
    for (std::size_t i = 0; i < 3; ++i) {
        table.cols.push_back(make(n, i));
        pd.identity_polynomials.push_back(make(n, 10 + i));
        pd.permutation_polynomials.push_back(make(n, 20 + i));
        cs._permuted.push_back({i});
    }
    pd.q_last = make(n, 31), pd.q_blind = make(n, 32), pd.common_data.lagrange_0 = make(n, 33);
    pd.common_data.max_quotient_chunks = 3;    // 3 factors in groups of 2: 2 parts
    pd.common_data.permutation_parts = 2;
    pd.common_data.desc.usable_rows_amount = usable;
    cs._tables.push_back({0, 2, {{{1}, {2}}}});    // one table, one option, two constant columns
    ref_like_ph::counting_transcript<Fr> transcript;
    ref_like_ph::counting_scheme<Curve> scheme;
    auto root = [](std::size_t l) { return Fr((std::uint64_t)(l + 2)); };
    auto perm = placeholder_permutation_argument_hip<Curve>::prove_eval(cs, pd, pd.common_data.desc, table, scheme, transcript, root);
    EXPECT(transcript.challenges == 3 && scheme.appended[PERMUTATION_BATCH] == 2 && perm.parts_dfs.size() == 1);
    std::vector<poly> inputs = {make(n, 50), make(2 * n, 51)};
    auto prover = make_placeholder_lookup_argument_prover<Curve>(cs, pd, table, scheme, transcript, [&](const Fr &) { return inputs; }, root);
    EXPECT(transcript.challenges == 4);
    auto look = prover.prove_eval();
    /* theta | beta, gamma, one part alpha | two F_3 alphas; three sorted vectors committed; V_L and one intermediate polynomial appended */
    EXPECT(transcript.challenges == 4 + 3 + 2 && transcript.absorbed == 1 && look.lookup_commitment.size() == 3 && scheme.appended[PERMUTATION_BATCH] == 4);
    EXPECT(prover.sorted_dfs.size() == 3 && prover.parts_dfs.size() == 1 && prover.V_L_dfs.size() == 1);
    set_default_context(nullptr);
}

template <typename Curve>
void bulk_transfers() {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    context ctx(0);
    const std::size_t count = ((std::size_t)1 << 18) * 2 + 1000;    // several staging slices per thread
    std::vector<Fr> v(count);
    for (std::size_t i = 0; i < count; ++i) v[i] = Fr((std::uint64_t)i);
    auto d = ctx.alloc(count * 32);
    upload_scalars<A>(ctx, d.get(), v.data(), count);
    std::vector<Fr> back;
    download_scalars<A>(ctx, d.get(), count, back);
    EXPECT(back.size() == count && back[12345] == v[12345] && back[count - 1] == v[count - 1]);
    std::vector<typename A::g1_value_type> pts(((std::size_t)1 << 14) + 77, A::g1_value_type::zero());
    device_bases<Curve, ZKHIP_G1> b(ctx, pts.begin(), pts.end());
    EXPECT(b.size() == pts.size());
}

// ---- the parser fuzz ---------------------------------------------------------------------------------------------------
struct blob_writer {
    std::vector<std::uint8_t> b;
    std::vector<std::size_t> count_fields;    // offsets of every u32 count / length / index field
    void u32(std::size_t v) {
        count_fields.push_back(b.size());
        for (int s = 24; s >= 0; s -= 8) b.push_back((std::uint8_t)(v >> s));
    }
    void bytes(std::size_t n, std::uint8_t fill) { b.insert(b.end(), n, fill); }
    void patch(std::size_t at, std::size_t v) {
        for (int i = 0; i < 4; ++i) b[at + i] = (std::uint8_t)(v >> (24 - 8 * i));
    }
};
/// a syntactically valid key blob for a tiny instance (the stub does not decode points: the framing is what is under test)
blob_writer valid_blob(std::size_t M, std::size_t n) {
    const std::size_t N = M + 2;
    int kind = 0;
    std::size_t m = 0;
    zkhip_domain_choice(ZKHIP_BLS12_381, M + n + 1, &kind, &m);
    blob_writer w;
    w.bytes(48, 0xc0), w.bytes(48, 0xc0), w.bytes(96, 0xc0), w.bytes(48, 0xc0), w.bytes(96, 0xc0);
    w.u32(N + 1);
    w.bytes((N + 1) * 48, 0xc0);
    {
        blob_writer body;
        const std::size_t cnt = N / 2;
        body.u32(cnt);
        for (std::size_t i = 0; i < cnt; ++i) body.u32(2 * i);
        body.bytes(cnt * (96 + 48), 0xc0);
        body.u32(N + 1);
        w.u32(body.b.size());
        const std::size_t base = w.b.size();
        for (std::size_t f : body.count_fields) w.count_fields.push_back(base + f);
        w.b.insert(w.b.end(), body.b.begin(), body.b.end());
    }
    w.u32(m - 1);
    w.bytes((m - 1) * 48, 0xc0);
    w.u32(N - n);
    w.bytes((N - n) * 48, 0xc0);
    w.u32(n), w.u32(N - n), w.u32(M);
    for (std::size_t i = 0; i < M; ++i) {
        blob_writer c;
        for (int k = 0; k < 3; ++k) {
            c.u32(1);
            c.u32(k == 2 ? N : (i % N) + 1);
            c.bytes(32, 1);
        }
        w.u32(c.b.size());
        const std::size_t base = w.b.size();
        for (std::size_t f : c.count_fields) w.count_fields.push_back(base + f);
        w.b.insert(w.b.end(), c.b.begin(), c.b.end());
    }
    return w;
}

void parser_fuzz() {
    typedef bls12_381 C;
    typedef curve_adapter<C>::scalar_value_type Fr;
    context ctx(0);
    domain_params<C> dom {Fr(7), Fr(5)};
    std::size_t accepted = 0, refused = 0;
    auto attempt = [&](const std::vector<std::uint8_t> &b) {
        try {
            auto key = proving_key_from_bytes<C>(ctx, b.data(), b.size(), dom);
            ++accepted;
            /* whatever was accepted must also be PROVABLE-WITH without touching memory it does not own */
            const auto &cs = key->host.constraint_system;
            if (cs.num_variables() < 4096) {
                std::vector<Fr> primary(cs.num_inputs(), Fr(1)), auxiliary(cs.num_variables() - cs.num_inputs(), Fr(2));
                (void)r1cs_gg_ppzksnark_prover_hip<C>::process(*key->device, primary, auxiliary, Fr(3), Fr(4));
            }
        } catch (const std::exception &) {
            ++refused;
        }
    };
    const blob_writer good = valid_blob(20, 3);
    attempt(good.b);
    EXPECT(accepted == 1);    // the unmutated blob parses
    /* truncations: every length around every framing boundary, and a stride through the rest */
    for (std::size_t f : good.count_fields)
        for (std::size_t cut : {f, f + 1, f + 3, f + 4, f + 5}) attempt(std::vector<std::uint8_t>(good.b.begin(), good.b.begin() + std::min(cut, good.b.size())));
    for (std::size_t cut = 0; cut < good.b.size(); cut += 37) attempt(std::vector<std::uint8_t>(good.b.begin(), good.b.begin() + cut));
    attempt({});
    /* every count / length / index field: zero, off by one, huge, all ones */
    for (std::size_t f : good.count_fields) {
        const std::size_t old = ((std::size_t)good.b[f] << 24) | ((std::size_t)good.b[f + 1] << 16) | ((std::size_t)good.b[f + 2] << 8) | good.b[f + 3];
        for (std::size_t v : {(std::size_t)0, old + 1, old ? old - 1 : 0, (std::size_t)0x7fffffff, (std::size_t)0xffffffff, (std::size_t)0x01000000}) {
            blob_writer w = good;
            w.patch(f, v);
            attempt(w.b);
        }
    }
    /* random byte flips */
    std::mt19937_64 rng(20260103);
    for (int it = 0; it < 600; ++it) {
        std::vector<std::uint8_t> b = good.b;
        const int flips = 1 + (int)(rng() % 4);
        for (int k = 0; k < flips; ++k) b[rng() % b.size()] ^= (std::uint8_t)(1u << (rng() % 8));
        attempt(b);
    }
    fprintf(stderr, "parser fuzz: %zu blobs accepted, %zu refused\n", accepted, refused);
    EXPECT(refused > 100);
}
}    // namespace

int main(int argc, char **argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    if (what == "all" || what == "threads") {
        groth16_host_paths<bls12_381>((1 << 14) + 5);    // past the thresholds where the shim spreads work over threads
        groth16_host_paths<converting_curve>(300);       // the staging path of the assignment, converting transfers
        groth16_host_paths<alt_bn128_254>(64);
        scheme_host_paths<bls12_381>();
        scheme_host_paths<converting_curve>();
        placeholder_host_paths<bls12_381>();
        placeholder_host_paths<converting_curve>();
        placeholder_reference_entry_points<bls12_381>();
        placeholder_reference_entry_points<converting_curve>();
        bulk_transfers<converting_curve>();
        bulk_transfers<bls12_381>();
    }
    if (what == "all" || what == "fuzz") parser_fuzz();
    fprintf(stderr, failures ? "sanitize_main: %d FAILED expectation(s)\n" : "sanitize_main: ok\n", failures);
    return failures ? 1 : 0;
}
