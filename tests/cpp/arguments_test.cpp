// Test harness, second translation unit: placeholder's permutation and lookup arguments through the REFERENCE-SHAPED entry points of
// hip/placeholder_arguments.hpp, driven the way the unchanged reference code drives them -- a constraint system, preprocessed data, a
// table description, a polynomial table, a commitment scheme and a transcript, all declared HERE like the reference's (distinct from
// anything the shim ships; polynomials with private storage) --, with a transcript and a scheme that RECORD the order of every
// challenge draw, append_to_batch, commit and absorb.  tests/test_gpu_shim.py holds the recorded order against the oracle's replay of
//   permutation_argument.hpp:95-97, 139, 181-183, 200      and      lookup_argument.hpp:150, 192-206, 213, 267, 282-283,
// and the polynomials against the oracle's; the explicit-challenge overloads must give the same bits.
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

#include <nil/crypto3/zk/hip/kzg_v2.hpp>
#include <nil/crypto3/zk/hip/placeholder_arguments.hpp>

using namespace nil::crypto3::zk::hip;

/// a curve that names its scalar field, as crypto3-algebra's curves do: the transcript is then asked through challenge<FieldType>()
template <int Curve>
struct named_field_curve {
    struct scalar_field_type {
        typedef typename curve_adapter<native_curve<Curve>>::scalar_value_type value_type;
    };
};
namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {
template <int Curve>
struct curve_adapter<named_field_curve<Curve>> : curve_adapter<native_curve<Curve>> { };
}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

namespace {

enum : uint64_t { EV_CHALLENGE = 1, EV_ABSORB = 3, EV_APPEND = 100, EV_COMMIT = 200 };

namespace ref_like {
    /// math::polynomial_dfs as the shim may read it: size(), operator[], degree(); storage private
    template <typename FieldValueType>
    class polynomial_dfs {
        std::vector<FieldValueType> val;
        std::size_t _d = 0;

    public:
        typedef FieldValueType value_type;
        polynomial_dfs() = default;
        polynomial_dfs(std::size_t d, std::vector<FieldValueType> v) : val(std::move(v)), _d(d) { }
        std::size_t size() const { return val.size(); }
        std::size_t degree() const { return _d; }
        const FieldValueType &operator[](std::size_t i) const { return val[i]; }
    };

    /// plonk_variable: a column reference
    struct variable {
        std::size_t index;
        enum column_type { witness, public_input, constant, selector } type;
        bool operator<(const variable &o) const { return type != o.type ? type < o.type : index < o.index; }
    };
    /// plonk_table_description::global_index (table_description.hpp): witnesses, public inputs, constants, selectors in that order
    struct table_description {
        std::size_t witness_columns, public_input_columns, constant_columns, selector_columns, usable_rows_amount, rows_amount;
        std::size_t global_index(const variable &a) const {
            switch (a.type) {
                case variable::witness: return a.index;
                case variable::public_input: return witness_columns + a.index;
                case variable::constant: return witness_columns + public_input_columns + a.index;
                default: return witness_columns + public_input_columns + constant_columns + a.index;
            }
        }
    };
    /// plonk_lookup_table (lookup_table.hpp): tag_index, columns_number, lookup_options[o][i] = a constant column
    struct lookup_table {
        std::size_t tag_index, columns_number;
        std::vector<std::vector<variable>> lookup_options;
    };
    struct constraint_system {
        std::set<variable> _permuted;
        std::vector<lookup_table> _tables;
        std::vector<std::size_t> _parts;
        const std::set<variable> &permuted_columns() const { return _permuted; }
        const std::vector<lookup_table> &lookup_tables() const { return _tables; }
        std::vector<std::size_t> lookup_parts(std::size_t /*max_quotient_chunks*/) const { return _parts; }    // :56-107, the constraint system's arithmetic
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
    /// plonk_polynomial_dfs_table: operator[](global index), selector(i), constant(i)
    template <typename Poly>
    struct polynomial_table {
        std::vector<Poly> _witnesses, _constants, _selectors;
        const Poly &operator[](std::size_t g) const {
            if (g < _witnesses.size()) return _witnesses[g];
            g -= _witnesses.size();
            if (g < _constants.size()) return _constants[g];
            return _selectors[g - _constants.size()];
        }
        const Poly &constant(std::size_t i) const { return _constants[i]; }
        const Poly &selector(std::size_t i) const { return _selectors[i]; }
    };
}    // namespace ref_like

/// fiat_shamir_heuristic_sequential's surface: operator()(anything), challenge<FieldType>() -- and the shim's own .challenge() --, recording
template <typename Fr>
struct recording_transcript {
    std::vector<uint64_t> *events;
    std::vector<Fr> challenges;
    std::size_t next = 0;
    template <typename T>
    void operator()(const T &) {
        events->push_back(EV_ABSORB);
    }
    template <typename FieldType>
    typename FieldType::value_type challenge() {
        events->push_back(EV_CHALLENGE);
        return challenges.at(next++);
    }
    Fr challenge() {
        events->push_back(EV_CHALLENGE);
        return challenges.at(next++);
    }
};

/// the commitment scheme placeholder holds, recording; the work goes to the real kzg_commitment_scheme_v2_hip
template <typename Scheme>
struct recording_scheme {
    typedef typename Scheme::commitment_type commitment_type;
    Scheme &inner;
    std::vector<uint64_t> *events;
    template <typename Poly>
    void append_to_batch(std::size_t batch, const Poly &p) {
        events->push_back(EV_APPEND + batch);
        inner.append_to_batch(batch, p);
    }
    commitment_type commit(std::size_t batch) {
        events->push_back(EV_COMMIT + batch);
        return inner.commit(batch);
    }
};

template <typename Curve>
bool same_bits(const context &ctx, const device_polynomial_dfs<Curve> &a, const device_polynomial_dfs<Curve> &b) {
    if (a.size() != b.size() || a.degree() != b.degree()) return false;
    std::vector<uint64_t> x(4 * a.size()), y(4 * a.size());
    ctx.d2h(x.data(), a.data(), a.size() * 32);
    ctx.d2h(y.data(), b.data(), b.size() * 32);
    return x == y;
}

/// evals (2^log_n each): k permuted witness columns | k S_id | k S_sigma | q_last | q_blind | lagrange_0 | the lookup tag selector |
///   `options * columns` constant columns (option-major) | k_in lookup inputs (what prepare_lookup_input returns for the scripted theta;
///   input i on 2^in_logs[i] points).
/// challenges: beta_p, gamma_p, (permutation_parts - 1) alphas | theta, beta_l, gamma_l, (lookup parts - 1) alphas, (k_in + options - 1) alphas.
/// out_events: the recorded order, permutation argument first, a 0 between the two.
/// out_perm: V_P (n) | parts (n each) | 3 F as coefficients in slots of 8 n;  out_look: V_L (n) | parts (n each) | sorted (n each) | 4 F in slots of 16 n.
template <typename Curve>
int arguments_transcript_t(const uint64_t *srs, size_t n_srs, const uint64_t *evals, size_t k, size_t log_n, size_t usable_rows, size_t max_quotient_chunks,
                           size_t permutation_parts, size_t options, size_t columns, size_t k_in, const uint64_t *in_logs, const uint64_t *lookup_part_sizes,
                           size_t n_lookup_parts, const uint64_t *challenges, size_t n_challenges, uint64_t *out_events, size_t cap_events, uint64_t *n_events,
                           uint64_t *out_perm, uint64_t *perm_sizes, uint64_t *out_look, uint64_t *look_sizes) {
    typedef curve_adapter<Curve> A;
    typedef typename A::scalar_value_type Fr;
    typedef typename A::g1_value_type G1;
    typedef ref_like::polynomial_dfs<Fr> poly;
    typedef device_polynomial_dfs<Curve> dfs;
    const size_t n = (size_t)1 << log_n, L1 = 2 * A::g1_coord_limbs;
    context ctx(0);
    set_default_context(&ctx);
    struct reset_default {
        ~reset_default() { set_default_context(nullptr); }
    } reset;
    const uint64_t *at = evals;
    auto take = [&](size_t size) {
        std::vector<Fr> v;
        for (size_t i = 0; i < size; ++i) v.push_back(A::scalar_from_limbs(at + 4 * i));
        at += 4 * size;
        return poly(size - 1, std::move(v));
    };
    ref_like::constraint_system cs;
    ref_like::preprocessed_data<poly> pd;
    ref_like::polynomial_table<poly> table;
    for (size_t i = 0; i < k; ++i) table._witnesses.push_back(take(n));
    for (size_t i = 0; i < k; ++i) pd.identity_polynomials.push_back(take(n));
    for (size_t i = 0; i < k; ++i) pd.permutation_polynomials.push_back(take(n));
    pd.q_last = take(n);
    pd.q_blind = take(n);
    pd.common_data.lagrange_0 = take(n);
    pd.common_data.max_quotient_chunks = max_quotient_chunks;
    pd.common_data.permutation_parts = permutation_parts;
    pd.common_data.desc = ref_like::table_description {k, 0, options * columns, 1, usable_rows, n};
    table._selectors.push_back(take(n));
    for (size_t i = 0; i < options * columns; ++i) table._constants.push_back(take(n));
    std::vector<poly> prepared_inputs;
    for (size_t i = 0; i < k_in; ++i) prepared_inputs.push_back(take((size_t)1 << in_logs[i]));
    for (size_t i = 0; i < k; ++i) cs._permuted.insert(ref_like::variable {i, ref_like::variable::witness});
    ref_like::lookup_table lt {0, columns, {}};
    for (size_t o = 0; o < options; ++o) {
        lt.lookup_options.emplace_back();
        for (size_t c = 0; c < columns; ++c) lt.lookup_options.back().push_back(ref_like::variable {o * columns + c, ref_like::variable::constant});
    }
    cs._tables.push_back(lt);
    cs._parts.assign(lookup_part_sizes, lookup_part_sizes + n_lookup_parts);

    std::vector<G1> ck;
    for (size_t i = 0; i < n_srs; ++i) ck.push_back(G1::from_affine(srs + i * L1));
    kzg_params_hip<Curve> params(ctx, ck.begin(), ck.end());
    typedef recording_transcript<Fr> transcript_type;
    typedef kzg_commitment_scheme_v2_hip<Curve, transcript_type> scheme_type;
    scheme_type inner(params, detail::field_roots<Curve>());
    std::vector<uint64_t> events;
    recording_scheme<scheme_type> scheme {inner, &events};
    transcript_type transcript {&events, {}, 0};
    for (size_t i = 0; i < n_challenges; ++i) transcript.challenges.push_back(A::scalar_from_limbs(challenges + 4 * i));
    auto ch = [&](size_t i) { return transcript.challenges.at(i); };
    const auto root = detail::field_roots<Curve>();

    /* ---- the permutation argument as prover.hpp:170-177 calls it */
    auto perm = placeholder_permutation_argument_hip<Curve>::prove_eval(cs, pd, pd.common_data.desc, table, scheme, transcript);
    const size_t perm_challenges = transcript.next;
    if (perm_challenges != 2 + permutation_parts - 1) return -301;
    events.push_back(0);
    {   // the explicit-challenge overload over device polynomials: the same bits
        std::vector<dfs> cols, sid, ssig;
        for (size_t i = 0; i < k; ++i) {
            cols.emplace_back(ctx, table._witnesses[i], n - 1);
            sid.emplace_back(ctx, pd.identity_polynomials[i], n - 1);
            ssig.emplace_back(ctx, pd.permutation_polynomials[i], n - 1);
        }
        std::vector<Fr> alphas;
        for (size_t i = 0; i + 1 < permutation_parts; ++i) alphas.push_back(ch(2 + i));
        auto ex = placeholder_permutation_hip<Curve>::prove_eval(ctx, cols, sid, ssig, dfs(ctx, pd.q_last, n - 1), dfs(ctx, pd.q_blind, n - 1),
                                                                dfs(ctx, pd.common_data.lagrange_0, n - 1), ch(0), ch(1), root, max_quotient_chunks, alphas, usable_rows);
        if (!same_bits(ctx, ex.permutation_polynomial_dfs, perm.permutation_polynomial_dfs) || ex.parts_dfs.size() != perm.parts_dfs.size()) return -302;
        for (size_t i = 0; i < ex.parts_dfs.size(); ++i)
            if (!same_bits(ctx, ex.parts_dfs[i], perm.parts_dfs[i])) return -303;
        for (int f = 0; f < 3; ++f)
            if (!same_bits(ctx, ex.F_dfs[f], perm.F_dfs[f])) return -304;
    }
    uint64_t *o = out_perm;
    ctx.d2h(o, perm.permutation_polynomial_dfs.data(), n * 32);
    o += 4 * n;
    for (auto &p : perm.parts_dfs) {
        ctx.d2h(o, p.data(), n * 32);
        o += 4 * n;
    }
    for (int f = 0; f < 3; ++f) {
        const size_t sz = perm.F_dfs[f].size();
        if (sz > 8 * n) return -305;
        perm_sizes[f] = sz;
        auto c = perm.F_dfs[f].coefficients(root);
        ctx.d2h(o + 4 * (size_t)f * 8 * n, c.get(), sz * 32);
    }

    /* ---- the lookup argument as prover.hpp:192-205 calls it: the class, then prove_eval() */
    size_t hook_calls = 0;
    const Fr theta_scripted = ch(perm_challenges);
    auto prepare_lookup_input = [&](const Fr &theta) {
        ++hook_calls;
        if (!(theta == theta_scripted)) throw std::runtime_error("prepare_lookup_input: theta is not the constructor's challenge");
        return prepared_inputs;
    };
    auto prover = make_placeholder_lookup_argument_prover<Curve>(cs, pd, table, scheme, transcript, prepare_lookup_input);
    if (transcript.next != perm_challenges + 1 || !(prover.theta == theta_scripted)) return -311;    // theta is drawn by the constructor (:150)
    auto look = prover.prove_eval();
    if (hook_calls != 1 || look.lookup_commitment.size() != k_in + options) return -312;
    const size_t total = k_in + options;
    if (transcript.next != perm_challenges + 3 + (n_lookup_parts - 1) + (total - 1)) return -313;
    {   // the explicit-challenge overload from the sorted vectors on, fed with this run's own intermediate vectors: the same bits
        size_t c0 = perm_challenges + 1;
        std::vector<Fr> part_alphas, alphas;
        for (size_t i = 0; i + 1 < n_lookup_parts; ++i) part_alphas.push_back(ch(c0 + 2 + i));
        for (size_t i = 0; i + 1 < total; ++i) alphas.push_back(ch(c0 + 2 + (n_lookup_parts - 1) + i));
        dfs q_last(ctx, pd.q_last, n - 1), q_blind(ctx, pd.q_blind, n - 1), l0(ctx, pd.common_data.lagrange_0, n - 1);
        dfs mask = placeholder_lookup_hip<Curve>::affine(q_last, &q_blind, Fr::zero() - Fr::one(), Fr::zero() - Fr::one(), Fr::one());
        std::vector<dfs> value = prover.prepare_lookup_value(mask), input;
        for (auto &p : prepared_inputs) input.emplace_back(ctx, p, p.degree());
        std::vector<std::size_t> ps;
        if (n_lookup_parts > 1) ps.assign(lookup_part_sizes, lookup_part_sizes + n_lookup_parts);
        auto ex = placeholder_lookup_hip<Curve>::prove_eval(ctx, input, value, prover.sorted_dfs, q_last, q_blind, l0, ch(c0), ch(c0 + 1), alphas, usable_rows, root, ps,
                                                            part_alphas);
        if (!same_bits(ctx, ex.V_L, prover.V_L_dfs.at(0)) || ex.parts_dfs.size() != prover.parts_dfs.size()) return -314;
        for (size_t i = 0; i < ex.parts_dfs.size(); ++i)
            if (!same_bits(ctx, ex.parts_dfs[i], prover.parts_dfs[i])) return -315;
        for (int f = 0; f < 4; ++f)
            if (!same_bits(ctx, ex.F_dfs[f], look.F_dfs[f])) return -316;
    }
    o = out_look;
    ctx.d2h(o, prover.V_L_dfs.at(0).data(), n * 32);
    o += 4 * n;
    for (auto &p : prover.parts_dfs) {
        ctx.d2h(o, p.data(), n * 32);
        o += 4 * n;
    }
    for (auto &p : prover.sorted_dfs) {
        ctx.d2h(o, p.data(), n * 32);
        o += 4 * n;
    }
    for (int f = 0; f < 4; ++f) {
        const size_t sz = look.F_dfs[f].size();
        if (sz > 16 * n) return -317;
        look_sizes[f] = sz;
        auto c = look.F_dfs[f].coefficients(root);
        ctx.d2h(o + 4 * (size_t)f * 16 * n, c.get(), sz * 32);
    }
    if (events.size() > cap_events) return -318;
    std::memcpy(out_events, events.data(), events.size() * 8);
    *n_events = events.size();
    {
        /* a looked-up value that is in NO table (the reference's BOOST_ASSERT, lookup_argument.hpp:583) must throw BEFORE the
           argument touches the commitment scheme or the transcript -- a second prover over fresh recorders, one input entry made foreign */
        std::vector<poly> bad_inputs = prepared_inputs;
        if (!bad_inputs.empty() && usable_rows > 1) {
            {
                std::vector<Fr> v;
                for (size_t i = 0; i < bad_inputs[0].size(); ++i) v.push_back(bad_inputs[0][i]);
                const size_t row1 = v.size() / n;    // an input on the 2 n-point domain is reduced to every second entry: this one is row 1
                v[row1] = v[row1] + Fr((std::uint64_t)0x1234567) * ch(0);    // no table holds this (overwhelmingly)
                bad_inputs[0] = poly(bad_inputs[0].degree(), std::move(v));
            }
            std::vector<uint64_t> ev2;
            scheme_type inner2(params, detail::field_roots<Curve>());
            recording_scheme<scheme_type> scheme2 {inner2, &ev2};
            transcript_type tr2 {&ev2, {}, 0};
            tr2.challenges = transcript.challenges;
            tr2.next = perm_challenges;
            auto bad_hook = [&](const Fr &) { return bad_inputs; };
            auto prover2 = make_placeholder_lookup_argument_prover<Curve>(cs, pd, table, scheme2, tr2, bad_hook);
            const size_t events_before = ev2.size(), drawn_before = tr2.next;
            bool thrown = false;
            try {
                (void)prover2.prove_eval();
            } catch (const std::runtime_error &e) {
                thrown = std::string(e.what()).find("in no lookup table") != std::string::npos;
            }
            if (!thrown) return -320;
            if (ev2.size() != events_before || tr2.next != drawn_before) return -321;    // nothing appended, committed, absorbed or drawn
            if (ctx.device_status() != 0) return -322;                                   // ... and the sticky word was cleared by the report
        }
    }
    return 0;
}

}    // namespace

extern "C" int shim_placeholder_arguments_transcript(int curve, const uint64_t *srs, size_t n_srs, const uint64_t *evals, size_t k, size_t log_n, size_t usable_rows,
                                                     size_t max_quotient_chunks, size_t permutation_parts, size_t options, size_t columns, size_t k_in, const uint64_t *in_logs,
                                                     const uint64_t *lookup_part_sizes, size_t n_lookup_parts, const uint64_t *challenges, size_t n_challenges,
                                                     uint64_t *out_events, size_t cap_events, uint64_t *n_events, uint64_t *out_perm, uint64_t *perm_sizes,
                                                     uint64_t *out_look, uint64_t *look_sizes) {
    try {
        /* BLS12-381 through a curve that names its scalar field (challenge<FieldType>()), BN254 through the shim's own curve (.challenge()) */
        if (curve == ZKHIP_BLS12_381)
            return arguments_transcript_t<named_field_curve<ZKHIP_BLS12_381>>(srs, n_srs, evals, k, log_n, usable_rows, max_quotient_chunks, permutation_parts, options, columns,
                                                                              k_in, in_logs, lookup_part_sizes, n_lookup_parts, challenges, n_challenges, out_events,
                                                                              cap_events, n_events, out_perm, perm_sizes, out_look, look_sizes);
        return arguments_transcript_t<alt_bn128_254>(srs, n_srs, evals, k, log_n, usable_rows, max_quotient_chunks, permutation_parts, options, columns, k_in, in_logs,
                                                     lookup_part_sizes, n_lookup_parts, challenges, n_challenges, out_events, cap_events, n_events, out_perm, perm_sizes,
                                                     out_look, look_sizes);
    } catch (const std::exception &e) {
        fprintf(stderr, "shim_placeholder_arguments_transcript: %s\n", e.what());
        return -1;
    }
}
