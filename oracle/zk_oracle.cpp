// TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the reference's MSM / NTT / Groth16 hot path.
//
// This file is the checker and the timed CPU baseline, never the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.  The product
// (crypto3-zk_amd/libzkhip.so) neither links nor calls it.
//
// The reference (NilFoundation/crypto3-zk) cannot be built here: its arithmetic lives in the
// un-vendored crypto3-algebra / -math / -multiprecision + Boost (CMakeLists.txt:59-64), none of which
// is on disk.  This file restates the published algorithms those libraries implement (libff lineage)
// and is anchored on the reference's call sites:
//   algebra::multiexp<BDLO12>(b,e,s,e,chunks)      g16 prover.hpp:108-139, kzg.hpp:143-148,409-435
//   kc_multiexp_with_mixed_addition                 knowledge_commitment_multiexp.hpp:57-108
//   evaluation_domain::fft / inverse_fft            r1cs_to_qap.hpp:250-310
//   multiply_by_coset / divide_by_z_on_coset        r1cs_to_qap.hpp:266-315
//   r1cs_to_qap::witness_map                        r1cs_to_qap.hpp:219-325
//   r1cs_to_qap::instance_map_with_evaluation       r1cs_to_qap.hpp:138-187
//   r1cs_gg_ppzksnark_generator (fixed trapdoor)    generator.hpp:86-236, 240-377
//   r1cs_gg_ppzksnark_prover::process               prover.hpp:73-158
//   generate_r1cs_example_with_field_input          test/systems/ppzksnark/r1cs_examples.hpp:77-140
// Pinned by tests/test_oracle_*.py against oracle/pyoracle.py, which itself reproduces the reference's
// bellperson known-answer vectors (AGG:578-862, :864-930; kzg.cpp:75-103).  NTT outputs and whole
// Groth16 proofs are "parity unpinned" in the reference (no test asserts them); they are pinned here
// to the DFT definition and to the trapdoor identities.
//
// Build: g++ -O3 -march=native -fopenmp -shared -fPIC zk_oracle.cpp -o liboracle.so   (see Makefile)

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------------------
// Montgomery prime field, N 64-bit limbs
// ------------------------------------------------------------------------------------------------
template <int N>
struct FieldConsts {
    uint64_t mod[N], r1[N], r2[N], inv;
};

template <int N>
static inline bool geq(const uint64_t *a, const uint64_t *b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
}
template <int N>
static inline uint64_t add_n(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    u128 c = 0;
    for (int i = 0; i < N; ++i) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
template <int N>
static inline uint64_t sub_n(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    uint64_t br = 0;
    for (int i = 0; i < N; ++i) {
        u128 d = (u128)a[i] - b[i] - br;
        r[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}

template <int N_, int ID>
struct Fp {
    static constexpr int N = N_;
    static FieldConsts<N_> C;
    uint64_t v[N];

    static void init(const uint64_t *modulus) {
        memcpy(C.mod, modulus, sizeof(C.mod));
        uint64_t inv = 1;
        for (int i = 0; i < 6; ++i) inv *= 2 - modulus[0] * inv;  // Newton: inv = mod^-1 mod 2^64
        C.inv = (uint64_t)0 - inv;
        // r1 = 2^(64N) mod p by doubling 1
        uint64_t x[N] = {1};
        for (int i = 0; i < 64 * N * 2; ++i) {
            uint64_t carry = add_n<N>(x, x, x);
            if (carry || geq<N>(x, C.mod)) sub_n<N>(x, x, C.mod);
            if (i == 64 * N - 1) memcpy(C.r1, x, sizeof(x));
        }
        memcpy(C.r2, x, sizeof(x));
    }
    static Fp zero() {
        Fp r;
        memset(r.v, 0, sizeof(r.v));
        return r;
    }
    static Fp one() {
        Fp r;
        memcpy(r.v, C.r1, sizeof(r.v));
        return r;
    }
    bool is_zero() const {
        uint64_t o = 0;
        for (int i = 0; i < N; ++i) o |= v[i];
        return o == 0;
    }
    bool operator==(const Fp &b) const { return memcmp(v, b.v, sizeof(v)) == 0; }
    bool operator!=(const Fp &b) const { return !(*this == b); }
    Fp operator+(const Fp &b) const {
        Fp r;
        uint64_t c = add_n<N>(r.v, v, b.v);
        if (c || geq<N>(r.v, C.mod)) sub_n<N>(r.v, r.v, C.mod);
        return r;
    }
    Fp operator-(const Fp &b) const {
        Fp r;
        if (sub_n<N>(r.v, v, b.v)) add_n<N>(r.v, r.v, C.mod);
        return r;
    }
    Fp neg() const { return zero() - *this; }
    Fp dbl() const { return *this + *this; }
    // CIOS Montgomery product
    static inline void mont(uint64_t *out, const uint64_t *a, const uint64_t *b) {
        uint64_t t[N + 2];
        memset(t, 0, sizeof(t));
        for (int i = 0; i < N; ++i) {
            u128 c = 0;
            for (int j = 0; j < N; ++j) {
                c += (u128)a[j] * b[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N] = (uint64_t)c;
            t[N + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * C.inv;
            c = (u128)m * C.mod[0] + t[0];
            c >>= 64;
            for (int j = 1; j < N; ++j) {
                c += (u128)m * C.mod[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N - 1] = (uint64_t)c;
            t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        if (t[N] || geq<N>(t, C.mod)) sub_n<N>(t, t, C.mod);
        memcpy(out, t, sizeof(uint64_t) * N);
    }
    Fp operator*(const Fp &b) const {
        Fp r;
        mont(r.v, v, b.v);
        return r;
    }
    Fp sqr() const { return *this * *this; }
    // canonical <-> Montgomery
    static Fp from_canonical(const uint64_t *c) {
        Fp r;
        mont(r.v, c, C.r2);
        return r;
    }
    void to_canonical(uint64_t *out) const {
        uint64_t o[N] = {1};
        mont(out, v, o);
    }
    static Fp from_u64(uint64_t k) {
        uint64_t c[N] = {k};
        return from_canonical(c);
    }
    Fp pow(const uint64_t *e, int limbs) const {
        Fp r = one();
        for (int i = limbs * 64 - 1; i >= 0; --i) {
            r = r.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) r = r * *this;
        }
        return r;
    }
    Fp pow_u64(uint64_t e) const { return pow(&e, 1); }
    Fp inv() const {  // Fermat
        uint64_t e[N], two[N] = {2};
        sub_n<N>(e, C.mod, two);
        return pow(e, N);
    }
};
template <int N_, int ID>
FieldConsts<N_> Fp<N_, ID>::C;

// Fq2 = Fq[u]/(u^2+1) (both curves)
template <class F>
struct Fp2T {
    F c0, c1;
    static Fp2T zero() { return {F::zero(), F::zero()}; }
    static Fp2T one() { return {F::one(), F::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2T &b) const { return c0 == b.c0 && c1 == b.c1; }
    bool operator!=(const Fp2T &b) const { return !(*this == b); }
    Fp2T operator+(const Fp2T &b) const { return {c0 + b.c0, c1 + b.c1}; }
    Fp2T operator-(const Fp2T &b) const { return {c0 - b.c0, c1 - b.c1}; }
    Fp2T neg() const { return {c0.neg(), c1.neg()}; }
    Fp2T dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fp2T operator*(const Fp2T &b) const {
        F a = c0 * b.c0, bb = c1 * b.c1;
        F c = (c0 + c1) * (b.c0 + b.c1);
        return {a - bb, c - a - bb};
    }
    Fp2T sqr() const {
        F a = (c0 + c1) * (c0 - c1);
        F b = c0 * c1;
        return {a, b.dbl()};
    }
    Fp2T inv() const {
        F n = (c0.sqr() + c1.sqr()).inv();
        return {c0 * n, (c1 * n).neg()};
    }
};

// ------------------------------------------------------------------------------------------------
// y^2 = x^3 + b, Jacobian
// ------------------------------------------------------------------------------------------------
template <class F>
struct Affine {
    F x, y;
    bool inf;
};
template <class F>
struct Jac {
    F X, Y, Z;
    static Jac infinity() { return {F::one(), F::one(), F::zero()}; }
    bool is_inf() const { return Z.is_zero(); }
    static Jac from_affine(const Affine<F> &a) {
        if (a.inf) return infinity();
        return {a.x, a.y, F::one()};
    }
    Jac neg() const { return {X, Y.neg(), Z}; }
    Jac dbl() const {
        if (is_inf() || Y.is_zero()) return infinity();
        F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        F D = ((X + B).sqr() - A - C).dbl();
        F E = A.dbl() + A;
        F Fv = E.sqr();
        F X3 = Fv - D.dbl();
        F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
        F Z3 = (Y * Z).dbl();
        return {X3, Y3, Z3};
    }
    Jac add(const Jac &q) const {
        if (is_inf()) return q;
        if (q.is_inf()) return *this;
        F Z1Z1 = Z.sqr(), Z2Z2 = q.Z.sqr();
        F U1 = X * Z2Z2, U2 = q.X * Z1Z1;
        F S1 = Y * q.Z * Z2Z2, S2 = q.Y * Z * Z1Z1;
        if (U1 == U2) {
            if (S1 == S2) return dbl();
            return infinity();
        }
        F H = U2 - U1, R = S2 - S1;
        F HH = H.sqr(), HHH = H * HH, V = U1 * HH;
        F X3 = R.sqr() - HHH - V.dbl();
        F Y3 = R * (V - X3) - S1 * HHH;
        F Z3 = Z * q.Z * H;
        return {X3, Y3, Z3};
    }
    Jac madd(const Affine<F> &q) const {  // mixed addition (Z2 = 1)
        if (q.inf) return *this;
        if (is_inf()) return from_affine(q);
        F Z1Z1 = Z.sqr();
        F U2 = q.x * Z1Z1, S2 = q.y * Z * Z1Z1;
        if (X == U2) {
            if (Y == S2) return dbl();
            return infinity();
        }
        F H = U2 - X, R = S2 - Y;
        F HH = H.sqr(), HHH = H * HH, V = X * HH;
        F X3 = R.sqr() - HHH - V.dbl();
        F Y3 = R * (V - X3) - Y * HHH;
        F Z3 = Z * H;
        return {X3, Y3, Z3};
    }
    Affine<F> to_affine() const {
        if (is_inf()) return {F::zero(), F::zero(), true};
        F zi = Z.inv(), zi2 = zi.sqr();
        return {X * zi2, Y * zi2 * zi, false};
    }
    Jac mul(const uint64_t *k, int limbs) const {
        Jac r = infinity();
        for (int i = limbs * 64 - 1; i >= 0; --i) {
            r = r.dbl();
            if ((k[i / 64] >> (i % 64)) & 1) r = r.add(*this);
        }
        return r;
    }
};

// ------------------------------------------------------------------------------------------------
// curve instantiation
// ------------------------------------------------------------------------------------------------
typedef Fp<6, 0> FqBLS;
typedef Fp<4, 1> FrBLS;
typedef Fp<4, 2> FqBN;
typedef Fp<4, 3> FrBN;

static const uint64_t BLS_P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                  0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t BLS_R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                  0x73eda753299d7d48ULL};
static const uint64_t BN_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                 0x30644e72e131a029ULL};
static const uint64_t BN_R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                 0x30644e72e131a029ULL};
// standard generators (canonical limbs, little-endian)
static const uint64_t BLS_G1[12] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                                    0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL,
                                    0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                                    0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
static const uint64_t BLS_G2[24] = {
    0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL,
    0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL, 0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL,
    0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL,
    0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL,
    0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL, 0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL,
    0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};
static const uint64_t BN_G1[8] = {1, 0, 0, 0, 2, 0, 0, 0};
static const uint64_t BN_G2[16] = {
    0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL,
    0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL,
    0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL,
    0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};

static void init_fields() {
    static bool done = false;
    if (done) return;
    FqBLS::init(BLS_P);
    FrBLS::init(BLS_R);
    FqBN::init(BN_P);
    FrBN::init(BN_R);
    done = true;
}
struct Init {
    Init() { init_fields(); }
} g_init;

// traits per (curve, group)
template <int CURVE, int GROUP>
struct Tr;
template <>
struct Tr<0, 1> {
    typedef FqBLS F;
    typedef FrBLS S;
    static constexpr int FL = 6;
    static const uint64_t *gen() { return BLS_G1; }
};
template <>
struct Tr<0, 2> {
    typedef Fp2T<FqBLS> F;
    typedef FrBLS S;
    static constexpr int FL = 12;
    static const uint64_t *gen() { return BLS_G2; }
};
template <>
struct Tr<1, 1> {
    typedef FqBN F;
    typedef FrBN S;
    static constexpr int FL = 4;
    static const uint64_t *gen() { return BN_G1; }
};
template <>
struct Tr<1, 2> {
    typedef Fp2T<FqBN> F;
    typedef FrBN S;
    static constexpr int FL = 8;
    static const uint64_t *gen() { return BN_G2; }
};

template <class F>
struct IO;
template <int N, int ID>
struct IO<Fp<N, ID>> {
    static constexpr int L = N;
    static Fp<N, ID> load(const uint64_t *p) { return Fp<N, ID>::from_canonical(p); }
    static void store(uint64_t *p, const Fp<N, ID> &f) { f.to_canonical(p); }
};
template <class B>
struct IO<Fp2T<B>> {
    static constexpr int L = 2 * B::N;
    static Fp2T<B> load(const uint64_t *p) { return {B::from_canonical(p), B::from_canonical(p + B::N)}; }
    static void store(uint64_t *p, const Fp2T<B> &f) {
        f.c0.to_canonical(p);
        f.c1.to_canonical(p + B::N);
    }
};

template <class F>
static Affine<F> load_affine(const uint64_t *p, bool inf) {
    if (inf) return {F::zero(), F::zero(), true};
    return {IO<F>::load(p), IO<F>::load(p + IO<F>::L), false};
}
template <class F>
static void store_affine(uint64_t *p, uint8_t *inf, const Affine<F> &a) {
    if (a.inf) {
        memset(p, 0, sizeof(uint64_t) * 2 * IO<F>::L);
        if (inf) *inf = 1;
        return;
    }
    IO<F>::store(p, a.x);
    IO<F>::store(p + IO<F>::L, a.y);
    if (inf) *inf = 0;
}

// ------------------------------------------------------------------------------------------------
// MSM: BDLO12 bucket method with `chunks` (prover.hpp:94-99 passes omp_get_max_threads())
// ------------------------------------------------------------------------------------------------
static inline unsigned get_bits(const uint64_t *s, int limbs, int lo, int c) {
    // bits [lo, lo+c) of the little-endian integer s
    unsigned r = 0;
    int limb = lo / 64, off = lo % 64;
    if (limb >= limbs) return 0;
    uint64_t v = s[limb] >> off;
    if (off + c > 64 && limb + 1 < limbs) v |= s[limb + 1] << (64 - off);
    r = (unsigned)(v & ((1ULL << c) - 1));
    return r;
}
static inline int bit_length(const uint64_t *s, int limbs) {
    for (int i = limbs - 1; i >= 0; --i)
        if (s[i]) return 64 * i + 64 - __builtin_clzll(s[i]);
    return 0;
}

template <class F>
static Jac<F> pippenger_inner(const Affine<F> *bases, const uint64_t *scalars /*canonical, 4 limbs each*/,
                              size_t n) {
    if (n == 0) return Jac<F>::infinity();
    int lg = 0;
    while (((size_t)2 << lg) <= n) ++lg;
    int c = lg < 6 ? std::max(1, lg) : lg - (lg / 3 - 2);
    int nbits = 0;
    for (size_t i = 0; i < n; ++i) nbits = std::max(nbits, bit_length(scalars + 4 * i, 4));
    int groups = (nbits + c - 1) / c;
    Jac<F> result = Jac<F>::infinity();
    std::vector<Jac<F>> buckets((size_t)1 << c);
    for (int k = groups - 1; k >= 0; --k) {
        for (int i = 0; i < c; ++i) result = result.dbl();
        std::fill(buckets.begin(), buckets.end(), Jac<F>::infinity());
        for (size_t i = 0; i < n; ++i) {
            unsigned d = get_bits(scalars + 4 * i, 4, k * c, c);
            if (d) buckets[d] = buckets[d].madd(bases[i]);
        }
        Jac<F> running = Jac<F>::infinity();
        for (size_t i = ((size_t)1 << c) - 1; i >= 1; --i) {
            running = running.add(buckets[i]);
            result = result.add(running);
        }
    }
    return result;
}

template <class F>
static Jac<F> multiexp(const Affine<F> *bases, const uint64_t *scalars, size_t n, int chunks) {
    if (chunks < 1) chunks = 1;
    if ((size_t)chunks > n) chunks = n ? (int)n : 1;
    if (chunks == 1) return pippenger_inner<F>(bases, scalars, n);
    std::vector<Jac<F>> part(chunks);
    size_t per = n / chunks;
#pragma omp parallel for schedule(static, 1)
    for (int i = 0; i < chunks; ++i) {
        size_t lo = per * i, hi = (i == chunks - 1) ? n : per * (i + 1);
        part[i] = pippenger_inner<F>(bases + lo, scalars + 4 * lo, hi - lo);
    }
    Jac<F> r = Jac<F>::infinity();
    for (int i = 0; i < chunks; ++i) r = r.add(part[i]);
    return r;
}

template <class F>
static Jac<F> msm_naive(const Affine<F> *bases, const uint64_t *scalars, size_t n) {
    Jac<F> r = Jac<F>::infinity();
    for (size_t i = 0; i < n; ++i) r = r.add(Jac<F>::from_affine(bases[i]).mul(scalars + 4 * i, 4));
    return r;
}

// fixed-base multiples of a point with an 8-bit window table (batch_exp in generator.hpp:187-214)
template <class F>
static void batch_mul(const Affine<F> &base, const uint64_t *scalars, size_t n, Affine<F> *out) {
    // fixed-base windows of W bits with an AFFINE table (mixed additions: 7M + 4S instead of 11M + 5S); for many scalars 12-bit windows
    // (22 additions per point instead of 32; the table -- 22 x 4096 points -- is built once, in parallel over the windows).  Round 5: the
    // 2^20-constraint key the CPU baseline needs took 18-21 s with 8-bit windows over a Jacobian table, half of bench.py's default run.
    const int W = n >= 4096 ? 12 : 8, NW = (256 + W - 1) / W;
    std::vector<Affine<F>> table((size_t)NW << W);
    {
        std::vector<Jac<F>> start(NW);
        Jac<F> b = Jac<F>::from_affine(base);
        for (int w = 0; w < NW; ++w) {
            start[w] = b;
            for (int i = 0; i < W; ++i) b = b.dbl();
        }
#pragma omp parallel for schedule(dynamic, 1)
        for (int w = 0; w < NW; ++w) {
            const size_t cnt = (size_t)1 << W;
            std::vector<Jac<F>> row(cnt);
            row[0] = Jac<F>::infinity();
            for (size_t i = 1; i < cnt; ++i) row[i] = row[i - 1].add(start[w]);
            // to affine with one inversion per row (Montgomery's trick)
            std::vector<F> pre(cnt);
            F run = F::one();
            for (size_t i = 0; i < cnt; ++i) {
                pre[i] = run;
                if (!row[i].is_inf()) run = run * row[i].Z;
            }
            F inv = run.inv();
            Affine<F> *dst = &table[(size_t)w << W];
            for (size_t i = cnt; i-- > 0;) {
                if (row[i].is_inf()) {
                    dst[i] = {F::zero(), F::zero(), true};
                    continue;
                }
                F zi = inv * pre[i], zi2 = zi.sqr();
                inv = inv * row[i].Z;
                dst[i] = {row[i].X * zi2, row[i].Y * zi2 * zi, false};
            }
        }
    }
    // chunks of 256 results share ONE inversion (Montgomery's trick; batch_to_special in generator.hpp:190-192 does the same):
    // a 2^20-constraint key (5.2 M G1 + 1 M G2 points) takes tens of seconds instead of minutes, which is what lets bench.py time
    // the CPU prover at the size of the headline metric.  Same affine points as one inversion each.
    const size_t CH = 256, nch = (n + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t c = 0; c < nch; ++c) {
        const size_t lo = c * CH, hi = std::min(n, lo + CH);
        std::vector<Jac<F>> acc(hi - lo);
        std::vector<F> pre(hi - lo);
        F run = F::one();
        for (size_t i = lo; i < hi; ++i) {
            Jac<F> a = Jac<F>::infinity();
            for (int w = 0; w < NW; ++w) {
                unsigned d = get_bits(scalars + 4 * i, 4, w * W, W);
                if (d) a = a.madd(table[((size_t)w << W) + d]);
            }
            acc[i - lo] = a;
            pre[i - lo] = run;  // product of the non-zero Z before this one
            if (!a.is_inf()) run = run * a.Z;
        }
        F inv = run.inv();
        for (size_t i = hi; i-- > lo;) {
            const Jac<F> &a = acc[i - lo];
            if (a.is_inf()) {
                out[i] = {F::zero(), F::zero(), true};
                continue;
            }
            F zi = inv * pre[i - lo], zi2 = zi.sqr();
            inv = inv * a.Z;
            out[i] = {a.X * zi2, a.Y * zi2 * zi, false};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NTT (natural in, natural out; data canonical, twiddles Montgomery)
// ------------------------------------------------------------------------------------------------
template <class S>
static void ntt_inplace(S *a, size_t log_m, const S &omega) {
    size_t m = (size_t)1 << log_m;
    for (size_t i = 0; i < m; ++i) {
        size_t j = 0;
        for (size_t b = 0; b < log_m; ++b) j |= ((i >> b) & 1) << (log_m - 1 - b);
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t s = 1; s <= log_m; ++s) {
        size_t half = (size_t)1 << (s - 1);
        S wm = omega.pow_u64(m >> s);
        for (size_t k = 0; k < m; k += 2 * half) {
            S w = S::one();
            for (size_t j = 0; j < half; ++j) {
                S t = w * a[k + j + half];
                S u = a[k + j];
                a[k + j] = u + t;
                a[k + j + half] = u - t;
                w = w * wm;
            }
        }
    }
}

// The same transform with the threads INSIDE it (a batch smaller than the host's thread count would leave most cores idle when
// every polynomial is one thread's work: VERDICT r5 weak #3).  Same butterflies on the same operands -- the twiddle of (stage s, j)
// is omega^(j m / 2^s) either way, a unique field element --, so the output is the serial transform's bit for bit.
template <class S>
static void ntt_inplace_threads(S *a, size_t log_m, const S &omega, const std::vector<S> &tw /* omega^i, i < m / 2 */) {
    const size_t m = (size_t)1 << log_m;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; ++i) {
        size_t j = 0;
        for (size_t b = 0; b < log_m; ++b) j |= ((i >> b) & 1) << (log_m - 1 - b);
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t s = 1; s <= log_m; ++s) {
        const size_t half = (size_t)1 << (s - 1), stride = m >> s;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < m / 2; ++i) {
            const size_t j = i & (half - 1), k = (i >> (s - 1)) << s;
            S t = tw[j * stride] * a[k + j + half];
            S u = a[k + j];
            a[k + j] = u + t;
            a[k + j + half] = u - t;
        }
    }
}

template <class S>
static void ntt_batch(uint64_t *data, size_t log_m, size_t batch, const uint64_t *omega_c, int inverse,
                      const uint64_t *coset_c) {
    size_t m = (size_t)1 << log_m;
    S omega = S::from_canonical(omega_c);
    if (inverse) omega = omega.inv();
    S minv = S::from_u64(m).inv();
    if (log_m >= 14 && batch * 2 <= (size_t)omp_get_max_threads()) {
        // fewer polynomials than half the threads: one polynomial at a time, all threads inside the transform
        std::vector<S> tw(m / 2), a(m), pw(m);
        const int nt = omp_get_max_threads();
        auto powers = [&](std::vector<S> &out, size_t count, const S &base, const S &first) {  // out[i] = first base^i, by blocks
            const size_t per = (count + nt - 1) / nt;
#pragma omp parallel for schedule(static, 1)
            for (int t = 0; t < nt; ++t) {
                const size_t lo = std::min(count, per * t), hi = std::min(count, lo + per);
                if (lo >= hi) continue;
                S x = first * base.pow_u64(lo);
                for (size_t i = lo; i < hi; ++i) {
                    out[i] = x;
                    x = x * base;
                }
            }
        };
        powers(tw, m / 2, omega, S::one());
        const bool pre = !inverse && coset_c, post = inverse != 0;
        if (pre) powers(pw, m, S::from_canonical(coset_c), S::one());
        if (post) powers(pw, m, coset_c ? S::from_canonical(coset_c).inv() : S::one(), minv);
        for (size_t b = 0; b < batch; ++b) {
            uint64_t *d = data + b * m * 4;
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < m; ++i) {
                a[i] = S::from_canonical(d + 4 * i);
                if (pre) a[i] = a[i] * pw[i];
            }
            ntt_inplace_threads<S>(a.data(), log_m, omega, tw);
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < m; ++i) {
                if (post) a[i] = a[i] * pw[i];
                a[i].to_canonical(d + 4 * i);
            }
        }
        return;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t b = 0; b < batch; ++b) {
        std::vector<S> a(m);
        uint64_t *d = data + b * m * 4;
        for (size_t i = 0; i < m; ++i) a[i] = S::from_canonical(d + 4 * i);
        if (!inverse && coset_c) {
            S g = S::from_canonical(coset_c), x = S::one();
            for (size_t i = 0; i < m; ++i) {
                a[i] = a[i] * x;
                x = x * g;
            }
        }
        ntt_inplace<S>(a.data(), log_m, omega);
        if (inverse) {
            S g = coset_c ? S::from_canonical(coset_c).inv() : S::one(), x = minv;
            for (size_t i = 0; i < m; ++i) {
                a[i] = a[i] * x;
                x = x * g;
            }
        }
        for (size_t i = 0; i < m; ++i) a[i].to_canonical(d + 4 * i);
    }
}

// ------------------------------------------------------------------------------------------------
// Evaluation domains: what math::make_evaluation_domain(min_size) returns (r1cs_to_qap.hpp:138-139, 229-230).
// crypto3-math is not in /root/reference; this restates the published algorithm of its libfqfft lineage
// (get_evaluation_domain: basic, extended, step radix-2 at min_size, then at big + rounded_small; the domains'
// FFT / iFFT / evaluate_all_lagrange_polynomials / compute_vanishing_polynomial / divide_by_Z_on_coset).
// Pinned by tests/test_oracle_kat.py against pyoracle.EvaluationDomain, which is held to the definitions over the
// domain's point set.
// ------------------------------------------------------------------------------------------------
static inline size_t ceil_log2(size_t n) {
    size_t r = 0;
    while (((size_t)1 << r) < n) ++r;
    return r;
}
enum { DOM_BASIC = 0, DOM_EXTENDED = 1, DOM_STEP = 2 };
// (kind, m) for a field of the given two-adicity; returns false when only a geometric / arithmetic domain would do
static bool domain_choice(size_t min_size, size_t s, int &kind, size_t &m) {
    auto basic_ok = [&](size_t n) { return n > 1 && n == (size_t)1 << ceil_log2(n) && ceil_log2(n) <= s; };
    auto ext_ok = [&](size_t n) { return n > 1 && ceil_log2(n) == s + 1 && n == (size_t)1 << (s + 1); };
    auto step_ok = [&](size_t n) {
        if (n <= 1) return false;
        size_t small = n - ((size_t)1 << (ceil_log2(n) - 1));
        return small == (size_t)1 << ceil_log2(small) && ceil_log2(n) <= s;
    };
    if (min_size <= 1) return false;
    size_t big = (size_t)1 << (ceil_log2(min_size) - 1), small = min_size - big, rounded = big + ((size_t)1 << ceil_log2(small));
    for (size_t n : {min_size, rounded}) {
        if (basic_ok(n)) return kind = DOM_BASIC, m = n, true;
        if (ext_ok(n)) return kind = DOM_EXTENDED, m = n, true;
        if (step_ok(n)) return kind = DOM_STEP, m = n, true;
    }
    return false;
}

template <class S>
struct Domain {
    int kind = DOM_BASIC;
    size_t m = 0, big_m = 0, small_m = 0, compr = 1;
    S omega, shift, big_omega, small_omega;  // omega: basic m-th / extended (m/2)-th / step (2 big_m)-th primitive root

    static Domain make(int kind, size_t m, const S &omega, const S &shift) {
        Domain d;
        d.kind = kind;
        d.m = m;
        d.omega = omega;
        d.shift = shift;
        if (kind == DOM_EXTENDED) {
            d.small_m = m / 2;
        } else if (kind == DOM_STEP) {
            d.big_m = (size_t)1 << (ceil_log2(m) - 1);
            d.small_m = m - d.big_m;
            d.compr = d.big_m / d.small_m;
            d.big_omega = omega * omega;
            d.small_omega = omega.pow_u64(2 * d.compr);
        }
        return d;
    }
    static void basic_fft(std::vector<S> &a, const S &w) { ntt_inplace<S>(a.data(), ceil_log2(a.size()), w); }
    static void scale(std::vector<S> &a, const S &c) {
        for (auto &x : a) x = x * c;
    }
    // L_i(t) over {w^i}, i < n: (t^n - 1) w^i / (n (t - w^i)), one inversion
    static std::vector<S> basic_lagrange(size_t n, const S &w, const S &t) {
        std::vector<S> u(n), den(n), pre(n);
        S x = S::one(), acc = S::one();
        for (size_t i = 0; i < n; ++i) {
            den[i] = t - x;
            u[i] = x;
            pre[i] = acc;
            acc = acc * den[i];
            x = x * w;
        }
        S inv = acc.inv(), z = (t.pow_u64(n) - S::one()) * S::from_u64(n).inv();
        for (size_t i = n; i-- > 0;) {
            S di = inv * pre[i];
            inv = inv * den[i];
            u[i] = u[i] * z * di;
        }
        return u;
    }
    void fft(std::vector<S> &a) const {
        if (kind == DOM_BASIC) return basic_fft(a, omega);
        if (kind == DOM_EXTENDED) {
            std::vector<S> a0(small_m), a1(small_m);
            S s_sm = shift.pow_u64(small_m), x = S::one();
            for (size_t i = 0; i < small_m; ++i) {
                a0[i] = a[i] + a[small_m + i];
                a1[i] = x * (a[i] + s_sm * a[small_m + i]);
                x = x * shift;
            }
            basic_fft(a0, omega);
            basic_fft(a1, omega);
            for (size_t i = 0; i < small_m; ++i) a[i] = a0[i], a[small_m + i] = a1[i];
            return;
        }
        std::vector<S> c(big_m), e(small_m, S::zero());
        S wi = S::one();
        for (size_t i = 0; i < big_m; ++i) {
            c[i] = i < small_m ? a[i] + a[i + big_m] : a[i];
            S d = wi * (i < small_m ? a[i] - a[i + big_m] : a[i]);
            e[i % small_m] = e[i % small_m] + d;
            wi = wi * omega;
        }
        basic_fft(c, big_omega);
        basic_fft(e, small_omega);
        for (size_t i = 0; i < big_m; ++i) a[i] = c[i];
        for (size_t i = 0; i < small_m; ++i) a[big_m + i] = e[i];
    }
    void ifft(std::vector<S> &a) const {
        if (kind == DOM_BASIC) {
            basic_fft(a, omega.inv());
            scale(a, S::from_u64(m).inv());
            return;
        }
        if (kind == DOM_EXTENDED) {
            std::vector<S> a0(a.begin(), a.begin() + small_m), a1(a.begin() + small_m, a.end());
            S winv = omega.inv();
            basic_fft(a0, winv);
            basic_fft(a1, winv);
            S s_sm = shift.pow_u64(small_m), sconst = (S::from_u64(small_m) * (S::one() - s_sm)).inv(), shinv = shift.inv(), x = S::one();
            for (size_t i = 0; i < small_m; ++i) {
                a[i] = sconst * (x * a1[i] - s_sm * a0[i]);
                a[small_m + i] = sconst * (a0[i] - x * a1[i]);
                x = x * shinv;
            }
            return;
        }
        std::vector<S> U0(a.begin(), a.begin() + big_m), U1(a.begin() + big_m, a.end());
        basic_fft(U0, big_omega.inv());
        basic_fft(U1, small_omega.inv());
        scale(U0, S::from_u64(big_m).inv());
        scale(U1, S::from_u64(small_m).inv());
        S wi = S::one();
        for (size_t i = 0; i < big_m; ++i) {
            if (i >= small_m) {
                a[i] = U0[i];
                U1[i % small_m] = U1[i % small_m] - U0[i] * wi;
            }
            wi = wi * omega;
        }
        S winv = omega.inv(), x = S::one(), half = S::from_u64(2).inv();
        for (size_t i = 0; i < small_m; ++i) {
            S u1 = U1[i] * x;
            a[i] = (U0[i] + u1) * half;
            a[big_m + i] = (U0[i] - u1) * half;
            x = x * winv;
        }
    }
    S vanishing(const S &t) const {
        if (kind == DOM_BASIC) return t.pow_u64(m) - S::one();
        if (kind == DOM_EXTENDED) {
            S tm = t.pow_u64(small_m);
            return (tm - S::one()) * (tm - shift.pow_u64(small_m));
        }
        return (t.pow_u64(big_m) - S::one()) * (t.pow_u64(small_m) - omega.pow_u64(small_m));
    }
    std::vector<S> lagrange(const S &t) const {
        if (kind == DOM_BASIC) return basic_lagrange(m, omega, t);
        std::vector<S> out(m);
        if (kind == DOM_EXTENDED) {
            auto T0 = basic_lagrange(small_m, omega, t), T1 = basic_lagrange(small_m, omega, t * shift.inv());
            S t_sm = t.pow_u64(small_m), s_sm = shift.pow_u64(small_m), ood = (s_sm - S::one()).inv();
            S c0 = (s_sm - t_sm) * ood, c1 = (t_sm - S::one()) * ood;
            for (size_t i = 0; i < small_m; ++i) out[i] = T0[i] * c0, out[small_m + i] = T1[i] * c1;
            return out;
        }
        auto ib = basic_lagrange(big_m, big_omega, t), is = basic_lagrange(small_m, small_omega, t * omega.inv());
        S w_sm = omega.pow_u64(small_m), L0 = t.pow_u64(small_m) - w_sm;
        // 1 / (big_omega^(small_m i) - omega^small_m) takes compr distinct values
        std::vector<S> dinv(compr);
        S step = big_omega.pow_u64(small_m), elt = S::one();
        for (size_t i = 0; i < compr; ++i) {
            dinv[i] = (elt - w_sm).inv();
            elt = elt * step;
        }
        for (size_t i = 0; i < big_m; ++i) out[i] = ib[i] * L0 * dinv[i % compr];
        S L1 = (t.pow_u64(big_m) - S::one()) * (omega.pow_u64(big_m) - S::one()).inv();
        for (size_t i = 0; i < small_m; ++i) out[big_m + i] = L1 * is[i];
        return out;
    }
    // P[i] /= Z(g x_i) with P the evaluations on the coset g * domain
    void divide_by_z_on_coset(std::vector<S> &P, const S &g) const {
        if (kind == DOM_BASIC) return scale(P, vanishing(g).inv());
        if (kind == DOM_EXTENDED) {
            S z0 = vanishing(g).inv(), z1 = vanishing(g * shift).inv();
            for (size_t i = 0; i < small_m; ++i) P[i] = P[i] * z0, P[small_m + i] = P[small_m + i] * z1;
            return;
        }
        S Z0 = g.pow_u64(big_m) - S::one(), a = g.pow_u64(small_m) * Z0, b = omega.pow_u64(small_m) * Z0;
        std::vector<S> zi(compr);
        S step = omega.pow_u64(2 * small_m), elt = S::one();
        for (size_t i = 0; i < compr; ++i) {
            zi[i] = (a * elt - b).inv();
            elt = elt * step;
        }
        for (size_t i = 0; i < big_m; ++i) P[i] = P[i] * zi[i % compr];
        S z1 = vanishing(g * omega).inv();
        for (size_t i = 0; i < small_m; ++i) P[big_m + i] = P[big_m + i] * z1;
    }
};

// ------------------------------------------------------------------------------------------------
// splitmix64, shared with pyoracle.SplitMix64
// ------------------------------------------------------------------------------------------------
struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    void next_mod(const uint64_t *mod, uint64_t *out) {  // 4 limbs mod `mod`
        for (int i = 0; i < 4; ++i) out[i] = next();
        while (geq<4>(out, mod)) sub_n<4>(out, out, mod);
    }
};

// ------------------------------------------------------------------------------------------------
// R1CS (CSR), QAP, Groth16
// ------------------------------------------------------------------------------------------------
template <class S>
struct Csr {
    std::vector<uint32_t> rowptr, col;
    std::vector<S> coeff;
    S row_dot(size_t i, const std::vector<S> &full /*index 0 = 1*/) const {
        S acc = S::zero();
        for (uint32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) acc = acc + coeff[k] * full[col[k]];
        return acc;
    }
};

template <int CURVE>
struct G16 {
    typedef typename Tr<CURVE, 1>::S S;
    typedef typename Tr<CURVE, 1>::F F1;
    typedef typename Tr<CURVE, 2>::F F2;
    size_t M = 0, n = 0, N = 0, m = 0, log_m = 0;
    // the evaluation domain: unset = the BASIC radix-2 domain of m = 2^log_m >= M + n + 1 points over the omega each call passes;
    // set_domain() installs what make_evaluation_domain(M + n + 1) picks (or any other) and then every call uses it
    bool dom_set = false;
    Domain<S> dom;
    Domain<S> domain(const S &omega) const { return dom_set ? dom : Domain<S>::make(DOM_BASIC, m, omega, S::zero()); }
    void set_domain(int kind, size_t size, const S &omega, const S &shift) {
        dom = Domain<S>::make(kind, size, omega, shift);
        dom_set = true;
        m = size;
        log_m = ceil_log2(size);
    }
    Csr<S> A, B, C;
    std::vector<S> assignment;  // N values (primary then auxiliary)
    // key
    Affine<F1> alpha_g1, beta_g1, delta_g1;
    Affine<F2> beta_g2, delta_g2;
    std::vector<Affine<F1>> A_query, H_query, L_query, B_h;
    std::vector<Affine<F2>> B_g;
    bool has_key = false;

    static const uint64_t *rmod() { return CURVE == 0 ? BLS_R : BN_R; }

    void make_example(size_t num_constraints, size_t num_inputs, uint64_t seed) {
        // r1cs_examples.hpp:77-140
        M = num_constraints;
        n = num_inputs;
        N = 2 + num_constraints;
        SplitMix64 rng{seed};
        uint64_t ac[4], bc[4];
        rng.next_mod(rmod(), ac);
        rng.next_mod(rmod(), bc);
        S a = S::from_canonical(ac), b = S::from_canonical(bc);
        assignment.clear();
        assignment.push_back(a);
        assignment.push_back(b);
        A = B = C = Csr<S>();
        A.rowptr.push_back(0);
        B.rowptr.push_back(0);
        C.rowptr.push_back(0);
        S one = S::one();
        for (size_t i = 0; i + 1 < M; ++i) {
            S tmp;
            if (i % 2) {
                A.col.push_back(i + 1);
                A.coeff.push_back(one);
                B.col.push_back(i + 2);
                B.coeff.push_back(one);
                tmp = a * b;
            } else {
                B.col.push_back(0);
                B.coeff.push_back(one);
                A.col.push_back(i + 1);
                A.coeff.push_back(one);
                A.col.push_back(i + 2);
                A.coeff.push_back(one);
                tmp = a + b;
            }
            C.col.push_back(i + 3);
            C.coeff.push_back(one);
            assignment.push_back(tmp);
            a = b;
            b = tmp;
            A.rowptr.push_back(A.col.size());
            B.rowptr.push_back(B.col.size());
            C.rowptr.push_back(C.col.size());
        }
        S fin = S::zero();
        for (size_t i = 1; i < N; ++i) {
            A.col.push_back(i);
            A.coeff.push_back(one);
            B.col.push_back(i);
            B.coeff.push_back(one);
            fin = fin + assignment[i - 1];
        }
        C.col.push_back(N);
        C.coeff.push_back(one);
        A.rowptr.push_back(A.col.size());
        B.rowptr.push_back(B.col.size());
        C.rowptr.push_back(C.col.size());
        assignment.push_back(fin.sqr());
        m = 1;
        log_m = 0;
        while (m < M + n + 1) {
            m *= 2;
            ++log_m;
        }
    }
    void swap_AB_if_beneficial() {  // r1cs.hpp:193-215
        std::vector<char> ta(N + 1, 0), tb(N + 1, 0);
        for (auto c : A.col) ta[c] = 1;
        for (auto c : B.col) tb[c] = 1;
        size_t na = 0, nb = 0;
        for (size_t i = 0; i <= N; ++i) {
            na += ta[i];
            nb += tb[i];
        }
        if (nb > na) std::swap(A, B);
    }
    std::vector<S> full_with_one() const {
        std::vector<S> f(N + 1);
        f[0] = S::one();
        for (size_t i = 0; i < N; ++i) f[i + 1] = assignment[i];
        return f;
    }
    bool is_satisfied() const {
        auto f = full_with_one();
        for (size_t i = 0; i < M; ++i)
            if (A.row_dot(i, f) * B.row_dot(i, f) != C.row_dot(i, f)) return false;
        return true;
    }
    // r1cs_to_qap.hpp:138-187
    void qap_at(const S &t, const S &omega, std::vector<S> &At, std::vector<S> &Bt, std::vector<S> &Ct,
                S &Zt) const {
        const Domain<S> d = domain(omega);
        Zt = d.vanishing(t);
        const std::vector<S> u = d.lagrange(t);
        At.assign(N + 1, S::zero());
        Bt.assign(N + 1, S::zero());
        Ct.assign(N + 1, S::zero());
        for (size_t i = 0; i <= n; ++i) At[i] = u[M + i];
        for (size_t i = 0; i < M; ++i) {
            for (uint32_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) At[A.col[k]] = At[A.col[k]] + u[i] * A.coeff[k];
            for (uint32_t k = B.rowptr[i]; k < B.rowptr[i + 1]; ++k) Bt[B.col[k]] = Bt[B.col[k]] + u[i] * B.coeff[k];
            for (uint32_t k = C.rowptr[i]; k < C.rowptr[i + 1]; ++k) Ct[C.col[k]] = Ct[C.col[k]] + u[i] * C.coeff[k];
        }
    }
    // generator.hpp:86-236 with fixed trapdoor and the standard generators
    void keygen(const uint64_t *trap /*5x4 canonical: t,alpha,beta,gamma,delta*/, const uint64_t *omega_c) {
        swap_AB_if_beneficial();
        S t = S::from_canonical(trap), alpha = S::from_canonical(trap + 4), beta = S::from_canonical(trap + 8),
          delta = S::from_canonical(trap + 16);
        S omega = S::from_canonical(omega_c);
        std::vector<S> At, Bt, Ct;
        S Zt;
        qap_at(t, omega, At, Bt, Ct, Zt);
        S dinv = delta.inv();
        std::vector<uint64_t> sc;
        auto canon = [&](const std::vector<S> &v) {
            sc.resize(v.size() * 4);
            for (size_t i = 0; i < v.size(); ++i) v[i].to_canonical(&sc[4 * i]);
        };
        Affine<F1> g1 = load_affine<F1>(Tr<CURVE, 1>::gen(), false);
        Affine<F2> g2 = load_affine<F2>(Tr<CURVE, 2>::gen(), false);
        canon(At);
        A_query.resize(N + 1);
        batch_mul<F1>(g1, sc.data(), N + 1, A_query.data());
        canon(Bt);
        B_h.resize(N + 1);
        B_g.resize(N + 1);
        batch_mul<F1>(g1, sc.data(), N + 1, B_h.data());
        batch_mul<F2>(g2, sc.data(), N + 1, B_g.data());
        std::vector<S> Lt(N - n), Hs(m - 1);
        for (size_t i = 0; i < N - n; ++i) Lt[i] = (beta * At[n + 1 + i] + alpha * Bt[n + 1 + i] + Ct[n + 1 + i]) * dinv;
        S ti = S::one(), zd = Zt * dinv;
        for (size_t i = 0; i + 1 < m; ++i) {
            Hs[i] = ti * zd;
            ti = ti * t;
        }
        canon(Lt);
        L_query.resize(N - n);
        batch_mul<F1>(g1, sc.data(), N - n, L_query.data());
        canon(Hs);
        H_query.resize(m - 1);
        batch_mul<F1>(g1, sc.data(), m - 1, H_query.data());
        std::vector<S> fx = {alpha, beta, delta};
        canon(fx);
        Affine<F1> o1[3];
        Affine<F2> o2[3];
        batch_mul<F1>(g1, sc.data(), 3, o1);
        batch_mul<F2>(g2, sc.data(), 3, o2);
        alpha_g1 = o1[0];
        beta_g1 = o1[1];
        delta_g1 = o1[2];
        beta_g2 = o2[1];
        delta_g2 = o2[2];
        has_key = true;
    }
    // The proof of a correct prover as three EXPONENTS of the standard generators, from the trapdoor alone (comment
    // formulas prover.hpp:141,145,151-153; H(t) Z(t) = A(t) B(t) - C(t) for a satisfying assignment):
    //   a = alpha + sum z_i A_i(t) + r delta,  b = beta + sum z_i B_i(t) + s delta,
    //   c = (sum_{i>n} z_i (beta A_i + alpha B_i + C_i)(t) + A(t) B(t) - C(t)) / delta + s a + r b - r s delta
    // No MSM, no NTT, no group arithmetic: O(nnz + m) field operations, usable at 2^20 constraints.
    void expected_exponents(const uint64_t *trap, const uint64_t *omega_c, const uint64_t *r_c, const uint64_t *s_c,
                            uint64_t *out /*3 x 4 canonical*/) {
        swap_AB_if_beneficial();  // the key is generated over the swapped system (generator.hpp:250-252); idempotent
        S t = S::from_canonical(trap), alpha = S::from_canonical(trap + 4), beta = S::from_canonical(trap + 8),
          delta = S::from_canonical(trap + 16);
        S r = S::from_canonical(r_c), s = S::from_canonical(s_c);
        std::vector<S> At, Bt, Ct;
        S Zt;
        qap_at(t, S::from_canonical(omega_c), At, Bt, Ct, Zt);
        auto z = full_with_one();
        S at = S::zero(), bt = S::zero(), ct = S::zero(), lt = S::zero();
        for (size_t i = 0; i <= N; ++i) {
            at = at + z[i] * At[i];
            bt = bt + z[i] * Bt[i];
            ct = ct + z[i] * Ct[i];
            if (i > n) lt = lt + z[i] * (beta * At[i] + alpha * Bt[i] + Ct[i]);
        }
        S a = alpha + at + r * delta, b = beta + bt + s * delta;
        S c = (lt + at * bt - ct) * delta.inv() + s * a + r * b - r * s * delta;
        a.to_canonical(out);
        b.to_canonical(out + 4);
        c.to_canonical(out + 8);
    }
    // r1cs_to_qap.hpp:219-325 (d1 = d2 = d3 = 0): m+1 coefficients
    std::vector<S> witness_map(const S &omega, const S &g) const {
        auto f = full_with_one();
        std::vector<S> aA(m, S::zero()), aB(m, S::zero()), aC(m, S::zero());
        for (size_t i = 0; i <= n; ++i) aA[i + M] = f[i];
#pragma omp parallel for
        for (size_t i = 0; i < M; ++i) {
            aA[i] = aA[i] + A.row_dot(i, f);
            aB[i] = aB[i] + B.row_dot(i, f);
            aC[i] = aC[i] + C.row_dot(i, f);
        }
        const Domain<S> d = domain(omega);
        auto coset_fft = [&](std::vector<S> &a) {
            d.ifft(a);
            S x = S::one();
            for (size_t i = 0; i < m; ++i) {
                a[i] = a[i] * x;
                x = x * g;
            }
            d.fft(a);
        };
#pragma omp parallel sections
        {
#pragma omp section
            coset_fft(aA);
#pragma omp section
            coset_fft(aB);
#pragma omp section
            coset_fft(aC);
        }
#pragma omp parallel for
        for (size_t i = 0; i < m; ++i) aA[i] = aA[i] * aB[i] - aC[i];
        d.divide_by_z_on_coset(aA, g);
        d.ifft(aA);
        S ginv = g.inv(), x = S::one();
        for (size_t i = 0; i < m; ++i) {
            aA[i] = aA[i] * x;
            x = x * ginv;
        }
        aA.push_back(S::zero());
        return aA;
    }
    // prover.hpp:73-158 with (r, s) injected
    void prove(const uint64_t *r_c, const uint64_t *s_c, const uint64_t *omega_c, const uint64_t *coset_c,
               int chunks, Affine<F1> &pA, Affine<F2> &pB, Affine<F1> &pC) const {
        S omega = S::from_canonical(omega_c), g = S::from_canonical(coset_c);
        std::vector<S> H = witness_map(omega, g);
        std::vector<uint64_t> cpa((N + 1) * 4), hc((m - 1) * 4);
        {
            uint64_t one[4] = {1, 0, 0, 0};
            memcpy(cpa.data(), one, 32);
            for (size_t i = 0; i < N; ++i) assignment[i].to_canonical(&cpa[4 * (i + 1)]);
            for (size_t i = 0; i + 1 < m; ++i) H[i].to_canonical(&hc[4 * i]);
        }
        Jac<F1> eA = multiexp<F1>(A_query.data(), cpa.data(), N + 1, chunks);
        // kc_multiexp_with_mixed_addition over the (dense-indexed) B query: zeros skipped, ones added
        Jac<F1> eBh = multiexp<F1>(B_h.data(), cpa.data(), N + 1, chunks);
        Jac<F2> eBg = multiexp<F2>(B_g.data(), cpa.data(), N + 1, chunks);
        Jac<F1> eH = multiexp<F1>(H_query.data(), hc.data(), m - 1, chunks);
        Jac<F1> eL = multiexp<F1>(L_query.data(), cpa.data() + 4 * (n + 1), N - n, chunks);
        uint64_t rs[4];
        (S::from_canonical(r_c) * S::from_canonical(s_c)).to_canonical(rs);
        Jac<F1> d1 = Jac<F1>::from_affine(delta_g1);
        Jac<F1> gA = Jac<F1>::from_affine(alpha_g1).add(eA).add(d1.mul(r_c, 4));
        Jac<F1> gB1 = Jac<F1>::from_affine(beta_g1).add(eBh).add(d1.mul(s_c, 4));
        Jac<F2> gB2 = Jac<F2>::from_affine(beta_g2).add(eBg).add(Jac<F2>::from_affine(delta_g2).mul(s_c, 4));
        Jac<F1> gC = eH.add(eL).add(gA.mul(s_c, 4)).add(gB1.mul(r_c, 4)).add(d1.mul(rs, 4).neg());
        pA = gA.to_affine();
        pB = gB2.to_affine();
        pC = gC.to_affine();
    }
};

// ------------------------------------------------------------------------------------------------
// extern "C" surface for ctypes (tests / bench cpu_baseline only)
// curve: 0 = BLS12-381, 1 = BN254; group: 1 = G1, 2 = G2.  All field elements canonical LE u64 limbs.
// ------------------------------------------------------------------------------------------------
#define DISPATCH_CG(curve, group, ...)     \
    do {                                    \
        if (curve == 0 && group == 1) {     \
            typedef Tr<0, 1> T;             \
            __VA_ARGS__;                    \
        } else if (curve == 0 && group == 2) { \
            typedef Tr<0, 2> T;             \
            __VA_ARGS__;                    \
        } else if (curve == 1 && group == 1) { \
            typedef Tr<1, 1> T;             \
            __VA_ARGS__;                    \
        } else if (curve == 1 && group == 2) { \
            typedef Tr<1, 2> T;             \
            __VA_ARGS__;                    \
        } else                              \
            return -1;                      \
    } while (0)

template <class T>
static std::vector<Affine<typename T::F>> load_bases(const uint64_t *p, const uint8_t *inf, size_t n) {
    std::vector<Affine<typename T::F>> v(n);
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) v[i] = load_affine<typename T::F>(p + i * 2 * T::FL, inf ? inf[i] != 0 : false);
    return v;
}

struct BasesHandle {
    int curve, group;
    size_t n;
    std::shared_ptr<void> data;
};

// one transform with the work of every stage spread over the threads (the batch transform above parallelises over polynomials)
template <class S>
static void ntt_wide(std::vector<S> &a, size_t log_m, const S &omega) {
    const size_t m = (size_t)1 << log_m;
    for (size_t i = 0; i < m; ++i) {
        size_t j = 0;
        for (size_t b = 0; b < log_m; ++b) j |= ((i >> b) & 1) << (log_m - 1 - b);
        if (i < j) std::swap(a[i], a[j]);
    }
    std::vector<S> tw(std::max<size_t>(1, m / 2));
    const size_t blk = 1024;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < (m / 2 + blk - 1) / blk; ++b) {
        S w = omega.pow_u64(b * blk);
        for (size_t i = b * blk; i < std::min(m / 2, (b + 1) * blk); ++i) {
            tw[i] = w;
            w = w * omega;
        }
    }
    for (size_t s = 1; s <= log_m; ++s) {
        const size_t half = (size_t)1 << (s - 1), stride = m >> s;
#pragma omp parallel for schedule(static)
        for (size_t t = 0; t < m / 2; ++t) {
            const size_t k = (t / half) * 2 * half, j = t % half;
            const S x = tw[j * stride] * a[k + j + half], u = a[k + j];
            a[k + j] = u + x;
            a[k + j + half] = u - x;
        }
    }
}

template <class S>
static void poly_mul_t(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, const uint64_t *omega_c, size_t log_m, uint64_t *out) {
    const size_t m = (size_t)1 << log_m, no = na + nb - 1;
    std::vector<S> x(m, S::zero()), y(m, S::zero());
    for (size_t i = 0; i < na; ++i) x[i] = S::from_canonical(a + 4 * i);
    for (size_t i = 0; i < nb; ++i) y[i] = S::from_canonical(b + 4 * i);
    const S omega = S::from_canonical(omega_c);
    ntt_wide<S>(x, log_m, omega);
    ntt_wide<S>(y, log_m, omega);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; ++i) x[i] = x[i] * y[i];
    ntt_wide<S>(x, log_m, omega.inv());
    const S minv = S::from_u64(m).inv();
    for (size_t i = 0; i < no; ++i) (x[i] * minv).to_canonical(out + 4 * i);
}
extern "C" {

int zko_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void zko_set_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t);
#endif
}

// pre-converted resident bases, so the timed baseline does not include canonical->Montgomery conversion
void *zko_bases_new(int curve, int group, const uint64_t *affine, const uint8_t *inf, size_t n) {
    BasesHandle *h = new BasesHandle{curve, group, n, nullptr};
    auto mk = [&](auto tag) {
        typedef decltype(tag) T;
        auto *v = new std::vector<Affine<typename T::F>>(load_bases<T>(affine, inf, n));
        h->data = std::shared_ptr<void>(v, [](void *p) { delete (std::vector<Affine<typename T::F>> *)p; });
    };
    if (curve == 0 && group == 1) mk(Tr<0, 1>());
    else if (curve == 0 && group == 2) mk(Tr<0, 2>());
    else if (curve == 1 && group == 1) mk(Tr<1, 1>());
    else if (curve == 1 && group == 2) mk(Tr<1, 2>());
    else {
        delete h;
        return nullptr;
    }
    return h;
}
void zko_bases_free(void *h) { delete (BasesHandle *)h; }

// algebra::multiexp<BDLO12>(bases[off..off+n), scalars, chunks) -> affine
int zko_msm_bases(void *hv, size_t off, size_t n, const uint64_t *scalars, int chunks, uint64_t *out,
                  uint8_t *out_inf) {
    BasesHandle *h = (BasesHandle *)hv;
    if (!h || off + n > h->n) return -1;
    int curve = h->curve, group = h->group;
    DISPATCH_CG(curve, group, {
        auto *v = (std::vector<Affine<typename T::F>> *)h->data.get();
        auto r = multiexp<typename T::F>(v->data() + off, scalars, n, chunks).to_affine();
        store_affine<typename T::F>(out, out_inf, r);
    });
    return 0;
}

int zko_msm(int curve, int group, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n,
            int chunks, uint64_t *out, uint8_t *out_inf) {
    DISPATCH_CG(curve, group, {
        auto v = load_bases<T>(bases, inf, n);
        auto r = multiexp<typename T::F>(v.data(), scalars, n, chunks).to_affine();
        store_affine<typename T::F>(out, out_inf, r);
    });
    return 0;
}

int zko_msm_naive(int curve, int group, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars,
                  size_t n, uint64_t *out, uint8_t *out_inf) {
    DISPATCH_CG(curve, group, {
        auto v = load_bases<T>(bases, inf, n);
        auto r = msm_naive<typename T::F>(v.data(), scalars, n).to_affine();
        store_affine<typename T::F>(out, out_inf, r);
    });
    return 0;
}

// out[i] = scalars[i] * base  (base == NULL: the standard generator)
int zko_batch_mul(int curve, int group, const uint64_t *base, const uint64_t *scalars, size_t n, uint64_t *out,
                  uint8_t *out_inf) {
    DISPATCH_CG(curve, group, {
        typedef typename T::F F;
        Affine<F> b = load_affine<F>(base ? base : T::gen(), false);
        std::vector<Affine<F>> o(n);
        batch_mul<F>(b, scalars, n, o.data());
        for (size_t i = 0; i < n; ++i) store_affine<F>(out + i * 2 * T::FL, out_inf ? out_inf + i : nullptr, o[i]);
    });
    return 0;
}

// Jacobian (X, Y, Z canonical) -> affine; used by tests to normalise the product's projective output
int zko_jac_to_affine(int curve, int group, const uint64_t *jac, uint64_t *out, uint8_t *out_inf) {
    DISPATCH_CG(curve, group, {
        typedef typename T::F F;
        Jac<F> j = {IO<F>::load(jac), IO<F>::load(jac + T::FL), IO<F>::load(jac + 2 * T::FL)};
        store_affine<F>(out, out_inf, j.to_affine());
    });
    return 0;
}

// a + b on affine points (test helper for linearity properties)
int zko_point_add(int curve, int group, const uint64_t *a, int a_inf, const uint64_t *b, int b_inf, uint64_t *out,
                  uint8_t *out_inf) {
    DISPATCH_CG(curve, group, {
        typedef typename T::F F;
        auto r = Jac<F>::from_affine(load_affine<F>(a, a_inf)).add(Jac<F>::from_affine(load_affine<F>(b, b_inf)));
        store_affine<F>(out, out_inf, r.to_affine());
    });
    return 0;
}

int zko_ntt(int curve, uint64_t *data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
            const uint64_t *coset) {
    if (curve == 0) ntt_batch<FrBLS>(data, log_m, batch, omega, inverse, coset);
    else if (curve == 1) ntt_batch<FrBN>(data, log_m, batch, omega, inverse, coset);
    else return -1;
    return 0;
}

// Fr helpers: out = a*b, a+b, a^e mod r  (tests use them to build large inputs fast)
int zko_fr_mul(int curve, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    if (curve == 0) (FrBLS::from_canonical(a) * FrBLS::from_canonical(b)).to_canonical(out);
    else (FrBN::from_canonical(a) * FrBN::from_canonical(b)).to_canonical(out);
    return 0;
}
// Horner evaluation of a polynomial (coefficients canonical) at x: random-point check of big NTTs
int zko_fr_horner(int curve, const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        int nt = zko_num_threads();
        std::vector<S> part(nt, S::zero());
        S xs = S::from_canonical(x);
        size_t per = (n + nt - 1) / nt;
#pragma omp parallel for
        for (int t = 0; t < nt; ++t) {
            size_t lo = std::min(n, per * t), hi = std::min(n, lo + per);
            S acc = S::zero();
            for (size_t i = hi; i-- > lo;) acc = acc * xs + S::from_canonical(coeffs + 4 * i);
            part[t] = acc * xs.pow_u64(lo);
        }
        S r = S::zero();
        for (auto &p : part) r = r + p;
        r.to_canonical(out);
    };
    if (curve == 0) run(FrBLS());
    else run(FrBN());
    return 0;
}
// seeded uniform Fr elements (SplitMix64 stream: element i uses draws 4i..4i+3 of one stream)
int zko_random_fr(int curve, uint64_t seed, size_t n, uint64_t *out) {
    SplitMix64 rng{seed};
    const uint64_t *mod = curve == 0 ? BLS_R : BN_R;
    for (size_t i = 0; i < n; ++i) rng.next_mod(mod, out + 4 * i);
    return 0;
}

// ---- placeholder's argument arithmetic at sizes the big-integer oracle cannot reach ------------------------------
// Primitives only -- the orchestration (which products, which domains) stays in oracle/cport.py, statement by statement next to
// pyoracle.permutation_argument / lookup_argument, whose dense O(n^2) arithmetic they are pinned to at <= 2^8 rows
// (tests/test_oracle_kat.py).  Polynomials cross as coefficient vectors of canonical Fr.

// out (na + nb - 1 coefficients) = a * b by transforms over 2^log_m >= na + nb - 1 points, omega the primitive 2^log_m-th root
int zko_poly_mul(int curve, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, const uint64_t *omega, size_t log_m, uint64_t *out) {
    if (na == 0 || nb == 0 || ((size_t)1 << log_m) < na + nb - 1) return -1;
    if (curve == 0) poly_mul_t<FrBLS>(a, na, b, nb, omega, log_m, out);
    else if (curve == 1) poly_mul_t<FrBN>(a, na, b, nb, omega, log_m, out);
    else return -1;
    return 0;
}
// one transform of one vector, threads over the butterflies (cport's dfs <-> coefficients at 2^16 .. 2^20)
int zko_ntt_wide(int curve, uint64_t *data, size_t log_m, const uint64_t *omega, int inverse) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        const size_t m = (size_t)1 << log_m;
        std::vector<S> a(m);
        for (size_t i = 0; i < m; ++i) a[i] = S::from_canonical(data + 4 * i);
        S w = S::from_canonical(omega);
        ntt_wide<S>(a, log_m, inverse ? w.inv() : w);
        const S minv = inverse ? S::from_u64(m).inv() : S::one();
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < m; ++i) (a[i] * minv).to_canonical(data + 4 * i);
    };
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}
// pointwise over n elements: op 0 a + b, 1 a - b, 2 a b, 3 a b[0] (scale), 4 a b / c (one inversion per element, as the reference's loops)
int zko_fr_vec(int curve, int op, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t *out, size_t n) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        const S k = op == 3 ? S::from_canonical(b) : S::zero();
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) {
            const S x = S::from_canonical(a + 4 * i);
            S r;
            if (op == 3) r = x * k;
            else {
                const S y = S::from_canonical(b + 4 * i);
                r = op == 0 ? x + y : op == 1 ? x - y : x * y;
                if (op == 4) r = r * S::from_canonical(c + 4 * i).inv();
            }
            r.to_canonical(out + 4 * i);
        }
    };
    if (op < 0 || op > 4) return -1;
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}
// detail::fold_polynomial, DFS form (commitments/detail/polynomial/fold_polynomial.hpp:68-93):
//   out[i] = 1/2 [(1 + alpha w^-i) f[i] + (1 - alpha w^-i) f[i + size/2]],  i < size/2 = 2^(log_size - 1)
// (w^-i by blocks: a power per thread block, then the running product the reference's loop keeps)
int zko_fri_fold(int curve, const uint64_t *f, size_t log_size, const uint64_t *alpha_c, const uint64_t *omega_c, uint64_t *out) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        const size_t half = ((size_t)1 << log_size) / 2;
        const S alpha = S::from_canonical(alpha_c), winv = S::from_canonical(omega_c).inv(), inv2 = S::from_u64(2).inv(), one = S::one();
        const int nt = omp_get_max_threads();
        const size_t per = (half + nt - 1) / nt;
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < nt; ++t) {
            const size_t lo = std::min(half, per * t), hi = std::min(half, lo + per);
            if (lo >= hi) continue;
            S wi = winv.pow_u64(lo);
            for (size_t i = lo; i < hi; ++i) {
                const S a = alpha * wi;
                const S r = inv2 * ((one + a) * S::from_canonical(f + 4 * i) + (one - a) * S::from_canonical(f + 4 * (half + i)));
                r.to_canonical(out + 4 * i);
                wi = wi * winv;
            }
        }
    };
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}
// synthetic division of f (n coefficients) by (X - z): quot (n - 1 coefficients) and the remainder f(z) -- one step of poly_divmod by a
// product of linear factors (kzg_v2.hpp:266-267 `f /= V`, :290-291 `L /= (X - theta_2)`; lpc.hpp's quotients by (X - point))
int zko_poly_div_linear(int curve, const uint64_t *f, size_t n, const uint64_t *z_c, uint64_t *quot, uint64_t *rem) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        const S z = S::from_canonical(z_c);
        S carry = S::zero();
        for (size_t i = n; i-- > 0;) {
            carry = carry * z + S::from_canonical(f + 4 * i);
            if (i > 0) carry.to_canonical(quot + 4 * (i - 1));
        }
        carry.to_canonical(rem);
    };
    if (n == 0) return -1;
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}
// the tests' stand-in for the caller's Merkle tree: (per_leaf + sum_i (i + 1) v_i) mod r over the concatenated leaves
int zko_toy_root(int curve, const uint64_t *leaves, size_t count, uint64_t per_leaf, uint64_t *out) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        const int nt = omp_get_max_threads();
        std::vector<S> part(nt, S::zero());
        const size_t per = (count + nt - 1) / nt;
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < nt; ++t) {
            const size_t lo = std::min(count, per * t), hi = std::min(count, lo + per);
            S acc = S::zero();
            for (size_t i = lo; i < hi; ++i) acc = acc + S::from_u64(i + 1) * S::from_canonical(leaves + 4 * i);
            part[t] = acc;
        }
        S r = S::from_u64(per_leaf);
        for (auto &p : part) r = r + p;
        r.to_canonical(out);
    };
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}
// permutation_argument.hpp:103-136: g_v / h_v (k x n each) and V_P by the row-by-row recurrence, one inversion per row (the inversions
// are independent of the recurrence: taken in parallel, then the serial product)
int zko_perm_grand_product(int curve, size_t k, size_t n, const uint64_t *cols, const uint64_t *sid, const uint64_t *ssig, const uint64_t *beta_c,
                           const uint64_t *gamma_c, uint64_t *out_g, uint64_t *out_h, uint64_t *out_v) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        const S beta = S::from_canonical(beta_c), gamma = S::from_canonical(gamma_c);
        std::vector<S> ratio(n);
#pragma omp parallel for schedule(static)
        for (size_t j = 0; j < n; ++j) {
            S nom = S::one(), den = S::one();
            for (size_t i = 0; i < k; ++i) {
                const S col = S::from_canonical(cols + 4 * (i * n + j));
                const S g = col + beta * S::from_canonical(sid + 4 * (i * n + j)) + gamma, h = col + beta * S::from_canonical(ssig + 4 * (i * n + j)) + gamma;
                g.to_canonical(out_g + 4 * (i * n + j));
                h.to_canonical(out_h + 4 * (i * n + j));
                nom = nom * g;
                den = den * h;
            }
            ratio[j] = nom * den.inv();
        }
        S v = S::one();
        for (size_t j = 0; j < n; ++j) {
            v.to_canonical(out_v + 4 * j);
            v = v * ratio[j];
        }
    };
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}
// compute_V_L, lookup_argument.hpp:375-409: V[0] = 1, V[k] = V[k - 1] g(k - 1) / h(k - 1) for k <= usable_rows, zero behind
int zko_lookup_grand_product(int curve, size_t k_in, size_t k_val, size_t k_sorted, size_t n, size_t usable_rows, const uint64_t *in, const uint64_t *val,
                             const uint64_t *sorted, const uint64_t *beta_c, const uint64_t *gamma_c, uint64_t *out_v) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        const S beta = S::from_canonical(beta_c), gamma = S::from_canonical(gamma_c), opb = S::one() + beta, part1 = opb * gamma;
        S pw = S::one();
        for (size_t i = 0; i < k_in; ++i) pw = pw * opb;
        std::vector<S> ratio(usable_rows);
#pragma omp parallel for schedule(static)
        for (size_t j = 0; j < usable_rows; ++j) {
            const size_t nx = j + 1 == n ? 0 : j + 1;
            S g = pw, h = S::one();
            for (size_t i = 0; i < k_in; ++i) g = g * (gamma + S::from_canonical(in + 4 * (i * n + j)));
            for (size_t i = 0; i < k_val; ++i) g = g * (part1 + S::from_canonical(val + 4 * (i * n + j)) + beta * S::from_canonical(val + 4 * (i * n + nx)));
            for (size_t i = 0; i < k_sorted; ++i) h = h * (part1 + S::from_canonical(sorted + 4 * (i * n + j)) + beta * S::from_canonical(sorted + 4 * (i * n + nx)));
            ratio[j] = g * h.inv();
        }
        S v = S::one();
        memset(out_v, 0, n * 32);
        for (size_t j = 0; j <= usable_rows && j < n; ++j) {
            v.to_canonical(out_v + 4 * j);
            if (j < usable_rows) v = v * ratio[j];
        }
    };
    if (usable_rows >= n) return -1;
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}
// f (len coefficients) = q (X^n - 1) + rem: q[i] = f[i + n] + q[i + n] top down; returns the number of non-zero remainder coefficients
int zko_poly_div_vanishing(int curve, const uint64_t *f, size_t len, size_t n, uint64_t *out_q, uint64_t *nonzero_rem) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        std::vector<S> a(len);
        for (size_t i = 0; i < len; ++i) a[i] = S::from_canonical(f + 4 * i);
        for (size_t i = len; i-- > n;) a[i - n] = a[i - n] + a[i];      // a[i] is final: it is q[i - n]
        for (size_t i = n; i < len; ++i) a[i].to_canonical(out_q + 4 * (i - n));
        uint64_t nz = 0;
        for (size_t i = 0; i < std::min(n, len); ++i) nz += !a[i].is_zero();
        *nonzero_rem = nz;
    };
    if (curve == 0) run(FrBLS());
    else if (curve == 1) run(FrBN());
    else return -1;
    return 0;
}

// ---- Groth16 handle -------------------------------------------------------------------------------
struct G16Handle {
    int curve;
    G16<0> *bls;
    G16<1> *bn;
};
#define G16_DISPATCH(h, ...) \
    do {                      \
        if (h->curve == 0) {  \
            auto *g = h->bls; \
            __VA_ARGS__;      \
        } else {              \
            auto *g = h->bn;  \
            __VA_ARGS__;      \
        }                     \
    } while (0)

void *zko_g16_new(int curve, size_t num_constraints, size_t num_inputs, uint64_t seed) {
    G16Handle *h = new G16Handle{curve, nullptr, nullptr};
    if (curve == 0) {
        h->bls = new G16<0>();
        h->bls->make_example(num_constraints, num_inputs, seed);
    } else if (curve == 1) {
        h->bn = new G16<1>();
        h->bn->make_example(num_constraints, num_inputs, seed);
    } else {
        delete h;
        return nullptr;
    }
    return h;
}
void zko_g16_free(void *hv) {
    G16Handle *h = (G16Handle *)hv;
    delete h->bls;
    delete h->bn;
    delete h;
}
// dims: M, n, N, m, log_m, nnzA, nnzB, nnzC
int zko_g16_dims(void *hv, uint64_t *d) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, {
        d[0] = g->M;
        d[1] = g->n;
        d[2] = g->N;
        d[3] = g->m;
        d[4] = g->log_m;
        d[5] = g->A.col.size();
        d[6] = g->B.col.size();
        d[7] = g->C.col.size();
    });
    return 0;
}
// kind: 0 basic, 1 extended, 2 step radix-2; omega / shift as pyoracle.EvaluationDomain takes them (shift nullable)
int zko_g16_set_domain(void *hv, int kind, size_t m, const uint64_t *omega, const uint64_t *shift) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, {
        typedef typename std::remove_reference<decltype(*g)>::type GT;
        typedef typename GT::S S;
        if (m < g->M + g->n + 1) return -1;
        g->set_domain(kind, m, S::from_canonical(omega), shift ? S::from_canonical(shift) : S::zero());
    });
    return 0;
}
// what make_evaluation_domain(min_size) picks over a field of the given two-adicity: out = {kind, m}; -1: none of the radix-2 family
int zko_domain_choice(size_t min_size, size_t two_adicity, uint64_t *out) {
    int kind;
    size_t m;
    if (!domain_choice(min_size, two_adicity, kind, m)) return -1;
    out[0] = (uint64_t)kind;
    out[1] = m;
    return 0;
}
// in-place transform of one vector of m elements over a domain (for the tests of the device transforms)
int zko_domain_fft(int curve, int kind, size_t m, const uint64_t *omega, const uint64_t *shift, uint64_t *data, int inverse) {
    auto run = [&](auto tag) {
        typedef decltype(tag) S;
        Domain<S> d = Domain<S>::make(kind, m, S::from_canonical(omega), shift ? S::from_canonical(shift) : S::zero());
        std::vector<S> a(m);
        for (size_t i = 0; i < m; ++i) a[i] = S::from_canonical(data + 4 * i);
        if (inverse) d.ifft(a);
        else d.fft(a);
        for (size_t i = 0; i < m; ++i) a[i].to_canonical(data + 4 * i);
    };
    if (curve == 0) run(typename Tr<0, 1>::S());
    else if (curve == 1) run(typename Tr<1, 1>::S());
    else return -1;
    return 0;
}
int zko_g16_is_satisfied(void *hv) {
    G16Handle *h = (G16Handle *)hv;
    int r = 0;
    G16_DISPATCH(h, r = g->is_satisfied() ? 1 : 0);
    return r;
}
int zko_g16_keygen(void *hv, const uint64_t *trapdoor, const uint64_t *omega) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, g->keygen(trapdoor, omega));
    return 0;
}
int zko_g16_expected_exponents(void *hv, const uint64_t *trapdoor, const uint64_t *omega, const uint64_t *r, const uint64_t *s,
                               uint64_t *out) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, g->expected_exponents(trapdoor, omega, r, s, out));
    return 0;
}
// which: 0 = A, 1 = B, 2 = C (after swap_AB if keygen ran)
int zko_g16_get_csr(void *hv, int which, uint32_t *rowptr, uint32_t *col, uint64_t *coeff) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, {
        auto &c = which == 0 ? g->A : which == 1 ? g->B : g->C;
        memcpy(rowptr, c.rowptr.data(), c.rowptr.size() * 4);
        memcpy(col, c.col.data(), c.col.size() * 4);
        for (size_t i = 0; i < c.coeff.size(); ++i) c.coeff[i].to_canonical(coeff + 4 * i);
    });
    return 0;
}
int zko_g16_get_assignment(void *hv, uint64_t *out) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, {
        for (size_t i = 0; i < g->N; ++i) g->assignment[i].to_canonical(out + 4 * i);
    });
    return 0;
}
// which: 0 = A_query (G1, N+1), 1 = B_query.h (G1, N+1), 2 = B_query.g (G2, N+1), 3 = H_query (G1, m-1),
//        4 = L_query (G1, N-n), 5 = {alpha_g1, beta_g1, delta_g1} (G1, 3), 6 = {beta_g2, delta_g2} (G2, 2)
int zko_g16_get_query(void *hv, int which, uint64_t *out, uint8_t *inf) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, {
        typedef typename std::remove_reference<decltype(*g)>::type GT;
        typedef typename GT::F1 F1;
        typedef typename GT::F2 F2;
        auto put1 = [&](const std::vector<Affine<F1>> &v) {
            for (size_t i = 0; i < v.size(); ++i) store_affine<F1>(out + i * 2 * IO<F1>::L, inf + i, v[i]);
        };
        auto put2 = [&](const std::vector<Affine<F2>> &v) {
            for (size_t i = 0; i < v.size(); ++i) store_affine<F2>(out + i * 2 * IO<F2>::L, inf + i, v[i]);
        };
        switch (which) {
            case 0: put1(g->A_query); break;
            case 1: put1(g->B_h); break;
            case 2: put2(g->B_g); break;
            case 3: put1(g->H_query); break;
            case 4: put1(g->L_query); break;
            case 5: put1({g->alpha_g1, g->beta_g1, g->delta_g1}); break;
            case 6: put2({g->beta_g2, g->delta_g2}); break;
            default: return -1;
        }
    });
    return 0;
}
// coefficients_for_H, m+1 elements
int zko_g16_witness_map(void *hv, const uint64_t *omega, const uint64_t *coset, uint64_t *out) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, {
        typedef typename std::remove_reference<decltype(*g)>::type GT;
        typedef typename GT::S S;
        auto H = g->witness_map(S::from_canonical(omega), S::from_canonical(coset));
        for (size_t i = 0; i < H.size(); ++i) H[i].to_canonical(out + 4 * i);
    });
    return 0;
}
// proof = A (G1 affine) | B (G2 affine) | C (G1 affine), canonical; returns 0
int zko_g16_prove(void *hv, const uint64_t *r, const uint64_t *s, const uint64_t *omega, const uint64_t *coset,
                  int chunks, uint64_t *proof) {
    G16Handle *h = (G16Handle *)hv;
    G16_DISPATCH(h, {
        typedef typename std::remove_reference<decltype(*g)>::type GT;
        typedef typename GT::F1 F1;
        typedef typename GT::F2 F2;
        Affine<F1> A, C;
        Affine<F2> B;
        g->prove(r, s, omega, coset, chunks, A, B, C);
        uint8_t inf;
        store_affine<F1>(proof, &inf, A);
        store_affine<F2>(proof + 2 * IO<F1>::L, &inf, B);
        store_affine<F1>(proof + 2 * IO<F1>::L + 2 * IO<F2>::L, &inf, C);
    });
    return 0;
}

}  // extern "C"
