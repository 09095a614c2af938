"""TEST INFRASTRUCTURE ONLY -- big-integer oracle for the MSM / NTT / Groth16 hot path.

This module is *the checker*, never the product.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import anything under oracle/.

It restates, with plain Python integers, the algorithms the reference
(NilFoundation/crypto3-zk) reaches through its un-vendored dependencies
crypto3-algebra / crypto3-math (no version pin exists; the only pin-like datum is the
CI ref NilFoundation/crypto3@1bd56b12f410f3f1a4891076705a9261a6b1efaa,
.github/workflows/pull-request.yml:29):

  * algebra::multiexp                       -> msm_naive / msm_pippenger
      call sites: zk/snark/systems/ppzksnark/r1cs_gg_ppzksnark/prover.hpp:108-139,
                  zk/commitments/polynomial/kzg.hpp:143-148,409-435
  * kc_multiexp_with_mixed_addition         -> kc_multiexp
      zk/commitments/polynomial/knowledge_commitment_multiexp.hpp:57-108
  * evaluation_domain::fft / inverse_fft    -> ntt / intt (definition: out[i] = sum_j in[j] w^(ij))
      call sites: zk/snark/reductions/r1cs_to_qap.hpp:250-310
  * r1cs_to_qap::witness_map                -> witness_map   (r1cs_to_qap.hpp:219-325)
  * r1cs_to_qap::instance_map_with_evaluation -> qap_evaluate_at (r1cs_to_qap.hpp:138-187)
  * r1cs_gg_ppzksnark_generator (fixed trapdoor) -> groth16_keygen (generator.hpp:86-236)
  * r1cs_gg_ppzksnark_prover::process       -> groth16_prove (prover.hpp:73-158)
  * ipp2 prove_commitment_{v,w} (the KAT carrier) -> ipp2_prove_commitment_{v,w}
      zk/snark/systems/ppzksnark/r1cs_gg_ppzksnark/ipp2/prover.hpp:99-290, ipp2/srs.hpp:44-56
  * kzg_commitment_scheme::proof_eval (v1)  -> kzg_v1_proof_eval   zk/commitments/polynomial/kzg.hpp:782-807
  * kzg_commitment_scheme_v2::proof_eval    -> kzg_v2_proof_eval (+ get_U / get_V / set_difference_polynom:
      lagrange_interpolation, vanishing_poly)
      zk/commitments/polynomial/kzg_v2.hpp:121-148, 236-305; zk/commitments/batched_commitment.hpp:79-111, 168-183

Parity pin: the bellperson-derived known-answer vectors of
test/systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark_aggregation_conformity.cpp
(:578-862 Fr chains, :864-930 G1/G2 MSM) and test/commitment/kzg.cpp:75-103 are reproduced by
tests/test_oracle_kat.py from tests/golden/ref_kat.json.  NTT outputs, full Groth16 proofs and KZG
opening proofs are NOT pinned by any reference test ("parity unpinned" there): the NTT is pinned to
the mathematical DFT definition, Groth16 proofs to the trapdoor ("in the exponent") identities, the
opening proofs to the reference's own BOOST_ASSERTs (f divisible by V, L(theta_2) = 0), to the
identities of kzg_basic_test and to the verifier's equation evaluated with alpha in the clear.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

MASK64 = (1 << 64) - 1


# --------------------------------------------------------------------------------------
# deterministic PRNG shared with oracle/zk_oracle.cpp and the HIP library's test helpers
# --------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def next_mod(self, modulus: int) -> int:
        """4 little-endian 64-bit limbs reduced mod `modulus` (same rule as the C++ oracle)."""
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % modulus


# --------------------------------------------------------------------------------------
# fields
# --------------------------------------------------------------------------------------
class Fq:
    """Prime field ops on plain ints."""

    def __init__(self, p: int):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def neg(self, a):
        return (-a) % self.p

    def mul(self, a, b):
        return (a * b) % self.p

    def sqr(self, a):
        return (a * a) % self.p

    def inv(self, a):
        return pow(a, -1, self.p)

    def is_zero(self, a):
        return a % self.p == 0

    def eq(self, a, b):
        return (a - b) % self.p == 0

    def from_int(self, k):
        return k % self.p


class Fq2:
    """Fq[u]/(u^2+1) on (c0, c1) tuples (both BLS12-381 and BN254 use u^2 = -1)."""

    def __init__(self, p: int):
        self.p = p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b):
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def neg(self, a):
        return ((-a[0]) % self.p, (-a[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a):
        return self.mul(a, a)

    def inv(self, a):
        p = self.p
        n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return (a[0] * n % p, (-a[1]) * n % p)

    def is_zero(self, a):
        return a[0] % self.p == 0 and a[1] % self.p == 0

    def eq(self, a, b):
        return self.is_zero(self.sub(a, b))

    def from_int(self, k):
        return (k % self.p, 0)


# --------------------------------------------------------------------------------------
# short Weierstrass y^2 = x^3 + b, Jacobian coordinates; affine None = infinity
# --------------------------------------------------------------------------------------
class Group:
    def __init__(self, F, b, gen, order: int, name: str):
        self.F = F
        self.b = b
        self.gen = gen  # affine (x, y)
        self.order = order
        self.name = name

    # ---- affine helpers
    def on_curve(self, P) -> bool:
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.eq(F.sqr(y), F.add(F.mul(F.sqr(x), x), self.b))

    def neg(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]))

    # ---- Jacobian
    def to_jac(self, P):
        if P is None:
            return (self.F.one, self.F.one, self.F.zero)
        return (P[0], P[1], self.F.one)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdbl(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        t = F.sub(F.sub(F.sqr(F.add(X, B)), A), C)
        D = F.add(t, t)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def jadd(self, P, Q):
        F = self.F
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        if F.is_zero(Z1):
            return Q
        if F.is_zero(Z2):
            return P
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        if F.eq(U1, U2):
            if F.eq(S1, S2):
                return self.jdbl(P)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        R = F.sub(S2, S1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(R), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(R, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def jmul(self, J, k: int):
        F = self.F
        R = (F.one, F.one, F.zero)
        if k < 0:
            J = (J[0], F.neg(J[1]), J[2])
            k = -k
        for bit in bin(k)[2:] if k else "":
            R = self.jdbl(R)
            if bit == "1":
                R = self.jadd(R, J)
        return R

    # ---- convenience on affine points
    def mul(self, P, k: int):
        return self.to_affine(self.jmul(self.to_jac(P), k % self.order))

    def add(self, P, Q):
        return self.to_affine(self.jadd(self.to_jac(P), self.to_jac(Q)))

    def batch_mul_gen(self, ks: Sequence[int]):
        """[k]G for many k, with a fixed-base 8-bit window table (keeps fixture generation fast)."""
        F = self.F
        W = 8
        nwin = (self.order.bit_length() + W - 1) // W
        table = []
        base = self.to_jac(self.gen)
        for _ in range(nwin):
            row = [(F.one, F.one, F.zero)]
            for i in range(1, 1 << W):
                row.append(self.jadd(row[-1], base))
            table.append(row)
            for _ in range(W):
                base = self.jdbl(base)
        out = []
        for k in ks:
            k %= self.order
            acc = (F.one, F.one, F.zero)
            w = 0
            while k:
                d = k & ((1 << W) - 1)
                if d:
                    acc = self.jadd(acc, table[w][d])
                k >>= W
                w += 1
            out.append(self.to_affine(acc))
        return out


@dataclass
class Curve:
    name: str
    p: int
    r: int
    g1: Group
    g2: Group
    fr_generator: int  # multiplicative generator of Fr* (crypto3 arithmetic_params convention)
    two_adicity: int

    def root_of_unity(self, log_m: int) -> int:
        """w = g^((r-1)/2^log_m): the conventional 2^log_m-th primitive root.  The reference's choice
        lives in crypto3-algebra (not in tree); the C ABI therefore takes w as an argument."""
        assert log_m <= self.two_adicity
        return pow(self.fr_generator, (self.r - 1) >> log_m, self.r)


def _make_bls12_381() -> Curve:
    p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    g1 = (
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    )
    g2 = (
        (
            0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
        ),
        (
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
        ),
    )
    return Curve(
        "bls12_381",
        p,
        r,
        Group(Fq(p), 4, g1, r, "bls12_381_g1"),
        Group(Fq2(p), (4, 4), g2, r, "bls12_381_g2"),
        7,
        32,
    )


def _make_bn254() -> Curve:
    p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    g2 = (
        (
            10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
        ),
        (
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531,
        ),
    )
    f2 = Fq2(p)
    b2 = f2.mul((3, 0), f2.inv((9, 1)))  # 3/(9+u)
    return Curve(
        "bn254",
        p,
        r,
        Group(Fq(p), 3, (1, 2), r, "bn254_g1"),
        Group(f2, b2, g2, r, "bn254_g2"),
        5,
        28,
    )


BLS12_381 = _make_bls12_381()
BN254 = _make_bn254()
CURVES = {"bls12_381": BLS12_381, "bn254": BN254}


# --------------------------------------------------------------------------------------
# MSM
# --------------------------------------------------------------------------------------
def msm_naive(G: Group, bases: Sequence, scalars: Sequence[int]):
    """sum_i s_i * P_i by double-and-add; the definition every MSM algorithm must hit."""
    F = G.F
    acc = (F.one, F.one, F.zero)
    for P, s in zip(bases, scalars):
        if P is None or s % G.order == 0:
            continue
        acc = G.jadd(acc, G.jmul(G.to_jac(P), s % G.order))
    return G.to_affine(acc)


def msm_pippenger(G: Group, bases: Sequence, scalars: Sequence[int], c: Optional[int] = None):
    """Bucket method (BDLO12 as in the libff lineage of algebra::multiexp): unsigned c-bit digits,
    windows from the top, per-window bucket accumulation and running-sum reduction."""
    F = G.F
    n = len(bases)
    if n == 0:
        return None
    if c is None:
        lg = max(1, n.bit_length() - 1)
        c = max(1, lg - (lg // 3 - 2)) if lg >= 6 else max(1, lg)
    nbits = max((s % G.order).bit_length() for s in scalars) if scalars else 0
    groups = (nbits + c - 1) // c
    inf = (F.one, F.one, F.zero)
    result = inf
    for k in range(groups - 1, -1, -1):
        for _ in range(c):
            result = G.jdbl(result)
        buckets = [inf] * (1 << c)
        for P, s in zip(bases, scalars):
            if P is None:
                continue
            d = ((s % G.order) >> (k * c)) & ((1 << c) - 1)
            if d:
                buckets[d] = G.jadd(buckets[d], G.to_jac(P))
        running = inf
        for i in range((1 << c) - 1, 0, -1):
            running = G.jadd(running, buckets[i])
            result = G.jadd(result, running)
    return G.to_affine(result)


def kc_multiexp(G2g: Group, G1g: Group, indices, values, min_idx, max_idx, scalars):
    """knowledge_commitment_multiexp.hpp:57-108 -- sparse vector of (g in G2, h in G1) pairs:
    skip 0, add 1-scalars directly, MSM on the rest.  Returns (g_sum, h_sum) affine."""
    import bisect

    off = bisect.bisect_left(indices, min_idx)
    acc_g = None
    acc_h = None
    p = []
    gg = []
    hh = []
    r = G1g.order
    for pos in range(off, len(indices)):
        idx = indices[pos]
        if idx >= max_idx:
            break
        s = scalars[idx - min_idx] % r
        g, h = values[pos]
        if s == 0:
            continue
        if s == 1:
            acc_g = G2g.add(acc_g, g)
            acc_h = G1g.add(acc_h, h)
        else:
            p.append(s)
            gg.append(g)
            hh.append(h)
    return (G2g.add(acc_g, msm_naive(G2g, gg, p)), G1g.add(acc_h, msm_naive(G1g, hh, p)))


# --------------------------------------------------------------------------------------
# NTT
# --------------------------------------------------------------------------------------
def dft_naive(a: Sequence[int], w: int, r: int) -> List[int]:
    n = len(a)
    out = []
    for i in range(n):
        wi = pow(w, i, r)
        acc = 0
        x = 1
        for j in range(n):
            acc = (acc + a[j] * x) % r
            x = x * wi % r
        out.append(acc)
    return out


def ntt(a: Sequence[int], w: int, r: int) -> List[int]:
    """Natural order in, natural order out: out[i] = sum_j a[j] w^(ij).  Iterative radix-2 DIT
    (bit-reverse then butterflies), the classic in-place form of evaluation_domain::fft."""
    n = len(a)
    lg = n.bit_length() - 1
    assert 1 << lg == n
    a = list(a)
    for i in range(n):
        j = int(bin(i)[2:].zfill(lg)[::-1], 2) if lg else 0
        if i < j:
            a[i], a[j] = a[j], a[i]
    m = 1
    while m < n:
        wm = pow(w, n // (2 * m), r)
        for k in range(0, n, 2 * m):
            x = 1
            for j in range(m):
                t = x * a[k + j + m] % r
                u = a[k + j]
                a[k + j] = (u + t) % r
                a[k + j + m] = (u - t) % r
                x = x * wm % r
        m *= 2
    return a


def intt(a: Sequence[int], w: int, r: int) -> List[int]:
    n = len(a)
    ninv = pow(n, -1, r)
    return [x * ninv % r for x in ntt(a, pow(w, -1, r), r)]


def multiply_by_coset(a: Sequence[int], g: int, r: int) -> List[int]:
    """math::multiply_by_coset: a[i] *= g^i (r1cs_to_qap.hpp:266-269)."""
    out = []
    x = 1
    for v in a:
        out.append(v * x % r)
        x = x * g % r
    return out


def stockham_model(a: Sequence[int], w: int, r: int, radices: Sequence[int]) -> List[int]:
    """Index-exact model of the multi-pass Stockham decomposition used by the HIP kernels
    (crypto3-zk_amd/csrc/ntt.hip): pass with radix R and running sub-transform size Ns:
        u[t]  = x[j + t*m/R] * w_{Ns*R}^(k t),  k = j mod Ns
        v     = DFT_R(u)
        y[(j-k)*R + k + t'*Ns] = v[t']
    Used by the CPU tests to pin the index math against dft_naive before any GPU run."""
    m = len(a)
    x = list(a)
    Ns = 1
    for R in radices:
        y = [0] * m
        wsub = pow(w, m // (Ns * R), r)  # primitive (Ns*R)-th root
        wR = pow(w, m // R, r)
        for j in range(m // R):
            k = j % Ns
            u = [x[j + t * (m // R)] * pow(wsub, k * t, r) % r for t in range(R)]
            v = dft_naive(u, wR, r)
            base = (j - k) * R + k
            for t in range(R):
                y[base + t * Ns] = v[t]
        x = y
        Ns *= R
    assert Ns == m
    return x


# --------------------------------------------------------------------------------------
# Evaluation domains (crypto3-math `evaluation_domain` and `make_evaluation_domain`)
#
# crypto3-math is NOT in /root/reference (un-vendored dependency, no version pin).  What is restated here is
# the published algorithm of the library it descends from -- libfqfft's `get_evaluation_domain`, `basic_radix2_domain`,
# `extended_radix2_domain`, `step_radix2_domain` (SCIPR-lab libfqfft, evaluation_domain/{get_evaluation_domain.tcc,
# domains/*.tcc}), whose selection order and member names crypto3-math keeps -- anchored on the reference's call sites:
#   reductions/r1cs_to_qap.hpp:229-230, 138-139      make_evaluation_domain(num_constraints + num_inputs + 1)
#   reductions/r1cs_to_qap.hpp:150-153, 250-315      compute_vanishing_polynomial, evaluate_all_lagrange_polynomials,
#                                                    inverse_fft, fft, add_poly_z, divide_by_z_on_coset
# PIN: every method is held to its DEFINITION over the domain's point set by tests/test_oracle_kat.py
# (fft = evaluation at get_domain_element(i), inverse_fft its inverse, Lagrange / vanishing polynomials by their
# products, divide_by_z_on_coset = division by Z(g x_i)); the reference asserts no transform output ("parity unpinned").
# Geometric / arithmetic sequence domains (the selection's last resort, reached only beyond 2^two_adicity points) are
# out of scope: make_evaluation_domain raises there.
# --------------------------------------------------------------------------------------
def _ceil_log2(n: int) -> int:
    """libff::log2 / std::ceil(std::log2(n))"""
    return 0 if n <= 1 else (n - 1).bit_length()


class EvaluationDomain:
    BASIC, EXTENDED, STEP = 0, 1, 2
    KIND_NAMES = {0: "basic_radix2", 1: "extended_radix2", 2: "step_radix2"}

    def __init__(self, kind: int, m: int, omega: int, r: int, shift: int = 0):
        """basic: omega = primitive m-th root.  extended: omega = primitive (m/2)-th root, the points are <omega> and
        shift <omega>.  step: big_m = 2^(ceil(log2 m) - 1), small_m = m - big_m (a power of two); omega = primitive
        (2 big_m)-th root; the points are <omega^2> followed by omega <omega^(2 big_m / small_m)>."""
        self.kind, self.m, self.omega, self.r, self.shift = kind, m, omega % r, r, shift % r
        if kind == self.BASIC:
            assert m > 1 and m & (m - 1) == 0
            assert pow(omega, m, r) == 1 and pow(omega, m // 2, r) != 1
        elif kind == self.EXTENDED:
            assert m > 1 and m & (m - 1) == 0
            self.small_m = m // 2
            assert pow(omega, self.small_m, r) == 1 and (self.small_m == 1 or pow(omega, self.small_m // 2, r) != 1)
        else:
            assert m > 1
            self.big_m = 1 << (_ceil_log2(m) - 1)
            self.small_m = m - self.big_m
            assert self.small_m == 1 << _ceil_log2(self.small_m), "step_radix2(): expected small_m == 1ul<<log2(small_m)"
            assert pow(omega, 2 * self.big_m, r) == 1 and pow(omega, self.big_m, r) != 1
            self.big_omega = omega * omega % r
            self.compr = self.big_m // self.small_m
            self.small_omega = pow(omega, 2 * self.compr, r)  # = unity_root(small_m) on the same root chain

    # ---- the point set
    def get_domain_element(self, idx: int) -> int:
        r = self.r
        if self.kind == self.BASIC:
            return pow(self.omega, idx, r)
        if self.kind == self.EXTENDED:
            return pow(self.omega, idx, r) if idx < self.small_m else self.shift * pow(self.omega, idx - self.small_m, r) % r
        if idx < self.big_m:
            return pow(self.big_omega, idx, r)
        return self.omega * pow(self.small_omega, idx - self.big_m, r) % r

    def elements(self) -> List[int]:
        return [self.get_domain_element(i) for i in range(self.m)]

    # ---- transforms
    def fft(self, a: Sequence[int]) -> List[int]:
        r = self.r
        assert len(a) == self.m
        if self.kind == self.BASIC:
            return ntt(a, self.omega, r)
        if self.kind == self.EXTENDED:
            sm, sh = self.small_m, self.shift
            s_to_sm = pow(sh, sm, r)
            a0 = [(a[i] + a[sm + i]) % r for i in range(sm)]
            a1 = [pow(sh, i, r) * (a[i] + s_to_sm * a[sm + i]) % r for i in range(sm)]
            return ntt(a0, self.omega, r) + ntt(a1, self.omega, r)
        bm, sm, w = self.big_m, self.small_m, self.omega
        c = [(a[i] + a[i + bm]) % r if i < sm else a[i] % r for i in range(bm)]
        d = []
        wi = 1
        for i in range(bm):
            d.append(wi * ((a[i] - a[i + bm]) if i < sm else a[i]) % r)
            wi = wi * w % r
        e = [sum(d[i + j * sm] for j in range(self.compr)) % r for i in range(sm)]
        return ntt(c, self.big_omega, r) + ntt(e, self.small_omega, r)

    def inverse_fft(self, a: Sequence[int]) -> List[int]:
        r = self.r
        assert len(a) == self.m
        if self.kind == self.BASIC:
            return intt(a, self.omega, r)
        if self.kind == self.EXTENDED:
            sm, sh = self.small_m, self.shift
            winv = pow(self.omega, -1, r)
            a0 = ntt(a[:sm], winv, r)
            a1 = ntt(a[sm:], winv, r)
            s_to_sm = pow(sh, sm, r)
            sconst = pow(sm * (1 - s_to_sm) % r, -1, r)
            shinv = pow(sh, -1, r)
            lo, hi = [], []
            x = 1
            for i in range(sm):
                lo.append(sconst * (-s_to_sm * a0[i] + x * a1[i]) % r)
                hi.append(sconst * (a0[i] - x * a1[i]) % r)
                x = x * shinv % r
            return lo + hi
        bm, sm, w = self.big_m, self.small_m, self.omega
        U0 = intt(a[:bm], self.big_omega, r)
        U1 = intt(a[bm:], self.small_omega, r)
        tmp = []
        wi = 1
        for i in range(bm):
            tmp.append(U0[i] * wi % r)
            wi = wi * w % r
        out = [0] * self.m
        for i in range(sm, bm):
            out[i] = U0[i]
        winv = pow(w, -1, r)
        half = pow(2, -1, r)
        x = 1
        for i in range(sm):
            u1 = (U1[i] - sum(tmp[i + j * sm] for j in range(1, self.compr))) % r * x % r
            out[i] = (U0[i] + u1) * half % r
            out[bm + i] = (U0[i] - u1) * half % r
            x = x * winv % r
        return out

    # ---- what the QAP reduction reads (r1cs_to_qap.hpp:150-153, 261, 308)
    def compute_vanishing_polynomial(self, t: int) -> int:
        r = self.r
        if self.kind == self.BASIC:
            return (pow(t, self.m, r) - 1) % r
        if self.kind == self.EXTENDED:
            tm = pow(t, self.small_m, r)
            return (tm - 1) * (tm - pow(self.shift, self.small_m, r)) % r
        return (pow(t, self.big_m, r) - 1) * (pow(t, self.small_m, r) - pow(self.omega, self.small_m, r)) % r

    def evaluate_all_lagrange_polynomials(self, t: int) -> List[int]:
        r = self.r
        if self.kind == self.BASIC:
            return lagrange_at(self.m, self.omega, t, r)
        if self.kind == self.EXTENDED:
            sm, sh = self.small_m, self.shift
            T0 = lagrange_at(sm, self.omega, t, r)
            T1 = lagrange_at(sm, self.omega, t * pow(sh, -1, r) % r, r)
            t_sm, s_sm = pow(t, sm, r), pow(sh, sm, r)
            ood = pow((s_sm - 1) % r, -1, r)
            c0 = (t_sm - s_sm) * (-ood) % r
            c1 = (t_sm - 1) * ood % r
            return [x * c0 % r for x in T0] + [x * c1 % r for x in T1]
        bm, sm, w = self.big_m, self.small_m, self.omega
        inner_big = lagrange_at(bm, self.big_omega, t, r)
        inner_small = lagrange_at(sm, self.small_omega, t * pow(w, -1, r) % r, r)
        L0 = (pow(t, sm, r) - pow(w, sm, r)) % r
        w_sm = pow(w, sm, r)
        bw_sm = pow(self.big_omega, sm, r)
        out = []
        elt = 1
        for i in range(bm):
            out.append(inner_big[i] * L0 % r * pow((elt - w_sm) % r, -1, r) % r)
            elt = elt * bw_sm % r
        L1 = (pow(t, bm, r) - 1) * pow((pow(w, bm, r) - 1) % r, -1, r) % r
        return out + [L1 * x % r for x in inner_small]

    def add_poly_z(self, coeff: int, H: List[int]) -> None:
        r = self.r
        assert len(H) == self.m + 1
        if self.kind == self.BASIC:
            H[self.m] = (H[self.m] + coeff) % r
            H[0] = (H[0] - coeff) % r
        elif self.kind == self.EXTENDED:
            s_sm = pow(self.shift, self.small_m, r)
            H[2 * self.small_m] = (H[2 * self.small_m] + coeff) % r
            H[self.small_m] = (H[self.small_m] - coeff * (s_sm + 1)) % r
            H[0] = (H[0] + coeff * s_sm) % r
        else:
            w_sm = pow(self.omega, self.small_m, r)
            H[self.m] = (H[self.m] + coeff) % r
            H[self.big_m] = (H[self.big_m] - coeff * w_sm) % r
            H[self.small_m] = (H[self.small_m] - coeff) % r
            H[0] = (H[0] + coeff * w_sm) % r

    def divide_by_z_on_coset(self, P: Sequence[int], g: int) -> List[int]:
        """P[i] / Z(g x_i): P holds evaluations on the coset g * domain (multiply_by_coset(g) followed by fft)."""
        r = self.r
        if self.kind == self.BASIC:
            zi = pow(self.compute_vanishing_polynomial(g), -1, r)
            return [x * zi % r for x in P]
        if self.kind == self.EXTENDED:
            z0 = pow(self.compute_vanishing_polynomial(g), -1, r)
            z1 = pow(self.compute_vanishing_polynomial(g * self.shift % r), -1, r)
            return [x * z0 % r for x in P[: self.small_m]] + [x * z1 % r for x in P[self.small_m :]]
        bm, sm, w = self.big_m, self.small_m, self.omega
        Z0 = (pow(g, bm, r) - 1) % r
        c_sm_Z0 = pow(g, sm, r) * Z0 % r
        w_sm_Z0 = pow(w, sm, r) * Z0 % r
        w_2sm = pow(w, 2 * sm, r)
        out = []
        elt = 1
        for i in range(bm):
            out.append(P[i] * pow((c_sm_Z0 * elt - w_sm_Z0) % r, -1, r) % r)
            elt = elt * w_2sm % r
        gw = g * w % r
        Z1 = (pow(gw, bm, r) - 1) * (pow(gw, sm, r) - pow(w, sm, r)) % r
        z1i = pow(Z1, -1, r)
        return out + [x * z1i % r for x in P[bm:]]

    def describe(self) -> str:
        return "%s(m=%d)" % (self.KIND_NAMES[self.kind], self.m)


def evaluation_domain_choice(min_size: int, two_adicity: int) -> Tuple[int, int]:
    """(kind, m) that make_evaluation_domain(min_size) returns over a field of the given two-adicity: the selection order of
    libfqfft's get_evaluation_domain -- basic, extended, step at min_size, then the same three at big + rounded_small."""
    s = two_adicity

    def basic_ok(n):
        return n > 1 and n == 1 << _ceil_log2(n) and _ceil_log2(n) <= s

    def extended_ok(n):
        return n > 1 and _ceil_log2(n) == s + 1 and n == 1 << (s + 1)

    def step_ok(n):
        if n <= 1:
            return False
        small = n - (1 << (_ceil_log2(n) - 1))
        return small == 1 << _ceil_log2(small) and _ceil_log2(n) <= s

    big = 1 << (_ceil_log2(min_size) - 1) if min_size > 1 else 0
    small = min_size - big
    rounded = big + (1 << _ceil_log2(small))
    for n in (min_size, rounded):
        for kind, ok in ((EvaluationDomain.BASIC, basic_ok), (EvaluationDomain.EXTENDED, extended_ok), (EvaluationDomain.STEP, step_ok)):
            if ok(n):
                return kind, n
    raise ValueError("make_evaluation_domain(%d): only the radix-2 family is restated (geometric / arithmetic sequence domains are out of scope)" % min_size)


def make_evaluation_domain(curve: "Curve", min_size: int, two_adicity: Optional[int] = None, root=None) -> EvaluationDomain:
    """math::make_evaluation_domain<Fr>(min_size).  `two_adicity` / `root` (log -> primitive 2^log-th root) let the tests
    exercise the extended domain -- which a real field only reaches at 2^(s+1) points -- over a pretended smaller s."""
    s = curve.two_adicity if two_adicity is None else two_adicity
    root = root or curve.root_of_unity
    kind, m = evaluation_domain_choice(min_size, s)
    if kind == EvaluationDomain.BASIC:
        return EvaluationDomain(kind, m, root(_ceil_log2(m)), curve.r)
    if kind == EvaluationDomain.EXTENDED:
        # detail::coset_shift<F>() = multiplicative_generator^2
        return EvaluationDomain(kind, m, root(_ceil_log2(m) - 1), curve.r, pow(curve.fr_generator, 2, curve.r))
    return EvaluationDomain(kind, m, root(_ceil_log2(m)), curve.r)


# --------------------------------------------------------------------------------------
# R1CS / QAP / Groth16
# --------------------------------------------------------------------------------------
@dataclass
class R1CS:
    """constraints[i] = (a, b, c), each a list of (variable_index, coeff); index 0 = constant 1.
    r1cs.hpp:61-64,125-133."""

    num_inputs: int
    num_aux: int
    constraints: List[Tuple[list, list, list]]

    @property
    def num_variables(self):
        return self.num_inputs + self.num_aux

    @property
    def num_constraints(self):
        return len(self.constraints)


def r1cs_example_field_input(r: int, num_constraints: int, num_inputs: int, seed: int):
    """test/systems/ppzksnark/r1cs_examples.hpp:77-140 with a, b drawn from SplitMix64(seed)."""
    rng = SplitMix64(seed)
    a = rng.next_mod(r)
    b = rng.next_mod(r)
    full = [a, b]
    cons = []
    for i in range(num_constraints - 1):
        if i % 2:
            A = [(i + 1, 1)]
            B = [(i + 2, 1)]
            C = [(i + 3, 1)]
            tmp = a * b % r
        else:
            B = [(0, 1)]
            A = [(i + 1, 1), (i + 2, 1)]
            C = [(i + 3, 1)]
            tmp = (a + b) % r
        full.append(tmp)
        a, b = b, tmp
        cons.append((A, B, C))
    nvars = 2 + num_constraints
    A = []
    B = []
    fin = 0
    for i in range(1, nvars):
        A.append((i, 1))
        B.append((i, 1))
        fin = (fin + full[i - 1]) % r
    C = [(nvars, 1)]
    cons.append((A, B, C))
    full.append(fin * fin % r)
    cs = R1CS(num_inputs, nvars - num_inputs, cons)
    assert len(full) == nvars
    return cs, full[:num_inputs], full[num_inputs:]


def _lc_eval(lc, full, r):
    acc = 0
    for idx, coeff in lc:
        acc += coeff * (1 if idx == 0 else full[idx - 1])
    return acc % r


def r1cs_is_satisfied(cs: R1CS, primary, aux, r):
    full = list(primary) + list(aux)
    for A, B, C in cs.constraints:
        if (_lc_eval(A, full, r) * _lc_eval(B, full, r) - _lc_eval(C, full, r)) % r:
            return False
    return True


def swap_AB_if_beneficial(cs: R1CS) -> R1CS:
    """r1cs.hpp:193-215."""
    ta = set()
    tb = set()
    for A, B, _ in cs.constraints:
        ta.update(i for i, _ in A)
        tb.update(i for i, _ in B)
    if len(tb) > len(ta):
        return R1CS(cs.num_inputs, cs.num_aux, [(B, A, C) for A, B, C in cs.constraints])
    return cs


def domain_size(cs: R1CS) -> int:
    """size of the BASIC radix-2 domain that holds the instance (what a bare root of unity `w` stands for below)"""
    need = cs.num_constraints + cs.num_inputs + 1
    m = 1
    while m < need:
        m *= 2
    return m


def qap_domain(curve: "Curve", cs: R1CS, two_adicity: Optional[int] = None) -> EvaluationDomain:
    """the domain the reference reduces over: make_evaluation_domain(num_constraints + num_inputs + 1)
    (r1cs_to_qap.hpp:138-139, 229-230)"""
    return make_evaluation_domain(curve, cs.num_constraints + cs.num_inputs + 1, two_adicity)


def _domain_of(cs: R1CS, w, r: int) -> EvaluationDomain:
    """`w`: an EvaluationDomain, or a bare primitive root = the basic radix-2 domain of domain_size(cs) points"""
    if isinstance(w, EvaluationDomain):
        assert w.m >= cs.num_constraints + cs.num_inputs + 1
        return w
    return EvaluationDomain(EvaluationDomain.BASIC, domain_size(cs), w, r)


def lagrange_at(m: int, w: int, t: int, r: int) -> List[int]:
    """evaluate_all_lagrange_polynomials on {w^i}: L_i(t) = (t^m - 1) w^i / (m (t - w^i))."""
    tm = pow(t, m, r)
    if tm == 1:  # t in the domain
        out = [0] * m
        x = 1
        for i in range(m):
            if x == t % r:
                out[i] = 1
            x = x * w % r
        return out
    z = (tm - 1) * pow(m, -1, r) % r
    out = []
    x = 1
    for i in range(m):
        out.append(z * x % r * pow((t - x) % r, -1, r) % r)
        x = x * w % r
    return out


def qap_evaluate_at(cs: R1CS, t: int, w, r: int):
    """r1cs_to_qap.hpp:138-187: (At, Bt, Ct, Ht, Zt) with At/Bt/Ct of length num_variables+1.  `w`: see _domain_of."""
    dom = _domain_of(cs, w, r)
    m = dom.m
    u = dom.evaluate_all_lagrange_polynomials(t)
    nv = cs.num_variables
    At = [0] * (nv + 1)
    Bt = [0] * (nv + 1)
    Ct = [0] * (nv + 1)
    for i in range(cs.num_inputs + 1):
        At[i] = u[cs.num_constraints + i]
    for i, (A, B, C) in enumerate(cs.constraints):
        for idx, co in A:
            At[idx] = (At[idx] + u[i] * co) % r
        for idx, co in B:
            Bt[idx] = (Bt[idx] + u[i] * co) % r
        for idx, co in C:
            Ct[idx] = (Ct[idx] + u[i] * co) % r
    Ht = [pow(t, i, r) for i in range(m + 1)]
    Zt = dom.compute_vanishing_polynomial(t)
    return At, Bt, Ct, Ht, Zt


def witness_map(cs: R1CS, primary, aux, w, g: int, r: int) -> List[int]:
    """r1cs_to_qap.hpp:219-325 with d1=d2=d3=0: coefficients of H, length m+1.  `w`: see _domain_of."""
    dom = _domain_of(cs, w, r)
    m = dom.m
    full = list(primary) + list(aux)
    aA = [0] * m
    aB = [0] * m
    aC = [0] * m
    for i in range(cs.num_inputs + 1):
        aA[i + cs.num_constraints] = full[i - 1] if i > 0 else 1
    for i, (A, B, C) in enumerate(cs.constraints):
        aA[i] = (aA[i] + _lc_eval(A, full, r)) % r
        aB[i] = (aB[i] + _lc_eval(B, full, r)) % r
        aC[i] = (aC[i] + _lc_eval(C, full, r)) % r
    aA = dom.fft(multiply_by_coset(dom.inverse_fft(aA), g, r))
    aB = dom.fft(multiply_by_coset(dom.inverse_fft(aB), g, r))
    aC = dom.fft(multiply_by_coset(dom.inverse_fft(aC), g, r))
    H = dom.divide_by_z_on_coset([(aA[i] * aB[i] - aC[i]) % r for i in range(m)], g)  # :283-308
    H = multiply_by_coset(dom.inverse_fft(H), pow(g, -1, r), r)
    return H + [0]


@dataclass
class Groth16Key:
    alpha_g1: object
    beta_g1: object
    beta_g2: object
    delta_g1: object
    delta_g2: object
    A_query: list  # G1, len N+1
    B_query: list  # (G2, G1) dense, len N+1 (None entries where B_i(t) == 0)
    H_query: list  # G1, len m-1
    L_query: list  # G1, len N-n
    cs: R1CS


def groth16_keygen(curve: Curve, cs: R1CS, trapdoor, w: int) -> Groth16Key:
    """generator.hpp:86-236 with fixed (t, alpha, beta, gamma, delta) and the standard generators
    (the reference draws random generators; deterministic_basic_process :240-377 fixes them)."""
    r = curve.r
    t, alpha, beta, gamma, delta = trapdoor
    cs = swap_AB_if_beneficial(cs)
    At, Bt, Ct, Ht, Zt = qap_evaluate_at(cs, t, w, r)
    dinv = pow(delta, -1, r)
    n = cs.num_inputs
    N = cs.num_variables
    Lt = [(beta * At[i] + alpha * Bt[i] + Ct[i]) * dinv % r for i in range(n + 1, N + 1)]
    m = _domain_of(cs, w, r).m
    Hs = [Ht[i] * Zt % r * dinv % r for i in range(m - 1)]
    g1, g2 = curve.g1, curve.g2
    A_query = g1.batch_mul_gen(At)
    Bh = g1.batch_mul_gen(Bt)
    Bg = g2.batch_mul_gen(Bt)
    return Groth16Key(
        g1.mul(g1.gen, alpha),
        g1.mul(g1.gen, beta),
        g2.mul(g2.gen, beta),
        g1.mul(g1.gen, delta),
        g2.mul(g2.gen, delta),
        A_query,
        list(zip(Bg, Bh)),
        g1.batch_mul_gen(Hs),
        g1.batch_mul_gen(Lt),
        cs,
    )


def groth16_prove(curve: Curve, pk: Groth16Key, primary, aux, rr: int, ss: int, w: int):
    """prover.hpp:73-158 with (r, s) injected.  Returns affine (A in G1, B in G2, C in G1)."""
    r = curve.r
    g1, g2 = curve.g1, curve.g2
    cs = pk.cs
    H = witness_map(cs, primary, aux, w, curve.fr_generator, r)
    m = _domain_of(cs, w, r).m
    assert H[m - 1] == 0 and H[m] == 0
    cpa = [1] + list(primary) + list(aux)
    N = cs.num_variables
    n = cs.num_inputs
    eA = msm_naive(g1, pk.A_query[: N + 1], cpa[: N + 1])
    eBg = msm_naive(g2, [b[0] for b in pk.B_query], cpa[: N + 1])
    eBh = msm_naive(g1, [b[1] for b in pk.B_query], cpa[: N + 1])
    eH = msm_naive(g1, pk.H_query[: m - 1], H[: m - 1])
    eL = msm_naive(g1, pk.L_query, cpa[n + 1 : N + 1])
    gA = g1.add(g1.add(pk.alpha_g1, eA), g1.mul(pk.delta_g1, rr))
    gB1 = g1.add(g1.add(pk.beta_g1, eBh), g1.mul(pk.delta_g1, ss))
    gB2 = g2.add(g2.add(pk.beta_g2, eBg), g2.mul(pk.delta_g2, ss))
    gC = g1.add(g1.add(eH, eL), g1.add(g1.mul(gA, ss), g1.mul(gB1, rr)))
    gC = g1.add(gC, g1.neg(g1.mul(pk.delta_g1, rr * ss % r)))
    return gA, gB2, gC


def groth16_expected_in_exponent(curve: Curve, cs: R1CS, primary, aux, trapdoor, rr, ss, w):
    """The proof a correct prover must output, computed from the trapdoor instead of the key
    (comment formulas prover.hpp:141,145,151-153): three scalar multiplications of the generators."""
    r = curve.r
    t, alpha, beta, gamma, delta = trapdoor
    cs = swap_AB_if_beneficial(cs)
    At, Bt, Ct, Ht, Zt = qap_evaluate_at(cs, t, w, r)
    cpa = [1] + list(primary) + list(aux)
    a = (alpha + sum(x * y for x, y in zip(cpa, At)) + rr * delta) % r
    b = (beta + sum(x * y for x, y in zip(cpa, Bt)) + ss * delta) % r
    c_w = sum(x * y for x, y in zip(cpa, Ct)) % r
    # A(t)B(t) - C(t) = H(t) Z(t)
    a0 = (a - alpha - rr * delta) % r
    b0 = (b - beta - ss * delta) % r
    hz = (a0 * b0 - c_w) % r
    n = cs.num_inputs
    N = cs.num_variables
    dinv = pow(delta, -1, r)
    l = sum(cpa[i] * ((beta * At[i] + alpha * Bt[i] + Ct[i]) % r) for i in range(n + 1, N + 1)) % r
    c = ((hz + l) * dinv + ss * a + rr * b - rr * ss % r * delta) % r
    g1, g2 = curve.g1, curve.g2
    return g1.mul(g1.gen, a), g2.mul(g2.gen, b), g1.mul(g1.gen, c)


# --------------------------------------------------------------------------------------
# LPC / FRI, as far as the polynomial arithmetic goes (hashing stays outside: `tree_root` is any function of
# (leaves, elements_per_leaf)); restates
#   zk/commitments/polynomial/lpc.hpp:101-106, 113-200                 commit, proof_eval's combined_Q
#   zk/commitments/detail/polynomial/basic_fri.hpp:433-496            precommit: resize to D, coset-ordered leaves
#   zk/commitments/detail/polynomial/basic_fri.hpp:705-742            the FRI commit phase
#   zk/commitments/detail/polynomial/fold_polynomial.hpp:68-93        fold_polynomial, DFS form
#   zk/commitments/batched_commitment.hpp:113-129, 168-183            get_unique_points, eval_polys
# --------------------------------------------------------------------------------------
def dfs_resize(evals: Sequence[int], new_size: int, root_of_unity, r: int) -> List[int]:
    """polynomial_dfs::resize to a LARGER power-of-two domain: coefficients, zero padding, evaluation"""
    n = len(evals)
    c = intt(list(evals), root_of_unity(n.bit_length() - 1), r)
    return ntt(c + [0] * (new_size - n), root_of_unity(new_size.bit_length() - 1), r)


def fri_leaves(polys_on_D: Sequence[Sequence[int]], fri_step: int) -> List[int]:
    """basic_fri.hpp:456-492 (FRI::m = 2): leaf x holds, for every polynomial, the pairs (f[s_i], f[s_i + D/2])"""
    D, m = len(polys_on_D[0]), 2
    coset = 1 << fri_step
    out = []
    for x in range(D // coset):
        for f in polys_on_D:
            s = [[0, 0] for _ in range(coset // m)]
            s[0] = [x, (x + D // 2) % D]
            out += [f[s[0][0]], f[s[0][1]]]
            base, prev_half, i = D // (m * m), 1, 1
            while i < coset // m:
                for j in range(prev_half):
                    s[i][0] = (base + s[j][0]) % D
                    s[i][1] = (s[i][0] + D // 2) % D
                    out += [f[s[i][0]], f[s[i][1]]]
                    i += 1
                base //= m
                prev_half <<= 1
    return out


def fold_polynomial_dfs(f: Sequence[int], alpha: int, omega: int, r: int) -> List[int]:
    """fold_polynomial.hpp:68-93: f'(i) = 1/2 [(1 + alpha w^-i) f(i) + (1 - alpha w^-i) f(i + size/2)]"""
    half, inv2, winv = len(f) // 2, pow(2, -1, r), pow(omega, -1, r)
    out, wi = [], 1
    for i in range(half):
        out.append(inv2 * ((1 + alpha * wi) * f[i] + (1 - alpha * wi) * f[half + i]) % r)
        wi = wi * winv % r
    return out


def polynomial_shift(evals: Sequence[int], shift: int, domain_size: int = 0) -> List[int]:
    """math::polynomial_shift on a DFS vector (placeholder/permutation_argument.hpp:148): f(omega^shift X)"""
    n = len(evals)
    domain_size = domain_size or n
    step = n // domain_size
    return [evals[(i + shift * step) % n] for i in range(n)]


# --------------------------------------------------------------------------------------
# placeholder's quotient-polynomial chain (the NTT consumers next to the commitment schemes); restates
#   zk/snark/systems/plonk/placeholder/prover.hpp:55-69       detail::split_polynomial
#   zk/snark/systems/plonk/placeholder/prover.hpp:220-259     quotient_polynomial_split_dfs
#   zk/snark/systems/plonk/placeholder/prover.hpp:262-277     quotient_polynomial
#   zk/snark/systems/plonk/placeholder/gates_argument.hpp:93-121, 203-216   the polynomial_dfs arithmetic of the gate argument
# PINNED: split_polynomial to the reference's own literals (test/systems/plonk/placeholder/placeholder.cpp:513-532) and its
# identity f(y) = sum_i f_i(y) y^((max_degree + 1) i); the rest to the definitions (dense big-integer polynomial arithmetic:
# poly_mul / poly_divmod below), in tests/test_oracle_kat.py::test_placeholder_quotient_chain_definitions.
# --------------------------------------------------------------------------------------
def split_polynomial(f: Sequence[int], max_degree: int) -> List[List[int]]:
    """prover.hpp:55-69: chunks of max_degree + 1 coefficients"""
    chunk = max_degree + 1
    return [list(f[i:i + chunk]) for i in range(0, len(f), chunk)]


def gate_argument_dfs(products, mask: Sequence[int], extended_size: int, root_of_unity, r: int) -> List[int]:
    """gates_argument.hpp:203-216 for gates that are products: products = [(coefficient, [(evals, rotation), ...]), ...] over the
    ORIGINAL n-point domain; every factor is shifted over that domain (:108-110), resized to the extended domain (:111-113),
    multiplied pointwise; the weighted sum is multiplied by the mask polynomial (:215)."""
    F = [0] * extended_size
    for coeff, factors in products:
        term = [1] * extended_size
        for evals, rot in factors:
            e = polynomial_shift(evals, rot) if rot else list(evals)
            e = dfs_resize(e, extended_size, root_of_unity, r) if len(e) != extended_size else e
            term = [a * b % r for a, b in zip(term, e)]
        F = [(a + coeff * b) % r for a, b in zip(F, term)]
    m = dfs_resize(list(mask), extended_size, root_of_unity, r) if len(mask) != extended_size else list(mask)
    return [a * b % r for a, b in zip(F, m)]


def quotient_polynomial(F_dfs: Sequence[Sequence[int]], alphas: Sequence[int], rows_amount: int, root_of_unity, r: int) -> List[int]:
    """prover.hpp:262-277: coefficients of (sum_i alphas[i] F_dfs[i]) / (X^rows_amount - 1); parts live on different power-of-two
    domains, math::polynomial_sum adds them on the largest.  NOT condensed: len = size - rows_amount.  Asserts exactness."""
    size = max(max(len(f) for f in F_dfs), 2 * rows_amount)
    acc = [0] * size
    for f, a in zip(F_dfs, alphas):
        if len(f) == 0:
            continue
        e = dfs_resize(list(f), size, root_of_unity, r) if len(f) != size else list(f)
        acc = [(x + a * y) % r for x, y in zip(acc, e)]
    coeffs = intt(acc, root_of_unity(size.bit_length() - 1), r)
    # long division by Z = X^n - 1, column by column: q[i] = f[i + n] + q[i + n] (checked against q Z == f in the tests)
    n = rows_amount
    q = [0] * (size - n)
    for i in range(size - n - 1, -1, -1):
        q[i] = (coeffs[i + n] + (q[i + n] if i + n < size - n else 0)) % r
    assert all((coeffs[i] + q[i]) % r == 0 for i in range(min(n, size - n))) and all(coeffs[i] == 0 for i in range(size - n, n)), \
        "F_consolidated is not divisible by Z"
    return q


def quotient_polynomial_split_dfs(T: Sequence[int], rows_amount: int, split_polynomial_size: int, dfs_size: int, root_of_unity, r: int) -> List[List[int]]:
    """prover.hpp:220-259: split into chunks of rows_amount coefficients, each from_coefficients over the dfs_size-point domain; parts
    the quotient does not reach are the zero polynomial"""
    parts = split_polynomial(poly_trim(T), rows_amount - 1)
    assert len(parts) <= split_polynomial_size
    w = root_of_unity(dfs_size.bit_length() - 1)
    out = []
    for k in range(split_polynomial_size):
        c = parts[k] if k < len(parts) else []
        out.append(ntt(list(c) + [0] * (dfs_size - len(c)), w, r))
    return out


# placeholder's permutation argument, prover side (zk/snark/systems/plonk/placeholder/permutation_argument.hpp:70-224, the
# permutation_parts == 1 form and the multi-part one of max_quotient_chunks != 0).  PINNED to its definition and its purpose
# (tests/test_oracle_kat.py: on a genuine instance the product closes and every F vanishes on the rows): V_P by the row-by-row recurrence of :126-136 (one inversion per row), the
# three constraint polynomials as DENSE coefficient-form polynomial arithmetic of the comment formulas (:166, 175, 215) -- domain sizes
# of the reference's intermediate polynomial_dfs objects are crypto3-math's business and do not show in the coefficients.
def inv0(x: int, r: int) -> int:
    """the field type's inversed() as the argument loops use it: 0 for 0 (crypto3-algebra is not in the tree; the value is the one both the
    round-4 kernels and ADVICE r4 assume -- a zero denominator has probability ~ k n / r for honest inputs)"""
    return pow(x, -1, r) if x % r else 0


def permutation_grand_product(cols, S_id, S_sigma, beta: int, gamma: int, r: int):
    n = len(cols[0])
    g = [[(c[j] + beta * s_[j] + gamma) % r for j in range(n)] for c, s_ in zip(cols, S_id)]
    h = [[(c[j] + beta * s_[j] + gamma) % r for j in range(n)] for c, s_ in zip(cols, S_sigma)]
    V = [1] * n
    for j in range(1, n):
        nom = den = 1
        for gi, hi in zip(g, h):
            nom = nom * gi[j - 1] % r
            den = den * hi[j - 1] % r
        V[j] = V[j - 1] * nom * inv0(den, r) % r
    return g, h, V


def permutation_argument(cols, S_id, S_sigma, q_last, q_blind, lagrange_0, beta: int, gamma: int, root_of_unity, r: int, max_quotient_chunks: int = 0,
                         alphas: Sequence[int] = (), usable_rows: int = None):
    """-> (V_P evaluations, [F0, F1, F2] as trimmed coefficient lists):
         F0 = L_0 (1 - V_P),  F1 = (1 - (q_last + q_blind)) (V_P(omega X) h - V_P g),  F2 = q_last V_P (V_P - 1),
       g = prod_i g_i, h = prod_i h_i.
    max_quotient_chunks = c != 0 (permutation_argument.hpp:147-160, 188-207): the factors go in groups of c - 1 (g_0, h_0), (g_1, h_1), ...;
    every group but the last gives an intermediate polynomial current[j] = previous[j] g_i[j] / h_i[j] over the usable rows (the rest keeps V_P's
    values), previous starting as V_P, and
         F1 = ((q_last + q_blind) - 1) (sum_i alphas[i] (previous_i g_i - current_i h_i) + previous_last g_last - V_P(omega X) h_last);
    -> (V_P, [F0, F1, F2], [current_0, ...] evaluations)"""
    n = len(cols[0])
    w = root_of_unity(n.bit_length() - 1)
    g, h, V = permutation_grand_product(cols, S_id, S_sigma, beta, gamma, r)
    co = lambda e: poly_trim(intt(list(e), w, r))
    VP, VPs = co(V), co(polynomial_shift(V, 1))
    one = [1]
    F0 = poly_mul(co(lagrange_0), poly_sub(one, VP, r), r)
    F2 = poly_mul(co(q_last), poly_mul(VP, poly_sub(VP, one, r), r), r)
    q = poly_add(co(q_last), co(q_blind), r)
    if max_quotient_chunks == 0:
        G, H = [1], [1]
        for gi, hi in zip(g, h):
            G, H = poly_mul(G, co(gi), r), poly_mul(H, co(hi), r)
        F1 = poly_mul(poly_sub(one, q, r), poly_sub(poly_mul(VPs, H, r), poly_mul(VP, G, r), r), r)
        return V, [poly_trim(F0), poly_trim(F1), poly_trim(F2)]
    step = max_quotient_chunks - 1
    groups = [(g[i:i + step], h[i:i + step]) for i in range(0, len(g), step)]
    assert len(alphas) == len(groups) - 1
    prev, currents, acc = list(V), [], []
    for idx, (gg, hh) in enumerate(groups):
        G, H = [1], [1]
        for gi, hi in zip(gg, hh):
            G, H = poly_mul(G, co(gi), r), poly_mul(H, co(hi), r)
        if idx < len(groups) - 1:
            cur = list(V)
            for j in range(usable_rows):
                nom = den = 1
                for gi, hi in zip(gg, hh):
                    nom, den = nom * gi[j] % r, den * hi[j] % r
                cur[j] = prev[j] * nom % r * pow(den, -1, r) % r
            currents.append(cur)
            acc = poly_add(acc, poly_scale(poly_sub(poly_mul(co(prev), G, r), poly_mul(co(cur), H, r), r), alphas[idx], r), r)
            prev = cur
        else:
            acc = poly_add(acc, poly_sub(poly_mul(co(prev), G, r), poly_mul(VPs, H, r), r), r)
    F1 = poly_mul(poly_sub(q, one, r), acc, r)
    return V, [poly_trim(F0), poly_trim(F1), poly_trim(F2)], currents


def reduce_dfs_polynomial_domain(evals: Sequence[int], new_size: int) -> List[int]:
    """lookup_argument.hpp:498-517: every (size / new_size)-th evaluation"""
    assert len(evals) % new_size == 0
    step = len(evals) // new_size
    return [evals[i * step] for i in range(new_size)]


def lookup_prepare_value(tables, selectors, constants, theta: int, mask, root_of_unity, r: int) -> List[List[int]]:
    """prepare_lookup_value, lookup_argument.hpp:411-433.  tables: [(tag_index, columns_number, lookup_options)], lookup_options[o][i] = index of a
    constant column; selectors / constants / mask: DFS vectors on the n-row basic domain (degree n - 1).  Per table t and option o
        v = (t + 1) tag;  v += theta^(i + 1) tag constant_i  (i < columns_number);  v *= mask
    as POLYNOMIALS: polynomial_dfs's operator* puts a product on the smallest power-of-two domain that holds the sum of the degrees, so v
    lives on 4n points (2n without columns).  Dense coefficient arithmetic here."""
    n = len(mask)
    co = lambda e: intt(list(e), root_of_unity(len(e).bit_length() - 1), r)
    out = []
    for t_id, (tag_index, columns_number, options) in enumerate(tables):
        tag = co(selectors[tag_index])
        for option in options:
            v = poly_scale(tag, (t_id + 1) % r, r)
            theta_acc, degree = theta, n - 1
            for i in range(columns_number):
                v = poly_add(v, poly_scale(poly_mul(tag, co(constants[option[i]]), r), theta_acc, r), r)
                theta_acc = theta_acc * theta % r
                degree = 2 * (n - 1)
            v = poly_mul(v, co(mask), r)
            degree += n - 1
            size = 1
            while size < degree + 1:
                size <<= 1
            v = (list(v) + [0] * size)[:size]
            out.append(ntt(v, root_of_unity(size.bit_length() - 1), r))
    return out


# the order in which the two arguments touch the transcript and the commitment scheme (what decides whether a proof is byte-identical):
# a replay of the reference's statements, for the shim's reference-shaped entry points to be held against (tests/cpp/arguments_test.cpp)
EV_CHALLENGE, EV_ABSORB, EV_APPEND, EV_COMMIT = 1, 3, 100, 200
PERMUTATION_BATCH, LOOKUP_BATCH = 2, 4      # systems/plonk/placeholder/proof.hpp:37-41


def permutation_argument_events(permutation_parts: int) -> List[int]:
    """placeholder_permutation_argument::prove_eval, permutation_argument.hpp:70-224"""
    ev = [EV_CHALLENGE, EV_CHALLENGE]                               # :95-97    beta, gamma
    ev.append(EV_APPEND + PERMUTATION_BATCH)                        # :139      V_P
    ev += [EV_CHALLENGE] * (permutation_parts - 1)                  # :181-183  permutation_alphas
    ev += [EV_APPEND + PERMUTATION_BATCH] * (permutation_parts - 1)  # :200      current_poly of every part but the last
    return ev


def lookup_argument_events(n_sorted: int, n_parts: int) -> List[int]:
    """placeholder_lookup_argument_prover: the constructor (:128-151) and prove_eval (:153-296)"""
    ev = [EV_CHALLENGE]                                             # :150      theta
    ev += [EV_APPEND + LOOKUP_BATCH] * n_sorted                     # :192-194  the sorted polynomials
    ev += [EV_COMMIT + LOOKUP_BATCH, EV_ABSORB]                     # :195-196  commit, transcript(lookup_commitment)
    ev += [EV_CHALLENGE, EV_CHALLENGE]                              # :199-200  beta, gamma
    ev += [EV_CHALLENGE] * (n_parts - 1)                            # :204-206  lookup_alphas
    ev.append(EV_APPEND + PERMUTATION_BATCH)                        # :213      V_L
    ev += [EV_APPEND + PERMUTATION_BATCH] * (n_parts - 1)           # :267      current_poly of every part but the last
    ev += [EV_CHALLENGE] * (n_sorted - 1)                           # :282      one alpha per sorted vector but the first
    return ev


def lookup_sort_polynomials(reduced_input, reduced_value, domain_size: int, usable_rows: int, strict: bool = True) -> List[List[int]]:
    """lookup_argument.hpp:565-638: the values of the table columns in their order, each repeated as often as it is looked up
    (+ once per table occurrence), dealt over |input| + |value| vectors of usable_rows entries; entry usable_rows of every vector
    but the last repeats the next vector's head.
    strict (the reference with its BOOST_ASSERTs, :583, :613-617): a looked-up value that is in no table, or more emitted entries than
    the vectors hold, is an error.  strict = False is the reference as it runs WITHOUT assertions where that is still defined: a
    looked-up value that is in no table is counted under a key the walk never reaches (:584 `sorting_map[...]++` inserts it), so it
    changes nothing; a value whose table entries are not adjacent is emitted `count` times at the end of EACH of its runs (:609-624);
    writing behind the last vector stays an error (the reference indexes past `sorted` there)."""
    count = {}
    for v in reduced_value:
        for j in range(usable_rows):
            count[v[j]] = count.get(v[j], 0) + 1
    for v in reduced_input:
        for j in range(usable_rows):
            if strict:
                assert v[j] in count, "a looked-up value that is in no table"
            count[v[j]] = count.get(v[j], 0) + 1
    total = len(reduced_input) + len(reduced_value)
    flat = []
    prev = 0
    for v in reduced_value:
        for j in range(usable_rows):
            if v[j] != prev:
                flat.extend([prev] if prev == 0 else [prev] * count[prev])
                prev = v[j]
    if prev != 0:
        flat.extend([prev] * count[prev])
    assert len(flat) <= total * usable_rows, "the emitted sequence does not fit the sorted vectors"
    out = [[0] * domain_size for _ in range(total)]
    for idx, val in enumerate(flat):
        out[idx // usable_rows][idx % usable_rows] = val
    for i in range(total - 1):
        out[i][usable_rows] = out[i + 1][0]
    return out


def lookup_grand_product(reduced_input, reduced_value, sorted_, beta: int, gamma: int, usable_rows: int, r: int) -> List[int]:
    """compute_V_L, lookup_argument.hpp:375-409 (the row-by-row recurrence, one inversion per row)"""
    n = len(sorted_[0])
    V = [0] * n
    V[0] = 1
    part1 = (1 + beta) * gamma % r
    for k in range(1, usable_rows + 1):
        g = pow(1 + beta, len(reduced_input), r)
        for v in reduced_input:
            g = g * (gamma + v[k - 1]) % r
        for v in reduced_value:
            g = g * (part1 + v[k - 1] + beta * v[k]) % r
        h = 1
        for v in sorted_:
            h = h * (part1 + v[k - 1] + beta * v[k]) % r
        V[k] = V[k - 1] * g % r * inv0(h, r) % r
    return V


def lookup_argument(lookup_input, lookup_value, sorted_, q_last, q_blind, lagrange_0, beta: int, gamma: int, alphas: Sequence[int], usable_rows: int,
                    root_of_unity, r: int, part_sizes: Sequence[int] = None, part_alphas: Sequence[int] = ()):
    """placeholder_lookup_argument_prover::prove_eval from `sorted` on (lookup_argument.hpp:198-296):
    -> (V_L evaluations, [F0, F1, F2, F3] as trimmed coefficient lists [, intermediate polynomials' evaluations]):
         F0 = L_0 (1 - V_L),  F1 = q_last (V_L^2 - V_L),
         F2 = ((q_last + q_blind) - 1) (V_L g - V_L(omega X) h),   g = prod_i (1 + beta)(gamma + input_i) prod_i ((1 + beta) gamma + value_i + beta value_i(omega X)),
                                                                   h = prod_i ((1 + beta) gamma + sorted_i + beta sorted_i(omega X)),
         F3 = sum_i alphas[i] L_0 (sorted_(i + 1) - sorted_i(omega^usable_rows X)).
    part_sizes (lookup_parts(max_quotient_chunks), :56-107; None = one part): the g factors (inputs first, then values) and the h factors go in
    groups of these sizes (:297-373); every group but the last gives an intermediate polynomial current[j] = previous[j] g_i[j] / h_i[j] over the
    usable rows (the rest keeps V_L's values), and F2 = ((q_last + q_blind) - 1)(sum_i part_alphas[i] (previous_i g_i - current_i h_i) +
    previous_last g_last - V_L(omega X) h_last) (:252-276).
    lookup_input may live on larger domains than n (expressions of degree > 1); everything is dense coefficient arithmetic here."""
    n = len(sorted_[0])
    co = lambda e: poly_trim(intt(list(e), root_of_unity(len(e).bit_length() - 1), r))
    red_in = [reduce_dfs_polynomial_domain(v, n) for v in lookup_input]
    red_val = [reduce_dfs_polynomial_domain(v, n) for v in lookup_value]
    V = lookup_grand_product(red_in, red_val, sorted_, beta, gamma, usable_rows, r)
    part1 = (1 + beta) * gamma % r
    # the factors as (coefficients, values on the rows)
    nxt = lambda v: polynomial_shift(v, 1, n)
    g_f = [(poly_scale(poly_add([gamma], co(v), r), (1 + beta) % r, r), [(1 + beta) * (gamma + x) % r for x in rv]) for v, rv in zip(lookup_input, red_in)]
    g_f += [(poly_add(poly_add([part1], co(v), r), poly_scale(co(nxt(v)), beta, r), r), [(part1 + a + beta * b) % r for a, b in zip(rv, rv[1:] + rv[:1])])
            for v, rv in zip(lookup_value, red_val)]    # lookup_value may live on a larger domain too (prepare_lookup_value's products: 4n)
    h_f = [(poly_add(poly_add([part1], co(v), r), poly_scale(co(nxt(v)), beta, r), r), [(part1 + a + beta * b) % r for a, b in zip(v, nxt(v))]) for v in sorted_]
    VL, VLs = co(V), co(polynomial_shift(V, 1))
    one = [1]
    L0 = co(lagrange_0)
    F0 = poly_mul(L0, poly_sub(one, VL, r), r)
    F1 = poly_mul(co(q_last), poly_sub(poly_mul(VL, VL, r), VL, r), r)
    sizes = list(part_sizes) if part_sizes is not None else [len(sorted_)]
    assert sum(sizes) == len(sorted_) == len(g_f) and len(part_alphas) == len(sizes) - 1
    prev, currents, acc, at = list(V), [], [], 0
    for idx, sz in enumerate(sizes):
        G, H = [1], [1]
        for (gc, _), (hc, _) in zip(g_f[at:at + sz], h_f[at:at + sz]):
            G, H = poly_mul(G, gc, r), poly_mul(H, hc, r)
        if idx < len(sizes) - 1:
            cur = list(V)
            for j in range(usable_rows):
                nom = den = 1
                for (_, gv), (_, hv) in zip(g_f[at:at + sz], h_f[at:at + sz]):
                    nom, den = nom * gv[j] % r, den * hv[j] % r
                cur[j] = prev[j] * nom % r * pow(den, -1, r) % r
            currents.append(cur)
            acc = poly_add(acc, poly_scale(poly_sub(poly_mul(co(prev), G, r), poly_mul(co(cur), H, r), r), part_alphas[idx], r), r)
            prev = cur
        else:
            acc = poly_add(acc, poly_sub(poly_mul(co(prev), G, r), poly_mul(VLs, H, r), r), r)
        at += sz
    F2 = poly_mul(poly_sub(poly_add(co(q_last), co(q_blind), r), one, r), acc, r)
    F3 = []
    # F_dfs_3_parts = sorted[1:], part i (0-based) subtracts sorted[i] shifted by usable_rows   (lookup_argument.hpp:281-288)
    for i in range(len(sorted_) - 1):
        d = poly_sub(co(sorted_[i + 1]), co(polynomial_shift(sorted_[i], usable_rows, n)), r)
        F3 = poly_add(F3, poly_scale(poly_mul(d, L0, r), alphas[i], r), r)
    F = [poly_trim(F0), poly_trim(F1), poly_trim(F2), poly_trim(F3)]
    return (V, F) if part_sizes is None else (V, F, currents)


def lpc_proof_eval(r: int, batches: dict, points: dict, fixed: Sequence[int], log_domain: int, step_list: Sequence[int], root_of_unity,
                   challenges: Sequence[int], tree_root):
    """batches[k] = list of DFS polynomials (lists of ints); points[k][i] = evaluation points of polynomial i of batch k;
    fixed = batch ids marked fixed; challenges = (etha_preprocess, etha_setup, theta, alpha_0, ...) in drawing order.
    -> (commit roots {k: root}, z {k: [[values]]}, fri_roots, final polynomial coefficients)"""
    D = 1 << log_domain
    ch = iter(challenges)
    coeffs = {k: [intt(list(p), root_of_unity(len(p).bit_length() - 1), r) for p in ps] for k, ps in batches.items()}
    roots = {}
    for k in sorted(batches):
        ext = [dfs_resize(p, D, root_of_unity, r) if len(p) < D else list(p) for p in batches[k]]
        roots[k] = tree_root(fri_leaves(ext, step_list[0]), len(ext) << step_list[0])
    etha = next(ch)
    fixed_vals = {k: [poly_eval(c, etha, r) for c in coeffs[k]] for k in fixed}
    assert next(ch) == etha  # setup draws the same challenge from its copy of the preprocessed transcript
    z = {k: [[poly_eval(coeffs[k][i], x, r) for x in points[k][i]] for i in range(len(coeffs[k]))] for k in sorted(batches)}
    theta = next(ch)
    uniq = []
    for k in sorted(points):
        for pl in points[k]:
            for x in pl:
                if x not in uniq:
                    uniq.append(x)
    theta_acc, combined = 1, []
    for pt in uniq:
        q = []
        for k in sorted(batches):
            for i, c in enumerate(coeffs[k]):
                if pt not in points[k][i]:
                    continue
                zi = z[k][i][points[k][i].index(pt)]
                q = poly_add(q, poly_scale(c, theta_acc, r), r)
                q = poly_sub(q, [zi * theta_acc % r], r)
                theta_acc = theta_acc * theta % r
        qq, rem = poly_divmod(q, [(-pt) % r, 1], r)
        assert not poly_trim(rem)
        combined = poly_add(combined, qq, r)
    for k in sorted(batches):
        if k not in fixed:
            continue
        q = []
        for i, c in enumerate(coeffs[k]):
            q = poly_add(q, poly_scale(c, theta_acc, r), r)
            q = poly_sub(q, [fixed_vals[k][i] * theta_acc % r], r)
            theta_acc = theta_acc * theta % r
        qq, rem = poly_divmod(q, [(-etha) % r, 1], r)
        assert not poly_trim(rem)
        combined = poly_add(combined, qq, r)
    f = ntt(list(combined) + [0] * (D - len(combined)), root_of_unity(log_domain), r)
    pre = tree_root(fri_leaves([f], step_list[0]), 1 << step_list[0])
    fri_roots, t = [], 0
    for i, step in enumerate(step_list):
        fri_roots.append(pre)
        for _ in range(step):
            f = fold_polynomial_dfs(f, next(ch), root_of_unity(log_domain - t), r)
            t += 1
        if i != len(step_list) - 1:
            pre = tree_root(fri_leaves([f], step_list[i + 1]), 1 << step_list[i + 1])
    final = intt(f, root_of_unity(log_domain - t), r)
    return roots, z, fri_roots, final


# --------------------------------------------------------------------------------------
# the reference's KAT carrier (SnarkPack commitment-key openings)
# --------------------------------------------------------------------------------------
def ipp2_poly_coeffs(tr: Sequence[int], r_shift: int, r: int) -> List[int]:
    """ipp2/prover.hpp:140-155."""
    c = [1]
    p2 = r_shift % r
    for x in tr:
        c = c + [cj * (x * p2 % r) % r for cj in c]
        p2 = p2 * p2 % r
    return c


def ipp2_poly_eval(tr: Sequence[int], z: int, r_shift: int, r: int) -> int:
    """ipp2/prover.hpp:99-125."""
    pw = z * r_shift % r
    res = 1
    for x in tr:
        res = res * (1 + x * pw) % r
        pw = pw * pw % r
    return res


def _quotient_by_linear(f: Sequence[int], fz: int, z: int, r: int) -> List[int]:
    """(f(X) - fz)/(X - z), padded with zeros to len(f) (ipp2/prover.hpp:171-200)."""
    g = list(f)
    g[0] = (g[0] - fz) % r
    n = len(g)
    q = [0] * n
    carry = 0
    for i in range(n - 1, 0, -1):
        carry = (g[i] + carry * z) % r
        q[i - 1] = carry
    assert (g[0] + carry * z) % r == 0, "remainder must vanish"
    return q


def structured_generators(G: Group, n: int, s: int):
    """ipp2/srs.hpp:44-56: {s^i * G}."""
    out = [G.gen]
    for _ in range(1, n):
        out.append(G.mul(out[-1], s))
    return out


def ipp2_prove_commitment_v(curve: Curve, n, alpha, beta, tr, z):
    r = curve.r
    f = ipp2_poly_coeffs(tr, 1, r)
    fz = ipp2_poly_eval(tr, z, 1, r)
    q = _quotient_by_linear(f, fz, z, r)
    ha = structured_generators(curve.g2, n, alpha)
    hb = structured_generators(curve.g2, n, beta)
    return q, ha, hb


def ipp2_prove_commitment_w(curve: Curve, n, alpha, beta, tr, r_shift, z):
    r = curve.r
    f = [0] * n + ipp2_poly_coeffs(tr, r_shift, r)
    fwz = ipp2_poly_eval(tr, z, r_shift, r) * pow(z, n, r) % r
    q = _quotient_by_linear(f, fwz, z, r)
    ga = structured_generators(curve.g1, 2 * n, alpha)
    gb = structured_generators(curve.g1, 2 * n, beta)
    return q, ga, gb


# --------------------------------------------------------------------------------------
# limb helpers shared by the tests (canonical little-endian u64 limbs at the C ABI)
# --------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------------
# KZG v2 batched opening proof (zk/commitments/polynomial/kzg_v2.hpp:236-305 on top of
# zk/commitments/batched_commitment.hpp:73-183).  Coefficient lists are low degree first.
def poly_trim(a: Sequence[int]) -> List[int]:
    a = list(a)
    while a and a[-1] == 0:
        a.pop()
    return a


def poly_add(a, b, r):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % r for i in range(n)]


def poly_sub(a, b, r):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % r for i in range(n)]


def poly_scale(a, c, r):
    return [x * c % r for x in a]


def poly_mul(a, b, r):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % r
    return out


def poly_eval(a, z, r):
    acc = 0
    for x in reversed(a):
        acc = (acc * z + x) % r
    return acc


def poly_divmod(a, b, r):
    """long division: returns (quotient, remainder); b's leading coefficient must be invertible"""
    a, b = poly_trim(a), poly_trim(b)
    if len(a) < len(b):
        return [], a
    inv = pow(b[-1], -1, r)
    q = [0] * (len(a) - len(b) + 1)
    rem = list(a)
    for i in range(len(q) - 1, -1, -1):
        c = rem[i + len(b) - 1] * inv % r
        q[i] = c
        if c:
            for j, y in enumerate(b):
                rem[i + j] = (rem[i + j] - c * y) % r
    return q, poly_trim(rem[: len(b) - 1])


def vanishing_poly(points, r):
    """get_V (batched_commitment.hpp:79-87): prod (X - x_i)"""
    v = [1]
    for x in points:
        v = poly_mul(v, [(-x) % r, 1], r)
    return v


def lagrange_interpolation(pairs, r):
    """math::lagrange_interpolation as get_U uses it (batched_commitment.hpp:100-111): the unique polynomial of
    degree < len(pairs) through (x_k, y_k)"""
    out = []
    for k, (xk, yk) in enumerate(pairs):
        num, den = [1], 1
        for j, (xj, _) in enumerate(pairs):
            if j != k:
                num = poly_mul(num, [(-xj) % r, 1], r)
                den = den * (xk - xj) % r
        out = poly_add(out, poly_scale(num, yk * pow(den, -1, r) % r, r), r)
    return out


def kzg_v2_proof_eval(r: int, polys: dict, points: dict, theta: int, theta2: int):
    """polys[k][i]: coefficient list of polynomial i of batch k; points[k][i]: its evaluation points.
    Returns (z, f, L): z[k][i][j] = poly(point) (eval_polys), f = the quotient committed as pi_1
    (kzg_v2.hpp:253-269), L = the quotient committed as pi_2 (:281-292).  The challenges are inputs: the
    transcript (hashing, byte packing) is outside this path."""
    z = {k: [[poly_eval(p, x, r) for x in points[k][i]] for i, p in enumerate(ps)] for k, ps in polys.items()}
    merged = sorted({x for k in points for pl in points[k] for x in pl})          # merge_eval_points (:121-130)
    V = vanishing_poly(merged, r)

    def diffpoly(pts):                                                             # set_difference_polynom (:132-148)
        return vanishing_poly([x for x in merged if x not in pts], r)

    theta_i, f = 1, []
    for k in sorted(polys):
        for i, p in enumerate(polys[k]):
            U = lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            f = poly_add(f, poly_scale(poly_mul(poly_sub(p, U, r), diffpoly(points[k][i]), r), theta_i, r), r)
            theta_i = theta_i * theta % r
    f, rem = poly_divmod(f, V, r)
    assert not rem                                                                 # BOOST_ASSERT(f % V == 0) (:266)
    theta_i, L = 1, []
    for k in sorted(polys):
        for i, p in enumerate(polys[k]):
            U = lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            zts = poly_eval(diffpoly(points[k][i]), theta2, r)
            L = poly_add(L, poly_scale(poly_sub(p, [poly_eval(U, theta2, r)], r), theta_i * zts % r, r), r)
            theta_i = theta_i * theta % r
    L = poly_sub(L, poly_scale(f, poly_eval(V, theta2, r), r), r)
    assert poly_eval(L, theta2, r) == 0                                            # (:290)
    L, rem = poly_divmod(L, [(-theta2) % r, 1], r)
    assert not rem
    return z, f, L


def kzg_v1_proof_eval(r: int, polys: dict, points: dict, gamma: int):
    """kzg_commitment_scheme::proof_eval (kzg.hpp:782-807), the first batched scheme: polys[k][i] = coefficient list of polynomial i
    of batch k, points[k][i] = its evaluation points.  Returns (z, accum): z[k][i][j] = poly(point) (eval_polys) and
    accum = sum_j gamma^j (f_j - U_j) / V(S_j) (:793-800), the polynomial committed as kzg_proof.  The challenge is an input:
    the transcript (hashing, byte packing) is outside this path."""
    z = {k: [[poly_eval(p, x, r) for x in points[k][i]] for i, p in enumerate(ps)] for k, ps in polys.items()}
    factor, accum = 1, [0]
    for k in sorted(polys):
        for i, p in enumerate(polys[k]):
            U = lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            q, rem = poly_divmod(poly_sub(p, U, r), vanishing_poly(points[k][i], r), r)
            assert not rem                                                         # (f - U) vanishes on S
            accum = poly_add(accum, poly_scale(q, factor, r), r)
            factor = factor * gamma % r
    return z, poly_trim(accum)


# ---------------------------------------------------------------------------------------------------------
# Compressed point encoding of BLS12-381 keys on the wire (the ZCash format; what
# nil::marshalling::bincode::curve<bls12<381>>::point_to_bytes / g1_point_from_bytes produce and read -- pinned by the
# literal vectors of AGG:932-1010 -- and what the proving-key serializer emits per point, g16/marshalling.hpp:111-112,
# 178-201: G1 = one Fq, G2 = two).  Byte 0 carries three flags: 0x80 compressed, 0x40 infinity, 0x20 "y is the
# lexicographically larger root"; the rest is x big-endian (G2: x.c1 then x.c0).
def _sqrt_fq(a: int, p: int):
    r = pow(a, (p + 1) // 4, p)  # p = 3 mod 4
    return r if r * r % p == a % p else None


def _sqrt_fq2(a, p: int):
    """complex method: a = a0 + a1 u, u^2 = -1"""
    a0, a1 = a[0] % p, a[1] % p
    if a1 == 0:
        r = _sqrt_fq(a0, p)
        if r is not None:
            return (r, 0)
        r = _sqrt_fq((-a0) % p, p)
        return None if r is None else (0, r)
    s = _sqrt_fq((a0 * a0 + a1 * a1) % p, p)
    if s is None:
        return None
    inv2 = pow(2, -1, p)
    x0 = _sqrt_fq((a0 + s) * inv2 % p, p)
    if x0 is None:
        x0 = _sqrt_fq((a0 - s) * inv2 % p, p)
    if x0 is None or x0 == 0:
        return None
    x1 = a1 * pow(2 * x0, -1, p) % p
    return (x0, x1)


def bls12_381_compress(group: int, P) -> bytes:
    p = BLS12_381.p
    n = 48 * group
    if P is None:
        return bytes([0xC0]) + bytes(n - 1)
    if group == 1:
        x, y = P
        larger = y > (p - 1) // 2
        body = x.to_bytes(48, "big")
    else:
        (x0, x1), (y0, y1) = P
        larger = (y1 > (p - 1) // 2) if y1 != 0 else (y0 > (p - 1) // 2)
        body = x1.to_bytes(48, "big") + x0.to_bytes(48, "big")
    return bytes([body[0] | 0x80 | (0x20 if larger else 0)]) + body[1:]


def bls12_381_decompress(group: int, data: bytes):
    """returns the affine point (None for infinity); raises ValueError on a malformed or off-curve encoding"""
    p = BLS12_381.p
    n = 48 * group
    if len(data) != n or not data[0] & 0x80:
        raise ValueError("not a compressed point")
    if data[0] & 0x40:
        if data[0] & 0x3F or any(data[1:]):
            raise ValueError("malformed infinity")
        return None
    larger = bool(data[0] & 0x20)
    body = bytes([data[0] & 0x1F]) + data[1:]
    if group == 1:
        x = int.from_bytes(body, "big")
        if x >= p:
            raise ValueError("x not reduced")
        y = _sqrt_fq((x * x * x + 4) % p, p)
        if y is None:
            raise ValueError("not on the curve")
        if (y > (p - 1) // 2) != larger:
            y = p - y
        return (x, y)
    x1, x0 = int.from_bytes(body[:48], "big"), int.from_bytes(body[48:], "big")
    if x0 >= p or x1 >= p:
        raise ValueError("x not reduced")
    F = Fq2(p)
    y = _sqrt_fq2(F.add(F.mul(F.sqr((x0, x1)), (x0, x1)), (4, 4)), p)
    if y is None:
        raise ValueError("not on the curve")
    is_larger = (y[1] > (p - 1) // 2) if y[1] != 0 else (y[0] > (p - 1) // 2)
    if is_larger != larger:
        y = F.neg(y)
    return ((x0, x1), y)


def to_limbs(v: int, n: int) -> List[int]:
    return [(v >> (64 * i)) & MASK64 for i in range(n)]


def from_limbs(limbs: Sequence[int]) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v
