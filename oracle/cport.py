"""TEST INFRASTRUCTURE ONLY -- ctypes view of oracle/liboracle.so (the C++ CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All field elements cross as canonical little-endian u64 limbs in numpy uint64 arrays:
  Fr: 4 limbs; BLS12-381 Fq: 6 limbs; BN254 Fq: 4 limbs; G1 affine = 2 Fq; G2 affine = 2 Fq2 = 4 Fq.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BLS12_381, BN254 = 0, 1
G1, G2 = 1, 2
FQ_LIMBS = {BLS12_381: 6, BN254: 4}


def point_limbs(curve: int, group: int) -> int:
    return 2 * FQ_LIMBS[curve] * group


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "zk_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return so


def _host_parallelism() -> int:
    """CPUs this process can actually keep busy: the affinity mask capped by the cgroup's CPU quota.  OpenMP's default (every CPU of the
    machine: 128+ on the GPU boxes) under a 16-CPU quota makes every parallel region spin against the throttle -- minutes instead of
    seconds for the many small regions of the transform-based oracle."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            cpus = min(cpus, max(1, int(int(q[0]) / int(q[1]))))
    except Exception:
        pass
    return max(1, cpus)


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.zko_g16_new.restype = ctypes.c_void_p
        _LIB.zko_bases_new.restype = ctypes.c_void_p
        global _OMP_DEFAULT
        _OMP_DEFAULT = _LIB.zko_num_threads()
        _LIB.zko_set_threads(min(_OMP_DEFAULT, _host_parallelism()))
    return _LIB


_OMP_DEFAULT = 1


def omp_default_threads() -> int:
    """OpenMP's own default thread count (before lib() capped it to the host's real parallelism): the upper end of bench.py's thread sweep"""
    lib()
    return _OMP_DEFAULT


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def num_threads() -> int:
    return lib().zko_num_threads()


def set_threads(t: int):
    lib().zko_set_threads(int(t))


def random_fr(curve: int, seed: int, n: int) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    lib().zko_random_fr(curve, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p(out))
    return out


def batch_mul(curve: int, group: int, scalars: np.ndarray, base=None):
    """[s_i] * base (default: the standard generator) -> (points (n, L), inf (n,))"""
    scalars = _u64(scalars)
    n = scalars.shape[0]
    out = np.zeros((n, point_limbs(curve, group)), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    rc = lib().zko_batch_mul(curve, group, _p(_u64(base)) if base is not None else None, _p(scalars),
                             ctypes.c_size_t(n), _p(out), _p(inf))
    assert rc == 0
    return out, inf


def msm(curve: int, group: int, bases: np.ndarray, scalars: np.ndarray, inf=None, chunks: int = 1, naive=False):
    bases = _u64(bases)
    scalars = _u64(scalars)
    n = scalars.shape[0]
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    infp = _p(np.ascontiguousarray(inf, dtype=np.uint8)) if inf is not None else None
    if naive:
        rc = lib().zko_msm_naive(curve, group, _p(bases), infp, _p(scalars), ctypes.c_size_t(n), _p(out), _p(oinf))
    else:
        rc = lib().zko_msm(curve, group, _p(bases), infp, _p(scalars), ctypes.c_size_t(n), chunks, _p(out), _p(oinf))
    assert rc == 0
    return out, int(oinf[0])


class Bases:
    """Pre-converted (Montgomery) bases resident in host memory, for timing the CPU baseline."""

    def __init__(self, curve, group, bases, inf=None):
        bases = _u64(bases)
        self.curve, self.group, self.n = curve, group, bases.shape[0]
        infp = _p(np.ascontiguousarray(inf, dtype=np.uint8)) if inf is not None else None
        self.h = ctypes.c_void_p(lib().zko_bases_new(curve, group, _p(bases), infp, ctypes.c_size_t(self.n)))
        assert self.h

    def msm(self, scalars, chunks=1, off=0, n=None):
        scalars = _u64(scalars)
        n = scalars.shape[0] if n is None else n
        out = np.zeros(point_limbs(self.curve, self.group), dtype=np.uint64)
        oinf = np.zeros(1, dtype=np.uint8)
        rc = lib().zko_msm_bases(self.h, ctypes.c_size_t(off), ctypes.c_size_t(n), _p(scalars), chunks, _p(out), _p(oinf))
        assert rc == 0
        return out, int(oinf[0])

    def __del__(self):
        if getattr(self, "h", None):
            lib().zko_bases_free(self.h)
            self.h = None


def jac_to_affine(curve: int, group: int, jac: np.ndarray):
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    assert lib().zko_jac_to_affine(curve, group, _p(_u64(jac)), _p(out), _p(oinf)) == 0
    return out, int(oinf[0])


def point_add(curve, group, a, a_inf, b, b_inf):
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    assert lib().zko_point_add(curve, group, _p(_u64(a)), int(a_inf), _p(_u64(b)), int(b_inf), _p(out), _p(oinf)) == 0
    return out, int(oinf[0])


def ntt(curve: int, data: np.ndarray, log_m: int, omega: np.ndarray, inverse=False, coset=None) -> np.ndarray:
    """data: (batch, m, 4) canonical; returns a transformed copy."""
    d = _u64(data).copy()
    batch = d.shape[0] if d.ndim == 3 else 1
    rc = lib().zko_ntt(curve, _p(d), ctypes.c_size_t(log_m), ctypes.c_size_t(batch), _p(_u64(omega)),
                       1 if inverse else 0, _p(_u64(coset)) if coset is not None else None)
    assert rc == 0
    return d


def domain_choice(min_size: int, two_adicity: int):
    """(kind, m) of make_evaluation_domain(min_size): 0 basic / 1 extended / 2 step radix-2"""
    out = np.zeros(2, dtype=np.uint64)
    assert lib().zko_domain_choice(ctypes.c_size_t(min_size), ctypes.c_size_t(two_adicity), _p(out)) == 0
    return int(out[0]), int(out[1])


def domain_fft(curve: int, kind: int, data: np.ndarray, omega, shift=None, inverse=False) -> np.ndarray:
    """one vector (m, 4) over the domain (kind, m = len(data)); returns a transformed copy"""
    d = _u64(data).copy()
    rc = lib().zko_domain_fft(curve, int(kind), ctypes.c_size_t(d.shape[0]), _p(_u64(omega)), _p(_u64(shift)) if shift is not None else None,
                              _p(d), 1 if inverse else 0)
    assert rc == 0
    return d


def fr_horner(curve: int, coeffs: np.ndarray, x: np.ndarray) -> np.ndarray:
    coeffs = _u64(coeffs)
    out = np.zeros(4, dtype=np.uint64)
    lib().zko_fr_horner(curve, _p(coeffs), ctypes.c_size_t(coeffs.shape[0]), _p(_u64(x)), _p(out))
    return out


def fr_mul(curve: int, a, b) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    lib().zko_fr_mul(curve, _p(_u64(a)), _p(_u64(b)), _p(out))
    return out


class Groth16:
    """Example R1CS (r1cs_examples.hpp:77-140) + fixed-trapdoor key + prover, all on the CPU."""

    def __init__(self, curve: int, num_constraints: int, num_inputs: int, seed: int):
        self.curve = curve
        self.h = ctypes.c_void_p(lib().zko_g16_new(curve, ctypes.c_size_t(num_constraints),
                                                   ctypes.c_size_t(num_inputs), ctypes.c_uint64(seed)))
        assert self.h
        self._dims()

    def _dims(self):
        d = np.zeros(8, dtype=np.uint64)
        lib().zko_g16_dims(self.h, _p(d))
        (self.M, self.n, self.N, self.m, self.log_m, self.nnzA, self.nnzB, self.nnzC) = (int(x) for x in d)

    def is_satisfied(self) -> bool:
        return bool(lib().zko_g16_is_satisfied(self.h))

    def set_domain(self, kind: int, m: int, omega, shift=None):
        """install the evaluation domain every later call reduces over (default: the basic radix-2 domain of 2^ceil(log2(M+n+1))
        points over the omega each call passes); kind / omega / shift as pyoracle.EvaluationDomain"""
        assert lib().zko_g16_set_domain(self.h, int(kind), ctypes.c_size_t(m), _p(_u64(omega)), _p(_u64(shift)) if shift is not None else None) == 0
        self._dims()

    def keygen(self, trapdoor: np.ndarray, omega: np.ndarray):
        assert lib().zko_g16_keygen(self.h, _p(_u64(trapdoor)), _p(_u64(omega))) == 0
        self._dims()

    def expected_exponents(self, trapdoor, omega, r, s) -> np.ndarray:
        """(a, b, c) with A = a G1, B = b G2, C = c G1 the proof a correct prover outputs under the key of `trapdoor` with
        blinders (r, s): from the trapdoor identities alone, no MSM / NTT / group arithmetic.  Swaps A / B like keygen."""
        out = np.zeros((3, 4), dtype=np.uint64)
        assert lib().zko_g16_expected_exponents(self.h, _p(_u64(trapdoor)), _p(_u64(omega)), _p(_u64(r)), _p(_u64(s)), _p(out)) == 0
        return out

    def csr(self, which: int):
        nnz = (self.nnzA, self.nnzB, self.nnzC)[which]
        rowptr = np.zeros(self.M + 1, dtype=np.uint32)
        col = np.zeros(nnz, dtype=np.uint32)
        coeff = np.zeros((nnz, 4), dtype=np.uint64)
        lib().zko_g16_get_csr(self.h, which, _p(rowptr), _p(col), _p(coeff))
        return rowptr, col, coeff

    def assignment(self) -> np.ndarray:
        out = np.zeros((self.N, 4), dtype=np.uint64)
        lib().zko_g16_get_assignment(self.h, _p(out))
        return out

    def query(self, which: int):
        count = {0: self.N + 1, 1: self.N + 1, 2: self.N + 1, 3: self.m - 1, 4: self.N - self.n, 5: 3, 6: 2}[which]
        group = G2 if which in (2, 6) else G1
        out = np.zeros((count, point_limbs(self.curve, group)), dtype=np.uint64)
        inf = np.zeros(count, dtype=np.uint8)
        assert lib().zko_g16_get_query(self.h, which, _p(out), _p(inf)) == 0
        return out, inf

    def witness_map(self, omega, coset) -> np.ndarray:
        out = np.zeros((self.m + 1, 4), dtype=np.uint64)
        lib().zko_g16_witness_map(self.h, _p(_u64(omega)), _p(_u64(coset)), _p(out))
        return out

    def prove(self, r, s, omega, coset, chunks=1) -> np.ndarray:
        """-> flat canonical limbs: A (G1) | B (G2) | C (G1)."""
        L1 = point_limbs(self.curve, G1)
        L2 = point_limbs(self.curve, G2)
        out = np.zeros(2 * L1 + L2, dtype=np.uint64)
        assert lib().zko_g16_prove(self.h, _p(_u64(r)), _p(_u64(s)), _p(_u64(omega)), _p(_u64(coset)), chunks, _p(out)) == 0
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().zko_g16_free(self.h)
            self.h = None


# ---------------------------------------------------------------------------------------------------------------------------
# placeholder's permutation / lookup / gate arguments and the quotient at sizes pyoracle's dense O(n^2) arithmetic cannot reach
# (VERDICT r4 #4).  The orchestration below follows pyoracle.permutation_argument / lookup_argument / gate_argument_dfs /
# quotient_polynomial statement by statement -- same products, same order -- over numpy limb arrays, with the arithmetic in
# liboracle.so (transform-based products, the reference's row-by-row recurrences with one inversion per row).  PINNED to pyoracle at
# <= 2^8 rows (tests/test_oracle_kat.py::test_fast_argument_oracle_equals_dense_oracle); pyoracle itself is pinned to what the
# arguments are for.  Polynomials are (len, 4) uint64 arrays of canonical limbs: DFS vectors or coefficient vectors as named.
# ---------------------------------------------------------------------------------------------------------------------------
_R = {BLS12_381: 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
      BN254: 21888242871839275222246405745257275088548364400416034343698204186575808495617}
_GEN = {BLS12_381: 7, BN254: 5}


def _limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def fr_root(curve: int, log_n: int) -> np.ndarray:
    """the primitive 2^log_n-th root of unity generator^((r - 1) / 2^log_n) (crypto3-algebra's arithmetic_params; pyoracle.Curve.root_of_unity)"""
    return _limbs(pow(_GEN[curve], (_R[curve] - 1) >> log_n, _R[curve]))


def fr_vec(curve: int, op: int, a, b, c=None) -> np.ndarray:
    """pointwise: 0 a + b, 1 a - b, 2 a b, 3 a * scalar b, 4 a b / c"""
    a = _u64(a)
    out = np.empty_like(a)
    assert lib().zko_fr_vec(curve, op, _p(a), _p(_u64(b)), _p(_u64(c)) if c is not None else None, _p(out), ctypes.c_size_t(a.shape[0])) == 0
    return out


def ntt_wide(curve: int, data, inverse=False) -> np.ndarray:
    d = _u64(data).copy()
    log_m = d.shape[0].bit_length() - 1
    assert d.shape[0] == 1 << log_m
    assert lib().zko_ntt_wide(curve, _p(d), ctypes.c_size_t(log_m), _p(fr_root(curve, log_m)), 1 if inverse else 0) == 0
    return d


def poly_trim(a) -> np.ndarray:
    a = _u64(a).reshape(-1, 4)
    nz = np.nonzero(a.any(axis=1))[0]
    return a[: (int(nz[-1]) + 1 if nz.size else 0)]


def _pad(a, size):
    a = _u64(a).reshape(-1, 4)
    if a.shape[0] >= size:
        return a
    return np.concatenate([a, np.zeros((size - a.shape[0], 4), dtype=np.uint64)])


def poly_add(curve, a, b):
    m = max(len(a), len(b))
    return fr_vec(curve, 0, _pad(a, m), _pad(b, m)) if m else np.zeros((0, 4), dtype=np.uint64)


def poly_sub(curve, a, b):
    m = max(len(a), len(b))
    return fr_vec(curve, 1, _pad(a, m), _pad(b, m)) if m else np.zeros((0, 4), dtype=np.uint64)


def poly_scale(curve, a, c: int):
    return fr_vec(curve, 3, a, _limbs(c % _R[curve])) if len(a) else _u64(a).reshape(0, 4)


def poly_mul(curve, a, b) -> np.ndarray:
    a, b = poly_trim(a), poly_trim(b)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    no = len(a) + len(b) - 1
    log_m = max(1, (no - 1).bit_length())
    out = np.zeros((no, 4), dtype=np.uint64)
    assert lib().zko_poly_mul(curve, _p(a), ctypes.c_size_t(len(a)), _p(b), ctypes.c_size_t(len(b)), _p(fr_root(curve, log_m)), ctypes.c_size_t(log_m), _p(out)) == 0
    return out


def dfs_coefficients(curve, evals) -> np.ndarray:
    """trimmed coefficients of the polynomial a DFS vector (any power-of-two size) holds"""
    return poly_trim(ntt_wide(curve, evals, inverse=True))


def polynomial_shift(evals, shift: int, domain_size: int = 0) -> np.ndarray:
    """math::polynomial_shift on a DFS vector: f(omega^shift X), omega the generator of the domain_size-point domain"""
    e = _u64(evals)
    n = e.shape[0]
    return np.roll(e, -shift * (n // (domain_size or n)), axis=0)


def reduce_dfs_polynomial_domain(evals, new_size: int) -> np.ndarray:
    e = _u64(evals)
    assert e.shape[0] % new_size == 0
    return np.ascontiguousarray(e[:: e.shape[0] // new_size])


def dfs_resize(curve, evals, new_size: int) -> np.ndarray:
    """polynomial_dfs::resize (pyoracle.dfs_resize): coefficients over the old domain, evaluations over the new power-of-two domain
    (a smaller one takes every k-th evaluation -- the caller knows the degree fits)"""
    e = _u64(evals)
    n = e.shape[0]
    if new_size == n:
        return e.copy()
    if new_size < n:
        return reduce_dfs_polynomial_domain(e, new_size)
    return ntt_wide(curve, _pad(ntt_wide(curve, e, inverse=True), new_size))


def gate_argument_dfs(curve, products, mask, extended_size: int) -> np.ndarray:
    """pyoracle.gate_argument_dfs (gates_argument.hpp:203-216) statement by statement over limb arrays: products = [(coefficient, [(evals,
    rotation), ...]), ...] over the ORIGINAL domain; every factor shifted there (:108-110), resized to the extended domain (:111-113),
    multiplied pointwise; the weighted sum times the extended mask (:215).  Pinned to pyoracle at <= 2^8 rows in tests/test_oracle_kat.py."""
    F = np.zeros((extended_size, 4), dtype=np.uint64)
    for coeff, factors in products:
        term = np.repeat(_ONE, extended_size, axis=0)
        for evals, rot in factors:
            e = polynomial_shift(evals, rot) if rot else _u64(evals)
            term = fr_vec(curve, 2, term, dfs_resize(curve, e, extended_size))
        F = fr_vec(curve, 0, F, fr_vec(curve, 3, term, _limbs(coeff % _R[curve])))
    return fr_vec(curve, 2, F, dfs_resize(curve, mask, extended_size))


_ONE = np.array([[1, 0, 0, 0]], dtype=np.uint64)


def permutation_grand_product(curve, cols, S_id, S_sigma, beta: int, gamma: int):
    """permutation_argument.hpp:103-136 -> (g_v (k, n, 4), h_v, V_P (n, 4))"""
    k, n = len(cols), len(cols[0])
    g, h, v = np.zeros((k, n, 4), dtype=np.uint64), np.zeros((k, n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)
    assert lib().zko_perm_grand_product(curve, ctypes.c_size_t(k), ctypes.c_size_t(n), _p(_u64(np.stack(cols))), _p(_u64(np.stack(S_id))), _p(_u64(np.stack(S_sigma))),
                                        _p(_limbs(beta)), _p(_limbs(gamma)), _p(g), _p(h), _p(v)) == 0
    return g, h, v


def permutation_argument(curve, cols, S_id, S_sigma, q_last, q_blind, lagrange_0, beta: int, gamma: int, max_quotient_chunks: int = 0, alphas=(), usable_rows=None):
    """pyoracle.permutation_argument over arrays -> (V_P, [F0, F1, F2] trimmed coefficients[, intermediate polynomials])"""
    co = lambda e: dfs_coefficients(curve, e)
    mul, add, sub = (lambda a, b: poly_mul(curve, a, b)), (lambda a, b: poly_add(curve, a, b)), (lambda a, b: poly_sub(curve, a, b))
    g, h, V = permutation_grand_product(curve, cols, S_id, S_sigma, beta, gamma)
    VP, VPs = co(V), co(polynomial_shift(V, 1))
    F0 = mul(co(lagrange_0), sub(_ONE, VP))
    F2 = mul(co(q_last), mul(VP, sub(VP, _ONE)))
    q = add(co(q_last), co(q_blind))
    if max_quotient_chunks == 0:
        G, H = _ONE, _ONE
        for gi, hi in zip(g, h):
            G, H = mul(G, co(gi)), mul(H, co(hi))
        F1 = mul(sub(_ONE, q), sub(mul(VPs, H), mul(VP, G)))
        return V, [poly_trim(F0), poly_trim(F1), poly_trim(F2)]
    step = max_quotient_chunks - 1
    groups = [(g[i:i + step], h[i:i + step]) for i in range(0, len(g), step)]
    assert len(alphas) == len(groups) - 1
    prev, currents, acc = V, [], np.zeros((0, 4), dtype=np.uint64)
    for idx, (gg, hh) in enumerate(groups):
        G, H = _ONE, _ONE
        for gi, hi in zip(gg, hh):
            G, H = mul(G, co(gi)), mul(H, co(hi))
        if idx < len(groups) - 1:
            nom, den = gg[0], hh[0]
            for gi, hi in zip(gg[1:], hh[1:]):
                nom, den = fr_vec(curve, 2, nom, gi), fr_vec(curve, 2, den, hi)
            cur = V.copy()
            cur[:usable_rows] = fr_vec(curve, 4, prev[:usable_rows], nom[:usable_rows], den[:usable_rows])
            currents.append(cur)
            acc = add(acc, poly_scale(curve, sub(mul(co(prev), G), mul(co(cur), H)), alphas[idx]))
            prev = cur
        else:
            acc = add(acc, sub(mul(co(prev), G), mul(VPs, H)))
    F1 = mul(sub(q, _ONE), acc)
    return V, [poly_trim(F0), poly_trim(F1), poly_trim(F2)], currents


def lookup_sort_polynomials(reduced_input, reduced_value, domain_size: int, usable_rows: int):
    """pyoracle.lookup_sort_polynomials (strict) over arrays: the reference's map + walk, keys = the 32 bytes of an element"""
    key = lambda v, j: v[j].tobytes()
    zero = bytes(32)
    count = {}
    for v in reduced_value:
        for j in range(usable_rows):
            count[key(v, j)] = count.get(key(v, j), 0) + 1
    for v in reduced_input:
        for j in range(usable_rows):
            assert key(v, j) in count, "a looked-up value that is in no table"
            count[key(v, j)] += 1
    total = len(reduced_input) + len(reduced_value)
    flat, prev = [], zero
    for v in reduced_value:
        for j in range(usable_rows):
            kj = key(v, j)
            if kj != prev:
                flat.extend([prev] if prev == zero else [prev] * count[prev])
                prev = kj
    if prev != zero:
        flat.extend([prev] * count[prev])
    assert len(flat) <= total * usable_rows
    out = np.zeros((total, domain_size, 4), dtype=np.uint64)
    seq = np.frombuffer(b"".join(flat), dtype=np.uint64).reshape(-1, 4) if flat else np.zeros((0, 4), dtype=np.uint64)
    for i in range(total):
        part = seq[i * usable_rows:(i + 1) * usable_rows]
        out[i, :len(part)] = part
    for i in range(total - 1):
        out[i, usable_rows] = out[i + 1, 0]
    return [out[i] for i in range(total)]


def lookup_grand_product(curve, reduced_input, reduced_value, sorted_, beta: int, gamma: int, usable_rows: int) -> np.ndarray:
    n = len(sorted_[0])
    stack = lambda vs: _u64(np.stack(vs)) if len(vs) else np.zeros((0, n, 4), dtype=np.uint64)
    v = np.zeros((n, 4), dtype=np.uint64)
    assert lib().zko_lookup_grand_product(curve, ctypes.c_size_t(len(reduced_input)), ctypes.c_size_t(len(reduced_value)), ctypes.c_size_t(len(sorted_)), ctypes.c_size_t(n),
                                          ctypes.c_size_t(usable_rows), _p(stack(reduced_input)), _p(stack(reduced_value)), _p(stack(sorted_)), _p(_limbs(beta)),
                                          _p(_limbs(gamma)), _p(v)) == 0
    return v


def lookup_argument(curve, lookup_input, lookup_value, sorted_, q_last, q_blind, lagrange_0, beta: int, gamma: int, alphas, usable_rows: int, part_sizes=None,
                    part_alphas=()):
    """pyoracle.lookup_argument over arrays -> (V_L, [F0 .. F3] trimmed coefficients[, intermediate polynomials])"""
    r = _R[curve]
    n = len(sorted_[0])
    co = lambda e: dfs_coefficients(curve, e)
    mul, add, sub = (lambda a, b: poly_mul(curve, a, b)), (lambda a, b: poly_add(curve, a, b)), (lambda a, b: poly_sub(curve, a, b))
    scale = lambda a, c: poly_scale(curve, a, c)
    red_in = [reduce_dfs_polynomial_domain(v, n) for v in lookup_input]
    red_val = [reduce_dfs_polynomial_domain(v, n) for v in lookup_value]
    V = lookup_grand_product(curve, red_in, red_val, sorted_, beta, gamma, usable_rows)
    part1 = (1 + beta) * gamma % r
    P1 = _limbs(part1).reshape(1, 4)
    G1 = _limbs(gamma).reshape(1, 4)
    nxt = lambda v: polynomial_shift(v, 1, n)
    rows = lambda rv, const: fr_vec(curve, 0, fr_vec(curve, 0, rv, np.broadcast_to(const, rv.shape).copy()), poly_scale(curve, np.roll(rv, -1, axis=0), beta))
    # the factors as (coefficients, values on the rows)
    g_f = [(scale(add(G1, co(v)), (1 + beta) % r), poly_scale(curve, fr_vec(curve, 0, rv, np.broadcast_to(G1, rv.shape).copy()), (1 + beta) % r)) for v, rv in zip(lookup_input, red_in)]
    g_f += [(add(add(P1, co(v)), scale(co(nxt(v)), beta)), rows(rv, P1)) for v, rv in zip(lookup_value, red_val)]
    h_f = [(add(add(P1, co(v)), scale(co(nxt(v)), beta)), rows(_u64(v), P1)) for v in sorted_]
    VL, VLs = co(V), co(polynomial_shift(V, 1))
    L0 = co(lagrange_0)
    F0 = mul(L0, sub(_ONE, VL))
    F1 = mul(co(q_last), sub(mul(VL, VL), VL))
    sizes = list(part_sizes) if part_sizes is not None else [len(sorted_)]
    assert sum(sizes) == len(sorted_) == len(g_f) and len(part_alphas) == len(sizes) - 1
    prev, currents, acc, at = V, [], np.zeros((0, 4), dtype=np.uint64), 0
    for idx, sz in enumerate(sizes):
        G, H = _ONE, _ONE
        for (gc, _), (hc, _) in zip(g_f[at:at + sz], h_f[at:at + sz]):
            G, H = mul(G, gc), mul(H, hc)
        if idx < len(sizes) - 1:
            nom, den = g_f[at][1], h_f[at][1]
            for (_, gv), (_, hv) in zip(g_f[at + 1:at + sz], h_f[at + 1:at + sz]):
                nom, den = fr_vec(curve, 2, nom, gv), fr_vec(curve, 2, den, hv)
            cur = V.copy()
            cur[:usable_rows] = fr_vec(curve, 4, prev[:usable_rows], nom[:usable_rows], den[:usable_rows])
            currents.append(cur)
            acc = add(acc, scale(sub(mul(co(prev), G), mul(co(cur), H)), part_alphas[idx]))
            prev = cur
        else:
            acc = add(acc, sub(mul(co(prev), G), mul(VLs, H)))
        at += sz
    F2 = mul(sub(add(co(q_last), co(q_blind)), _ONE), acc)
    F3 = np.zeros((0, 4), dtype=np.uint64)
    for i in range(len(sorted_) - 1):
        d = sub(co(sorted_[i + 1]), co(polynomial_shift(sorted_[i], usable_rows, n)))
        F3 = add(F3, scale(mul(d, L0), alphas[i]))
    F = [poly_trim(F0), poly_trim(F1), poly_trim(F2), poly_trim(F3)]
    return (V, F) if part_sizes is None else (V, F, currents)


def quotient_polynomial(curve, F_coeffs, alphas, n: int):
    """pyoracle.quotient_polynomial over coefficient arrays: sum_i alpha_i F_i divided EXACTLY by X^n - 1 (prover.hpp:262-277)"""
    acc = np.zeros((0, 4), dtype=np.uint64)
    for f, a in zip(F_coeffs, alphas):
        acc = poly_add(curve, acc, poly_scale(curve, f, a))
    acc = poly_trim(acc)
    if len(acc) <= n:
        assert len(acc) == 0, "the constraints do not vanish on the rows"
        return acc
    q = np.zeros((len(acc) - n, 4), dtype=np.uint64)
    nz = ctypes.c_uint64()
    assert lib().zko_poly_div_vanishing(curve, _p(acc), ctypes.c_size_t(len(acc)), ctypes.c_size_t(n), _p(q), ctypes.byref(nz)) == 0
    assert nz.value == 0, "the constraints do not vanish on the rows"
    return poly_trim(q)


# ---------------------------------------------------------------------------------------------------------------------------
# The commitment SCHEMES at sizes pyoracle's dense O(n^2) arithmetic cannot reach: LPC's leaf layout and opening
# proof through the FRI commit phase, the two batched KZG opening proofs.  Each function follows its pyoracle namesake statement by
# statement -- same sums, same divisions, same order -- over (len, 4) uint64 limb arrays; products by transforms (poly_mul), divisions
# by linear factors as synthetic divisions (exact: the remainders are the reference's BOOST_ASSERTs and are asserted here), the few-
# coefficient polynomials (U, V, the difference polynomials) in Python integers through pyoracle's own helpers.  PINNED to pyoracle at
# <= 2^8 in tests/test_oracle_kat.py::test_fast_scheme_oracle_*.
# ---------------------------------------------------------------------------------------------------------------------------
def _ints(a):
    return [int(x[0]) | (int(x[1]) << 64) | (int(x[2]) << 128) | (int(x[3]) << 192) for x in _u64(a).reshape(-1, 4)]


def _arr(vals):
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64).reshape(len(vals), 4)


def poly_eval(curve, coeffs, x: int) -> int:
    c = _u64(coeffs).reshape(-1, 4)
    if len(c) == 0:
        return 0
    return _ints(fr_horner(curve, c, _limbs(x % _R[curve])).reshape(1, 4))[0]


def poly_div_linear(curve, f, z: int):
    """(quotient, remainder) of f / (X - z), coefficient arrays"""
    f = _u64(f).reshape(-1, 4)
    if len(f) == 0:
        return f, 0
    q, rem = np.zeros((len(f) - 1, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    assert lib().zko_poly_div_linear(curve, _p(f), ctypes.c_size_t(len(f)), _p(_limbs(z % _R[curve])), _p(q), _p(rem)) == 0
    return q, _ints(rem.reshape(1, 4))[0]


def poly_div_exact_by_roots(curve, f, roots):
    """f / prod (X - x) for x in roots, asserting every remainder (pyoracle: poly_divmod + `assert not rem`)"""
    for x in roots:
        f, rem = poly_div_linear(curve, f, x)
        assert rem == 0, "the division is not exact"
    return f


def fri_leaves(polys_on_D, fri_step: int) -> np.ndarray:
    """pyoracle.fri_leaves (basic_fri.hpp:456-492, FRI::m = 2) as an index computation: the offsets of a leaf's pairs relative to x do not
    depend on x"""
    polys = [_u64(p).reshape(-1, 4) for p in polys_on_D]
    D, coset = polys[0].shape[0], 1 << fri_step
    off = [0] * (coset // 2)
    base, prev_half, i = D // 4, 1, 1
    while i < coset // 2:
        for j in range(prev_half):
            off[i] = (base + off[j]) % D
            i += 1
        base //= 2
        prev_half <<= 1
    x = np.arange(D // coset, dtype=np.int64)[:, None]
    first = (x + np.array(off, dtype=np.int64)[None, :]) % D                       # (leaves, coset / 2)
    idx = np.stack([first, (first + D // 2) % D], axis=2).reshape(D // coset, coset)    # the pairs (s, s + D / 2), in order
    out = np.stack([p[idx] for p in polys], axis=1)                                # (leaves, polys, coset, 4)
    return np.ascontiguousarray(out.reshape(-1, 4))


def fold_polynomial_dfs(curve, f, alpha: int, omega: int) -> np.ndarray:
    f = _u64(f).reshape(-1, 4)
    log_size = f.shape[0].bit_length() - 1
    out = np.zeros((f.shape[0] // 2, 4), dtype=np.uint64)
    assert lib().zko_fri_fold(curve, _p(f), ctypes.c_size_t(log_size), _p(_limbs(alpha % _R[curve])), _p(_limbs(omega % _R[curve])), _p(out)) == 0
    return out


def toy_root(curve):
    """the tests' stand-in for the caller's Merkle tree, on limb arrays: (per_leaf + sum_i (i + 1) v_i) mod r"""
    def root(leaves, per_leaf: int) -> int:
        a = _u64(leaves).reshape(-1, 4)
        out = np.zeros(4, dtype=np.uint64)
        assert lib().zko_toy_root(curve, _p(a), ctypes.c_size_t(a.shape[0]), ctypes.c_uint64(per_leaf), _p(out)) == 0
        return _ints(out.reshape(1, 4))[0]
    return root


def _root_int(curve, log_n: int) -> int:
    return pow(_GEN[curve], (_R[curve] - 1) >> log_n, _R[curve])


def lpc_proof_eval(curve, batches: dict, points: dict, fixed, log_domain: int, step_list, challenges, tree_root):
    """pyoracle.lpc_proof_eval (lpc.hpp:101-200 + basic_fri.hpp:433-496, 705-742 + fold_polynomial.hpp:68-93) over limb arrays.
    batches[k]: DFS polynomials as (2^l, 4) arrays; -> (roots {k: int}, z {k: [[int]]}, fri_roots [int], final coefficients array)"""
    r = _R[curve]
    D = 1 << log_domain
    ch = iter(challenges)
    coeffs = {k: [ntt_wide(curve, p, inverse=True) for p in ps] for k, ps in batches.items()}
    roots = {}
    for k in sorted(batches):
        ext = [dfs_resize(curve, p, D) for p in batches[k]]
        roots[k] = tree_root(fri_leaves(ext, step_list[0]), len(ext) << step_list[0])
    etha = next(ch)
    fixed_vals = {k: [poly_eval(curve, c, etha) for c in coeffs[k]] for k in fixed}
    assert next(ch) == etha
    z = {k: [[poly_eval(curve, coeffs[k][i], x) for x in points[k][i]] for i in range(len(coeffs[k]))] for k in sorted(batches)}
    theta = next(ch)
    uniq = []
    for k in sorted(points):
        for pl in points[k]:
            for x in pl:
                if x not in uniq:
                    uniq.append(x)
    theta_acc = 1
    combined = np.zeros((0, 4), dtype=np.uint64)

    def sub_const(q, c):
        return poly_sub(curve, q, _arr([c % r]))

    for pt in uniq:
        q = np.zeros((0, 4), dtype=np.uint64)
        for k in sorted(batches):
            for i, c in enumerate(coeffs[k]):
                if pt not in points[k][i]:
                    continue
                zi = z[k][i][points[k][i].index(pt)]
                q = poly_add(curve, q, poly_scale(curve, c, theta_acc))
                q = sub_const(q, zi * theta_acc)
                theta_acc = theta_acc * theta % r
        qq, rem = poly_div_linear(curve, q, pt)
        assert rem == 0
        combined = poly_add(curve, combined, qq)
    for k in sorted(batches):
        if k not in fixed:
            continue
        q = np.zeros((0, 4), dtype=np.uint64)
        for i, c in enumerate(coeffs[k]):
            q = poly_add(curve, q, poly_scale(curve, c, theta_acc))
            q = sub_const(q, fixed_vals[k][i] * theta_acc)
            theta_acc = theta_acc * theta % r
        qq, rem = poly_div_linear(curve, q, etha)
        assert rem == 0
        combined = poly_add(curve, combined, qq)
    f = ntt_wide(curve, _pad(combined, D))
    pre = tree_root(fri_leaves([f], step_list[0]), 1 << step_list[0])
    fri_roots, t = [], 0
    for i, step in enumerate(step_list):
        fri_roots.append(pre)
        for _ in range(step):
            f = fold_polynomial_dfs(curve, f, next(ch), _root_int(curve, log_domain - t))
            t += 1
        if i != len(step_list) - 1:
            pre = tree_root(fri_leaves([f], step_list[i + 1]), 1 << step_list[i + 1])
    final = ntt_wide(curve, f, inverse=True)
    return roots, z, fri_roots, final


def _small(curve):
    import pyoracle as po
    return po


def kzg_v2_proof_eval(curve, polys: dict, points: dict, theta: int, theta2: int):
    """pyoracle.kzg_v2_proof_eval (kzg_v2.hpp:236-305) over limb arrays: polys[k][i] coefficient arrays -> (z, f, L) with f, L coefficient
    arrays (the quotients committed as pi_1, pi_2)"""
    po = _small(curve)
    r = _R[curve]
    z = {k: [[poly_eval(curve, p, x) for x in points[k][i]] for i, p in enumerate(ps)] for k, ps in polys.items()}
    merged = sorted({x for k in points for pl in points[k] for x in pl})
    V = po.vanishing_poly(merged, r)

    def diffpoly(pts):
        return po.vanishing_poly([x for x in merged if x not in pts], r)

    theta_i = 1
    f = np.zeros((0, 4), dtype=np.uint64)
    for k in sorted(polys):
        for i, p in enumerate(polys[k]):
            U = po.lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            f = poly_add(curve, f, poly_scale(curve, poly_mul(curve, poly_sub(curve, p, _arr(U)), _arr(diffpoly(points[k][i]))), theta_i))
            theta_i = theta_i * theta % r
    f = poly_div_exact_by_roots(curve, f, merged)                                   # f / V, BOOST_ASSERT(f % V == 0) (:266)
    theta_i = 1
    L = np.zeros((0, 4), dtype=np.uint64)
    for k in sorted(polys):
        for i, p in enumerate(polys[k]):
            U = po.lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            zts = po.poly_eval(diffpoly(points[k][i]), theta2, r)
            L = poly_add(curve, L, poly_scale(curve, poly_sub(curve, p, _arr([po.poly_eval(U, theta2, r)])), theta_i * zts % r))
            theta_i = theta_i * theta % r
    L = poly_sub(curve, L, poly_scale(curve, f, po.poly_eval(V, theta2, r)))
    L, rem = poly_div_linear(curve, L, theta2)                                      # (:290-291)
    assert rem == 0
    return z, poly_trim(f), poly_trim(L)


def kzg_v1_proof_eval(curve, polys: dict, points: dict, gamma: int):
    """pyoracle.kzg_v1_proof_eval (kzg.hpp:782-807) over limb arrays -> (z, accum)"""
    po = _small(curve)
    r = _R[curve]
    z = {k: [[poly_eval(curve, p, x) for x in points[k][i]] for i, p in enumerate(ps)] for k, ps in polys.items()}
    factor = 1
    accum = np.zeros((0, 4), dtype=np.uint64)
    for k in sorted(polys):
        for i, p in enumerate(polys[k]):
            U = po.lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            q = poly_div_exact_by_roots(curve, poly_sub(curve, p, _arr(U)), points[k][i])
            accum = poly_add(curve, accum, poly_scale(curve, q, factor))
            factor = factor * gamma % r
    return z, poly_trim(accum)
