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


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.zko_g16_new.restype = ctypes.c_void_p
        _LIB.zko_bases_new.restype = ctypes.c_void_p
    return _LIB


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
