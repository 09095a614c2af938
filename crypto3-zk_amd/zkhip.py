"""ctypes binding of include/zkhip.h (plumbing for tests/ and bench.py; not a compute path)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BLS12_381, BN254 = 0, 1
G1, G2 = 1, 2
_FQ = {BLS12_381: 6, BN254: 4}

EXPORTS = [
    "zkhip_init", "zkhip_destroy", "zkhip_strerror", "zkhip_last_error", "zkhip_set_stream", "zkhip_sync", "zkhip_device_status", "zkhip_stream_wait", "zkhip_device",
    "zkhip_set_option", "zkhip_get_option", "zkhip_malloc", "zkhip_free", "zkhip_memcpy_h2d", "zkhip_memcpy_d2h", "zkhip_memcpy_h2d_async", "zkhip_memcpy_d2h_async", "zkhip_memcpy_d2d_async", "zkhip_memcpy_2d_d2d_async", "zkhip_host_alloc", "zkhip_host_free",
    "zkhip_bases_upload", "zkhip_bases_upload_compressed", "zkhip_bases_from_scalars", "zkhip_bases_spread", "zkhip_bases_download", "zkhip_bases_size",
    "zkhip_bases_free", "zkhip_msm", "zkhip_msm_dev", "zkhip_msm_batch_dev", "zkhip_jacobian_sum_dev", "zkhip_jacobian_to_affine", "zkhip_ntt", "zkhip_ntt_dev",
    "zkhip_domain_choice", "zkhip_domain_fft_dev", "zkhip_domain_lagrange_dev",
    "zkhip_r1cs_upload", "zkhip_r1cs_free", "zkhip_r1cs_set_domain", "zkhip_r1cs_domain_size", "zkhip_r1cs_domain_kind", "zkhip_groth16_scratch_bytes", "zkhip_groth16_witness_h_dev", "zkhip_groth16_witness_h_domain_dev", "zkhip_fr_gather_dev", "zkhip_poly_resize_dev", "zkhip_fri_fold_dev", "zkhip_fri_leaves_dev", "zkhip_ec_ntt_dev",
    "zkhip_fr_vec_op_dev", "zkhip_fr_vec_affine_dev", "zkhip_fr_vec_mul_div_dev", "zkhip_fr_vec_prod_dev", "zkhip_poly_shift_dev", "zkhip_poly_eval_dev", "zkhip_poly_div_linear_dev", "zkhip_poly_div_vanishing_dev", "zkhip_poly_lincomb_dev", "zkhip_perm_grand_product_dev", "zkhip_lookup_grand_product_dev", "zkhip_lookup_sort_dev", "zkhip_perm_factor_products_dev", "zkhip_gate_eval_dev",
    "zkhip_group_init", "zkhip_group_destroy", "zkhip_group_size", "zkhip_group_ctx", "zkhip_group_last_error", "zkhip_group_set_transport", "zkhip_group_transport",
    "zkhip_group_all_gather", "zkhip_group_copy", "zkhip_group_sync", "zkhip_group_bases_upload", "zkhip_group_bases_from_scalars", "zkhip_group_bases_free",
    "zkhip_group_bases_size", "zkhip_group_bases_member", "zkhip_group_msm", "zkhip_group_ntt",
    "zkhip_profile_enable", "zkhip_profile_reset", "zkhip_profile_filter", "zkhip_profile_get", "zkhip_profile_dump",
]


class ZkhipError(RuntimeError):
    pass


DOMAIN_BASIC, DOMAIN_EXTENDED, DOMAIN_STEP = 0, 1, 2


class Domain(ctypes.Structure):
    """zkhip_domain"""
    _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_uint32), ("m", ctypes.c_uint64), ("omega", ctypes.c_uint64 * 4),
                ("shift", ctypes.c_uint64 * 4)]

    @staticmethod
    def make(kind: int, m: int, omega, shift=None) -> "Domain":
        d = Domain()
        d.kind, d.reserved, d.m = int(kind), 0, int(m)
        for i, v in enumerate(np.asarray(omega, dtype=np.uint64).reshape(4)):
            d.omega[i] = int(v)
        if shift is not None:
            for i, v in enumerate(np.asarray(shift, dtype=np.uint64).reshape(4)):
                d.shift[i] = int(v)
        return d


class GateProgram(ctypes.Structure):
    """zkhip_gate_program"""
    _fields_ = [("n_gates", ctypes.c_uint32), ("n_terms", ctypes.c_uint32), ("n_factors", ctypes.c_uint32), ("n_slots", ctypes.c_uint32),
                ("gate_terms", ctypes.c_void_p), ("gate_selector", ctypes.c_void_p), ("gate_selector_rot", ctypes.c_void_p), ("term_factors", ctypes.c_void_p),
                ("factor_slot", ctypes.c_void_p), ("factor_rot", ctypes.c_void_p), ("term_coeff", ctypes.c_void_p)]


def domain_choice(curve: int, min_size: int):
    """(kind, m) of make_evaluation_domain(min_size) over the curve's scalar field (zkhip_domain_choice; host arithmetic only)"""
    kind, m = ctypes.c_int(), ctypes.c_size_t()
    rc = load_library().zkhip_domain_choice(curve, ctypes.c_size_t(min_size), ctypes.byref(kind), ctypes.byref(m))
    if rc != 0:
        raise ZkhipError(f"zkhip_domain_choice({min_size}): {rc}")
    return kind.value, m.value


def coord_limbs(curve: int, group: int) -> int:
    """u64 limbs per coordinate (Fq for G1, Fq2 for G2)."""
    return _FQ[curve] * (2 if group == G2 else 1)


def lib_path() -> str:
    return os.path.join(_HERE, "libzkhip.so")


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _HERE, "-j4", "all"]
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"])
    subprocess.check_call(args)
    return lib_path()


_LIB = None


def load_library() -> ctypes.CDLL:
    """Load libzkhip.so; raises if it has not been built (there is no fallback implementation)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ZkhipError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(zkhip has no CPU fallback)")
    lib = ctypes.CDLL(path)
    lib.zkhip_strerror.restype = ctypes.c_char_p
    lib.zkhip_last_error.restype = ctypes.c_char_p
    lib.zkhip_last_error.argtypes = [ctypes.c_void_p]
    lib.zkhip_bases_size.restype = ctypes.c_size_t
    lib.zkhip_bases_size.argtypes = [ctypes.c_void_p]
    lib.zkhip_profile_dump.restype = ctypes.c_size_t
    lib.zkhip_r1cs_domain_size.restype = ctypes.c_size_t
    lib.zkhip_r1cs_domain_size.argtypes = [ctypes.c_void_p]
    lib.zkhip_groth16_scratch_bytes.restype = ctypes.c_size_t
    lib.zkhip_groth16_scratch_bytes.argtypes = [ctypes.c_void_p]
    lib.zkhip_r1cs_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.zkhip_r1cs_set_domain.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    lib.zkhip_r1cs_domain_kind.argtypes = [ctypes.c_void_p]
    lib.zkhip_destroy.argtypes = [ctypes.c_void_p]
    lib.zkhip_bases_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.zkhip_group_destroy.argtypes = [ctypes.c_void_p]
    lib.zkhip_group_size.argtypes = [ctypes.c_void_p]
    lib.zkhip_group_ctx.restype = ctypes.c_void_p
    lib.zkhip_group_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zkhip_group_last_error.restype = ctypes.c_char_p
    lib.zkhip_group_last_error.argtypes = [ctypes.c_void_p]
    lib.zkhip_group_transport.argtypes = [ctypes.c_void_p]
    lib.zkhip_group_bases_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.zkhip_group_bases_size.restype = ctypes.c_size_t
    lib.zkhip_group_bases_size.argtypes = [ctypes.c_void_p]
    lib.zkhip_group_bases_member.restype = ctypes.c_void_p
    lib.zkhip_group_bases_member.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    _LIB = lib
    return lib


def _p(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


_BLOCK_HOLDER = {}  # device pointer of a live Context.malloc block -> the Context that tracks it


class Context:
    """One context per GPU per process (zkhip_init / zkhip_destroy)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.zkhip_init(int(device), ctypes.byref(h))
        if rc != 0:
            raise ZkhipError(f"zkhip_init: {self.lib.zkhip_strerror(rc).decode()}")
        self.h = h
        self.device = device
        self._live = set()  # blocks of malloc() not freed yet: close() hands them back (zkhip_destroy leaves live blocks alone, zkhip.h)
        self._owns = True

    def _check(self, rc, what):
        if rc != 0:
            raise ZkhipError(f"{what}: {self.lib.zkhip_strerror(rc).decode()} [{self.lib.zkhip_last_error(self.h).decode()}]")

    def close(self, release_blocks: bool = True):
        """zkhip_destroy; blocks of malloc() that were never freed go back first (zkhip_destroy itself leaves live blocks alone -- zkhip.h,
        "OWNERSHIP" -- so a script that forgets them would leak until the process ends: ADVICE r5).  release_blocks=False keeps them
        (they stay valid and can be freed through any other context)."""
        if getattr(self, "h", None):
            for p in list(getattr(self, "_live", ())):
                _BLOCK_HOLDER.pop(p, None)
                if release_blocks:
                    self.lib.zkhip_free(self.h, ctypes.c_void_p(p))
            self._live = set()
            if getattr(self, "_owns", True):
                self.lib.zkhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int):
        self._check(self.lib.zkhip_set_stream(self.h, ctypes.c_void_p(stream_handle)), "zkhip_set_stream")

    def sync(self):
        self._check(self.lib.zkhip_sync(self.h), "zkhip_sync")

    def device_status(self) -> int:
        """sticky kernel-raised error flags since the last call (raises ZkhipError if any was set)"""
        f = ctypes.c_uint32()
        self._check(self.lib.zkhip_device_status(self.h, ctypes.byref(f)), "zkhip_device_status")
        return f.value

    def fr_gather_dev(self, d_src: int, src_count: int, d_indices: int, count: int, d_dst: int):
        self._check(self.lib.zkhip_fr_gather_dev(self.h, ctypes.c_void_p(d_src), ctypes.c_size_t(src_count), ctypes.c_void_p(d_indices),
                                                 ctypes.c_size_t(count), ctypes.c_void_p(d_dst)), "zkhip_fr_gather_dev")

    def set_option(self, name: str, value: int):
        self._check(self.lib.zkhip_set_option(self.h, name.encode(), ctypes.c_int64(value)), "zkhip_set_option")

    def get_option(self, name: str) -> int:
        v = ctypes.c_int64(0)
        self._check(self.lib.zkhip_get_option(self.h, name.encode(), ctypes.byref(v)), "zkhip_get_option")
        return int(v.value)

    # ---- device memory
    def malloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._check(self.lib.zkhip_malloc(self.h, ctypes.c_size_t(nbytes), ctypes.byref(p)), "zkhip_malloc")
        self._live.add(p.value)
        _BLOCK_HOLDER[p.value] = self
        return p.value

    def free(self, dptr: int):
        holder = _BLOCK_HOLDER.pop(dptr, None)  # a block may be freed through another Context than the one that allocated it
        if holder is not None:
            holder._live.discard(dptr)
        self._check(self.lib.zkhip_free(self.h, ctypes.c_void_p(dptr)), "zkhip_free")

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.zkhip_memcpy_h2d(self.h, ctypes.c_void_p(dptr), _p(arr), ctypes.c_size_t(arr.nbytes)), "zkhip_memcpy_h2d")

    def d2h(self, arr: np.ndarray, dptr: int):
        self._check(self.lib.zkhip_memcpy_d2h(self.h, _p(arr), ctypes.c_void_p(dptr), ctypes.c_size_t(arr.nbytes)), "zkhip_memcpy_d2h")

    def copy_2d(self, dst: int, dst_pitch: int, src: int, src_pitch: int, width: int, rows: int):
        """rows of `width` bytes at a pitch, device to device on this context's GPU, in stream order (zkhip_memcpy_2d_d2d_async)"""
        self._check(self.lib.zkhip_memcpy_2d_d2d_async(self.h, ctypes.c_void_p(dst), ctypes.c_size_t(dst_pitch), ctypes.c_void_p(src), ctypes.c_size_t(src_pitch),
                                                       ctypes.c_size_t(width), ctypes.c_size_t(rows)), "zkhip_memcpy_2d_d2d_async")

    # ---- bases
    def upload_bases(self, curve: int, group: int, affine: np.ndarray, inf=None) -> "Bases":
        affine = _u64(affine)
        n = affine.shape[0] if affine.ndim == 2 else 0
        infa = np.ascontiguousarray(inf, dtype=np.uint8) if inf is not None else None
        h = ctypes.c_void_p()
        self._check(self.lib.zkhip_bases_upload(self.h, curve, group, _p(affine), _p(infa), ctypes.c_size_t(n), ctypes.byref(h)),
                    "zkhip_bases_upload")
        return Bases(self, h, curve, group, n)

    def upload_bases_compressed(self, curve: int, group: int, octets: bytes, n: int) -> "Bases":
        """bases from n compressed wire encodings (48 B per G1 point, 96 B per G2 point), decoded on the device"""
        buf = np.frombuffer(octets, dtype=np.uint8)
        assert len(buf) == n * 48 * group
        h = ctypes.c_void_p()
        self._check(self.lib.zkhip_bases_upload_compressed(self.h, curve, group, _p(np.ascontiguousarray(buf)), ctypes.c_size_t(n), ctypes.byref(h)),
                    "bases_upload_compressed")
        return Bases(self, h, curve, group, n)

    def bases_from_scalars(self, curve: int, group: int, scalars: np.ndarray, base=None) -> "Bases":
        scalars = _u64(scalars)
        n = scalars.shape[0]
        h = ctypes.c_void_p()
        b = _u64(base) if base is not None else None
        self._check(self.lib.zkhip_bases_from_scalars(self.h, curve, group, _p(b), _p(scalars), ctypes.c_size_t(n), ctypes.byref(h)),
                    "zkhip_bases_from_scalars")
        return Bases(self, h, curve, group, n)

    # ---- MSM
    def msm(self, bases: "Bases", scalars: np.ndarray, offset: int = 0, n=None) -> np.ndarray:
        """-> Jacobian (3, coord_limbs) canonical u64."""
        scalars = _u64(scalars)
        n = scalars.shape[0] if n is None else n
        out = np.zeros((3, coord_limbs(bases.curve, bases.group)), dtype=np.uint64)
        self._check(self.lib.zkhip_msm(self.h, bases.h, ctypes.c_size_t(offset), ctypes.c_size_t(n), _p(scalars), _p(out)), "zkhip_msm")
        return out

    def msm_dev(self, bases: "Bases", d_scalars: int, d_out: int, offset: int = 0, n=None):
        n = bases.n - offset if n is None else n
        self._check(self.lib.zkhip_msm_dev(self.h, bases.h, ctypes.c_size_t(offset), ctypes.c_size_t(n), ctypes.c_void_p(d_scalars),
                                           ctypes.c_void_p(d_out)), "zkhip_msm_dev")

    def msm_batch_dev(self, bases_list, d_scalars_list, d_out_list, offsets=None, ns=None):
        cnt = len(bases_list)
        offsets = offsets or [0] * cnt
        ns = ns or [b.n - o for b, o in zip(bases_list, offsets)]
        B = (ctypes.c_void_p * cnt)(*[b.h for b in bases_list])
        O = (ctypes.c_size_t * cnt)(*offsets)
        N = (ctypes.c_size_t * cnt)(*ns)
        S = (ctypes.c_void_p * cnt)(*d_scalars_list)
        D = (ctypes.c_void_p * cnt)(*d_out_list)
        self._check(self.lib.zkhip_msm_batch_dev(self.h, ctypes.c_size_t(cnt), B, O, N, S, D), "zkhip_msm_batch_dev")

    def jacobian_sum_dev(self, curve: int, group: int, d_points: int, count: int, d_out: int):
        self._check(self.lib.zkhip_jacobian_sum_dev(self.h, curve, group, ctypes.c_void_p(d_points), ctypes.c_size_t(count),
                                                    ctypes.c_void_p(d_out)), "zkhip_jacobian_sum_dev")

    def jacobian_to_affine(self, curve: int, group: int, jac: np.ndarray):
        jac = _u64(jac)
        out = np.zeros((2, coord_limbs(curve, group)), dtype=np.uint64)
        inf = np.zeros(1, dtype=np.uint8)
        self._check(self.lib.zkhip_jacobian_to_affine(self.h, curve, group, _p(jac), _p(out), _p(inf)), "zkhip_jacobian_to_affine")
        return out.reshape(-1), int(inf[0])

    def msm_affine(self, bases: "Bases", scalars: np.ndarray, offset: int = 0, n=None):
        """MSM followed by the on-device Jacobian -> affine conversion: (flat affine limbs, is_infinity)."""
        return self.jacobian_to_affine(bases.curve, bases.group, self.msm(bases, scalars, offset, n))

    # ---- NTT
    def ntt(self, curve: int, data: np.ndarray, log_m: int, omega, inverse=False, coset=None) -> np.ndarray:
        """data (batch, m, 4) canonical u64 -> transformed copy."""
        d = _u64(data).copy()
        batch = d.shape[0] if d.ndim == 3 else 1
        self._check(self.lib.zkhip_ntt(self.h, curve, _p(d), ctypes.c_size_t(log_m), ctypes.c_size_t(batch), _p(_u64(omega)),
                                       1 if inverse else 0, _p(_u64(coset)) if coset is not None else None), "zkhip_ntt")
        return d

    def ntt_dev(self, curve: int, d_data: int, log_m: int, batch: int, omega, inverse=False, coset=None):
        self._check(self.lib.zkhip_ntt_dev(self.h, curve, ctypes.c_void_p(d_data), ctypes.c_size_t(log_m), ctypes.c_size_t(batch),
                                           _p(_u64(omega)), 1 if inverse else 0, _p(_u64(coset)) if coset is not None else None),
                    "zkhip_ntt_dev")

    # ---- the gate argument's sum over a flat program (zkhip_gate_eval_dev)
    def gate_eval_dev(self, curve: int, gates, d_slots, log_size: int, d_out: int, d_mask: int = 0, accumulate: bool = False):
        """gates: [(selector or None, [(coeff int, [(slot, rot), ...]), ...]), ...] with selector = (slot, rot); rotations in rows of the
        2^log_size-point domain; d_slots: device pointers"""
        gate_terms, gate_sel, gate_rot, term_factors, fslot, frot, coeffs = [0], [], [], [0], [], [], []
        for sel, terms in gates:
            gate_sel.append(0xFFFFFFFF if sel is None else sel[0])
            gate_rot.append(0 if sel is None else sel[1])
            for c, factors in terms:
                for sl, rt in factors:
                    fslot.append(sl)
                    frot.append(rt)
                term_factors.append(len(fslot))
                coeffs.append([(c >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])
            gate_terms.append(len(term_factors) - 1)
        keep = [np.array(gate_terms, dtype=np.uint32), np.array(gate_sel, dtype=np.uint32), np.array(gate_rot, dtype=np.int32),
                np.array(term_factors, dtype=np.uint32), np.array(fslot or [0], dtype=np.uint32), np.array(frot or [0], dtype=np.int32),
                np.array(coeffs or [[0, 0, 0, 0]], dtype=np.uint64)]
        prog = GateProgram(len(gate_sel), len(coeffs), len(fslot), len(d_slots), *[a.ctypes.data_as(ctypes.c_void_p) for a in keep])
        S = (ctypes.c_void_p * max(1, len(d_slots)))(*d_slots)
        self._check(self.lib.zkhip_gate_eval_dev(self.h, curve, ctypes.byref(prog), S, ctypes.c_size_t(log_size), ctypes.c_void_p(d_mask) if d_mask else None,
                                                 1 if accumulate else 0, ctypes.c_void_p(d_out)), "zkhip_gate_eval_dev")

    # ---- Groth16 witness map
    def upload_r1cs(self, curve: int, M: int, n: int, N: int, csr_a, csr_b, csr_c) -> "R1CS":
        """csr_x = (rowptr u32 (M+1), col u32 (nnz), coeff u64 (nnz, 4))"""
        args = []
        keep = []
        for rp, cl, cf in (csr_a, csr_b, csr_c):
            rp = np.ascontiguousarray(rp, dtype=np.uint32)
            cl = np.ascontiguousarray(cl, dtype=np.uint32)
            cf = _u64(cf)
            keep += [rp, cl, cf]
            args += [_p(rp), _p(cl), _p(cf)]
        h = ctypes.c_void_p()
        self._check(self.lib.zkhip_r1cs_upload(self.h, curve, ctypes.c_size_t(M), ctypes.c_size_t(n), ctypes.c_size_t(N), *args, ctypes.byref(h)),
                    "zkhip_r1cs_upload")
        return R1CS(self, h, curve, M, n, N)

    def groth16_witness_h_dev(self, r1cs: "R1CS", d_assignment: int, omega, coset, d_h: int, d_scratch: int):
        self._check(self.lib.zkhip_groth16_witness_h_dev(self.h, r1cs.h, ctypes.c_void_p(d_assignment), _p(_u64(omega)), _p(_u64(coset)),
                                                         ctypes.c_void_p(d_h), ctypes.c_void_p(d_scratch)), "zkhip_groth16_witness_h_dev")

    def groth16_witness_h_domain_dev(self, r1cs: "R1CS", d_assignment: int, dom: Domain, coset, d_h: int, d_scratch: int):
        self._check(self.lib.zkhip_groth16_witness_h_domain_dev(self.h, r1cs.h, ctypes.c_void_p(d_assignment), ctypes.byref(dom), _p(_u64(coset)),
                                                                ctypes.c_void_p(d_h), ctypes.c_void_p(d_scratch)), "zkhip_groth16_witness_h_domain_dev")

    def domain_fft_dev(self, curve: int, dom: Domain, d_data: int, batch: int, inverse=False, coset=None):
        self._check(self.lib.zkhip_domain_fft_dev(self.h, curve, ctypes.byref(dom), ctypes.c_void_p(d_data), ctypes.c_size_t(batch), 1 if inverse else 0,
                                                  _p(_u64(coset)) if coset is not None else None), "zkhip_domain_fft_dev")

    def domain_lagrange(self, curve: int, dom: Domain, t) -> np.ndarray:
        """evaluate_all_lagrange_polynomials(t): (m, 4) canonical, in get_domain_element order"""
        out = np.zeros((int(dom.m), 4), dtype=np.uint64)
        dp = self.malloc(out.nbytes)
        try:
            self._check(self.lib.zkhip_domain_lagrange_dev(self.h, curve, ctypes.byref(dom), _p(_u64(t).reshape(4)), ctypes.c_void_p(dp)), "zkhip_domain_lagrange_dev")
            self.d2h(out, dp)
        finally:
            self.free(dp)
        return out

    def domain_fft(self, curve: int, dom: Domain, data: np.ndarray, inverse=False, coset=None) -> np.ndarray:
        """host convenience: (batch, m, 4) canonical -> transformed copy"""
        d = _u64(data).copy()
        batch = d.shape[0] if d.ndim == 3 else 1
        dp = self.malloc(d.nbytes)
        try:
            self.h2d(dp, d)
            self.domain_fft_dev(curve, dom, dp, batch, inverse, coset)
            self.d2h(d, dp)
        finally:
            self.free(dp)
        return d

    def groth16_witness_h(self, r1cs: "R1CS", assignment_with_one: np.ndarray, omega, coset) -> np.ndarray:
        """host convenience: (N+1, 4) canonical assignment (1 | primary | auxiliary) -> (m+1, 4) coefficients of H.  `omega`: the
        domain's root (4 limbs) or a Domain"""
        z = _u64(assignment_with_one)
        m = r1cs.m
        d_z = self.malloc(z.nbytes)
        d_h = self.malloc((m + 1) * 32)
        d_s = self.malloc(self.lib.zkhip_groth16_scratch_bytes(r1cs.h))
        try:
            self.h2d(d_z, z)
            if isinstance(omega, Domain):
                self.groth16_witness_h_domain_dev(r1cs, d_z, omega, coset, d_h, d_s)
            else:
                self.groth16_witness_h_dev(r1cs, d_z, omega, coset, d_h, d_s)
            out = np.zeros((m + 1, 4), dtype=np.uint64)
            self.d2h(out, d_h)
        finally:
            self.free(d_z)
            self.free(d_h)
            self.free(d_s)
        return out

    # ---- profiling
    def ec_ntt_dev(self, curve: int, group: int, d_jacobian: int, log_m: int, omega, inverse=False):
        """DFT over group elements, in place on 2^log_m canonical Jacobian points (powers-of-tau Lagrange basis)"""
        w = _u64(omega).reshape(4)
        self._check(self.lib.zkhip_ec_ntt_dev(self.h, curve, group, ctypes.c_void_p(d_jacobian), ctypes.c_size_t(log_m), _p(w), 1 if inverse else 0),
                    "ec_ntt_dev")

    # ---- coefficient-form polynomial arithmetic (KZG opening proofs) ----
    def fr_vec_op_dev(self, curve: int, op: int, d_a: int, d_b: int, d_out: int, count: int):
        self._check(self.lib.zkhip_fr_vec_op_dev(self.h, curve, op, ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_out),
                                                 ctypes.c_size_t(count)), "fr_vec_op_dev")

    def fr_vec_prod_dev(self, curve: int, d_in, d_out: int, n: int):
        ptrs = (ctypes.c_void_p * len(d_in))(*d_in)
        self._check(self.lib.zkhip_fr_vec_prod_dev(self.h, curve, ctypes.c_size_t(len(d_in)), ptrs, ctypes.c_void_p(d_out), ctypes.c_size_t(n)), "fr_vec_prod_dev")

    def poly_shift_dev(self, d_in: int, log_size: int, rotation: int, d_out: int):
        self._check(self.lib.zkhip_poly_shift_dev(self.h, ctypes.c_void_p(d_in), ctypes.c_size_t(log_size), ctypes.c_int64(rotation), ctypes.c_void_p(d_out)),
                    "poly_shift_dev")

    def poly_resize_dev(self, curve: int, d_in: int, log_n: int, batch: int, omega_n, d_out: int, log_out: int, omega_out):
        self._check(self.lib.zkhip_poly_resize_dev(self.h, curve, ctypes.c_void_p(d_in), ctypes.c_size_t(log_n), ctypes.c_size_t(batch), _p(_u64(omega_n)),
                                                   ctypes.c_void_p(d_out), ctypes.c_size_t(log_out), _p(_u64(omega_out))), "poly_resize_dev")

    def poly_eval_dev(self, curve: int, d_polys: int, n: int, batch: int, points: np.ndarray, stride=None) -> np.ndarray:
        """out[b, p] = poly_b(points[p]) for `batch` coefficient vectors of n elements, `stride` elements apart"""
        pts = _u64(points).reshape(-1, 4)
        out = np.zeros((batch, len(pts), 4), dtype=np.uint64)
        self._check(self.lib.zkhip_poly_eval_dev(self.h, curve, ctypes.c_void_p(d_polys), ctypes.c_size_t(n), ctypes.c_size_t(n if stride is None else stride),
                                                 ctypes.c_size_t(batch), _p(pts), ctypes.c_size_t(len(pts)), _p(out)), "poly_eval_dev")
        return out

    def poly_div_linear_dev(self, curve: int, d_f: int, n: int, z, d_out: int) -> np.ndarray:
        """d_out[0] = f(z), d_out[1:] = f / (X - z); returns f(z)"""
        zz = _u64(z).reshape(4)
        rem = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.zkhip_poly_div_linear_dev(self.h, curve, ctypes.c_void_p(d_f), ctypes.c_size_t(n), _p(zz), ctypes.c_void_p(d_out), _p(rem)),
                    "poly_div_linear_dev")
        return rem

    def poly_div_vanishing_dev(self, curve: int, d_f: int, length: int, n: int, d_quot: int) -> int:
        """d_quot[0 : length - n) = f / (X^n - 1) in coefficient form; returns the number of non-zero remainder coefficients"""
        bad = ctypes.c_uint64(0)
        self._check(self.lib.zkhip_poly_div_vanishing_dev(self.h, curve, ctypes.c_void_p(d_f), ctypes.c_size_t(length), ctypes.c_size_t(n), ctypes.c_void_p(d_quot),
                                                          ctypes.byref(bad)), "poly_div_vanishing_dev")
        return int(bad.value)

    def perm_grand_product_dev(self, curve: int, d_cols, d_sid, d_ssigma, n: int, beta, gamma, d_g: int, d_h: int, d_vp: int):
        """placeholder's permutation grand product: g / h vectors (k x n at d_g / d_h, 0 to skip) and V_P (n at d_vp)"""
        k = len(d_cols)
        arr = lambda ps: (ctypes.c_void_p * k)(*ps)
        self._check(self.lib.zkhip_perm_grand_product_dev(self.h, curve, ctypes.c_size_t(k), arr(d_cols), arr(d_sid), arr(d_ssigma), ctypes.c_size_t(n),
                                                          _p(_u64(beta).reshape(4)), _p(_u64(gamma).reshape(4)), ctypes.c_void_p(d_g or None),
                                                          ctypes.c_void_p(d_h or None), ctypes.c_void_p(d_vp)), "perm_grand_product_dev")

    def fr_vec_affine_dev(self, curve: int, d_x: int, d_y: int, a, b, c, d_out: int, count: int):
        """d_out[i] = a d_x[i] + b d_y[i] + c (d_y = 0: no second operand)"""
        self._check(self.lib.zkhip_fr_vec_affine_dev(self.h, curve, ctypes.c_void_p(d_x), ctypes.c_void_p(d_y or None), _p(_u64(a).reshape(4)),
                                                     _p(_u64(b).reshape(4)) if d_y else None, _p(_u64(c).reshape(4)), ctypes.c_void_p(d_out), ctypes.c_size_t(count)),
                    "fr_vec_affine_dev")

    def fr_vec_mul_div_dev(self, curve: int, d_a: int, d_b: int, d_c: int, d_out: int, count: int):
        """d_out[j] = d_a[j] d_b[j] / d_c[j], j < count"""
        self._check(self.lib.zkhip_fr_vec_mul_div_dev(self.h, curve, ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_c), ctypes.c_void_p(d_out),
                                                      ctypes.c_size_t(count)), "fr_vec_mul_div_dev")

    def lookup_grand_product_dev(self, curve: int, d_input, d_value, d_sorted, n: int, usable_rows: int, beta, gamma, d_vl: int):
        """placeholder's lookup grand product V_L (n at d_vl) from the reduced input / value / sorted vectors"""
        arr = lambda ps: (ctypes.c_void_p * max(len(ps), 1))(*ps)
        self._check(self.lib.zkhip_lookup_grand_product_dev(self.h, curve, ctypes.c_size_t(len(d_input)), arr(d_input), ctypes.c_size_t(len(d_value)), arr(d_value),
                                                            ctypes.c_size_t(len(d_sorted)), arr(d_sorted), ctypes.c_size_t(n), ctypes.c_size_t(usable_rows),
                                                            _p(_u64(beta).reshape(4)), _p(_u64(gamma).reshape(4)), ctypes.c_void_p(d_vl)), "lookup_grand_product_dev")

    def lookup_sort_dev(self, d_input, d_value, n: int, usable_rows: int, d_sorted):
        """placeholder's sort_polynomials (lookup_argument.hpp:565-638): len(d_input) + len(d_value) sorted vectors of n at d_sorted"""
        arr = lambda ps: (ctypes.c_void_p * max(len(ps), 1))(*ps)
        self._check(self.lib.zkhip_lookup_sort_dev(self.h, ctypes.c_size_t(len(d_input)), arr(d_input), ctypes.c_size_t(len(d_value)), arr(d_value),
                                                   ctypes.c_size_t(n), ctypes.c_size_t(usable_rows), arr(d_sorted)), "lookup_sort_dev")

    def poly_lincomb_dev(self, curve: int, d_polys, lens, coeffs: np.ndarray, taps: int, d_acc: int, acc_len: int, accumulate: bool):
        count = len(d_polys)
        ptrs = (ctypes.c_void_p * max(1, count))(*d_polys)
        ls = (ctypes.c_size_t * max(1, count))(*lens)
        cf = _u64(coeffs).reshape(-1, 4)
        assert len(cf) == count * taps
        self._check(self.lib.zkhip_poly_lincomb_dev(self.h, curve, ctypes.c_size_t(count), ptrs, ls, _p(cf), ctypes.c_size_t(taps), ctypes.c_void_p(d_acc),
                                                    ctypes.c_size_t(acc_len), 1 if accumulate else 0), "poly_lincomb_dev")

    def profile(self, on: bool):
        self._check(self.lib.zkhip_profile_enable(self.h, 1 if on else 0), "zkhip_profile_enable")

    def profile_filter(self, prefix: str = ""):
        self._check(self.lib.zkhip_profile_filter(self.h, prefix.encode()), "zkhip_profile_filter")

    def profile_reset(self):
        self._check(self.lib.zkhip_profile_reset(self.h), "zkhip_profile_reset")

    def profile_get(self, prefix: str):
        ms = ctypes.c_double()
        cnt = ctypes.c_uint64()
        self._check(self.lib.zkhip_profile_get(self.h, prefix.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "zkhip_profile_get")
        return ms.value, cnt.value

    def profile_dump(self) -> dict:
        need = self.lib.zkhip_profile_dump(self.h, None, ctypes.c_size_t(0))
        buf = ctypes.create_string_buffer(need + 16)
        self.lib.zkhip_profile_dump(self.h, buf, ctypes.c_size_t(need + 16))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt = line.rsplit(" ", 2)
            out[name] = (float(ms), int(cnt))
        return out


class R1CS:
    """Resident constraint system (zkhip_r1cs)."""

    def __init__(self, ctx: "Context", h, curve, M, n, N):
        self.ctx, self.h, self.curve, self.M, self.n, self.N = ctx, h, curve, M, n, N
        self._dims()

    def _dims(self):
        self.m = self.ctx.lib.zkhip_r1cs_domain_size(self.h)
        self.kind = self.ctx.lib.zkhip_r1cs_domain_kind(self.h)
        self.log_m = (self.m - 1).bit_length()    # log2 of the power of two that holds the domain (basic: of m itself)

    def set_domain(self, kind: int, m: int):
        """another evaluation domain than make_evaluation_domain's choice (e.g. the basic one of the next power of two)"""
        self.ctx._check(self.ctx.lib.zkhip_r1cs_set_domain(self.h, int(kind), ctypes.c_size_t(m)), "zkhip_r1cs_set_domain")
        self._dims()

    def free(self):
        if self.h is not None and self.ctx.h:
            self.ctx.lib.zkhip_r1cs_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


GROUP_AUTO, GROUP_RCCL, GROUP_PEER, GROUP_STAGED = 0, 1, 2, 3


class DeviceGroup:
    """zkhip_group_*: one context per entry of `devices` behind one caller (a device may repeat: several members on one GPU)."""

    def __init__(self, devices):
        self.lib = load_library()
        devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self.lib.zkhip_group_init(arr, len(devices), ctypes.byref(h))
        if rc != 0:
            raise ZkhipError(f"zkhip_group_init({devices}): {self.lib.zkhip_strerror(rc).decode()}")
        self.h = h
        self.devices = devices
        self.members = []
        for k in range(len(devices)):  # the members as ordinary Context objects that do not own their handle
            c = Context.__new__(Context)
            c.lib, c.h, c.device, c._live, c._owns = self.lib, ctypes.c_void_p(self.lib.zkhip_group_ctx(self.h, k)), devices[k], set(), False
            self.members.append(c)

    def __len__(self):
        return len(self.devices)

    def _check(self, rc, what):
        if rc != 0:
            raise ZkhipError(f"{what}: {self.lib.zkhip_strerror(rc).decode()} [{self.lib.zkhip_group_last_error(self.h).decode()}]")

    def close(self):
        if getattr(self, "h", None):
            for c in self.members:
                c.close()  # frees what was allocated through the member; the handle is the group's
            self.lib.zkhip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_transport(self, kind: int):
        self._check(self.lib.zkhip_group_set_transport(self.h, int(kind)), "zkhip_group_set_transport")

    def transport(self) -> int:
        return self.lib.zkhip_group_transport(self.h)

    def sync(self):
        self._check(self.lib.zkhip_group_sync(self.h), "zkhip_group_sync")

    def all_gather(self, d_send, d_recv, nbytes: int):
        """d_send / d_recv: one device pointer per member (d_recv entries may be None)"""
        n = len(self)
        S = (ctypes.c_void_p * n)(*d_send)
        R = (ctypes.c_void_p * n)(*[r if r else None for r in d_recv])
        self._check(self.lib.zkhip_group_all_gather(self.h, S, R, ctypes.c_size_t(nbytes)), "zkhip_group_all_gather")

    def copy(self, dst_member: int, d_dst: int, src_member: int, d_src: int, nbytes: int):
        self._check(self.lib.zkhip_group_copy(self.h, dst_member, ctypes.c_void_p(d_dst), src_member, ctypes.c_void_p(d_src), ctypes.c_size_t(nbytes)),
                    "zkhip_group_copy")

    def upload_bases(self, curve: int, group: int, affine: np.ndarray, inf=None) -> "GroupBases":
        affine = _u64(affine)
        n = affine.shape[0] if affine.ndim == 2 else 0
        infa = np.ascontiguousarray(inf, dtype=np.uint8) if inf is not None else None
        h = ctypes.c_void_p()
        self._check(self.lib.zkhip_group_bases_upload(self.h, curve, group, _p(affine), _p(infa), ctypes.c_size_t(n), ctypes.byref(h)), "zkhip_group_bases_upload")
        return GroupBases(self, h, curve, group, n)

    def bases_from_scalars(self, curve: int, group: int, scalars: np.ndarray, base=None) -> "GroupBases":
        scalars = _u64(scalars)
        h = ctypes.c_void_p()
        b = _u64(base) if base is not None else None
        self._check(self.lib.zkhip_group_bases_from_scalars(self.h, curve, group, _p(b), _p(scalars), ctypes.c_size_t(scalars.shape[0]), ctypes.byref(h)),
                    "zkhip_group_bases_from_scalars")
        return GroupBases(self, h, curve, group, scalars.shape[0])

    def msm(self, bases: "GroupBases", scalars: np.ndarray, offset: int = 0, n=None) -> np.ndarray:
        scalars = _u64(scalars)
        n = scalars.shape[0] if n is None else n
        out = np.zeros((3, coord_limbs(bases.curve, bases.group)), dtype=np.uint64)
        self._check(self.lib.zkhip_group_msm(self.h, bases.h, ctypes.c_size_t(offset), ctypes.c_size_t(n), _p(scalars), _p(out)), "zkhip_group_msm")
        return out

    def msm_affine(self, bases: "GroupBases", scalars: np.ndarray, offset: int = 0, n=None):
        return self.members[0].jacobian_to_affine(bases.curve, bases.group, self.msm(bases, scalars, offset, n))

    def ntt(self, curve: int, data: np.ndarray, log_m: int, omega, inverse=False, coset=None) -> np.ndarray:
        d = _u64(data).copy()
        batch = d.shape[0] if d.ndim == 3 else 1
        self._check(self.lib.zkhip_group_ntt(self.h, curve, _p(d), ctypes.c_size_t(log_m), ctypes.c_size_t(batch), _p(_u64(omega)), 1 if inverse else 0,
                                             _p(_u64(coset)) if coset is not None else None), "zkhip_group_ntt")
        return d


class GroupBases:
    """zkhip_group_bases: resident bases cut by point range over a group's members."""

    def __init__(self, group: DeviceGroup, h, curve, grp, n):
        self.g, self.h, self.curve, self.group, self.n = group, h, curve, grp, n

    def member_first(self, k: int) -> int:
        first = ctypes.c_size_t()
        self.g.lib.zkhip_group_bases_member(self.h, k, ctypes.byref(first))
        return first.value

    def free(self):
        if self.h is not None and self.g.h:
            self.g.lib.zkhip_group_bases_free(self.g.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Bases:
    """Resident proving-key query / SRS (zkhip_bases)."""

    def __init__(self, ctx: Context, h, curve, group, n):
        self.ctx, self.h, self.curve, self.group, self.n = ctx, h, curve, group, n

    def download(self, offset=0, n=None):
        n = self.n - offset if n is None else n
        out = np.zeros((n, 2 * coord_limbs(self.curve, self.group)), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        self.ctx._check(self.ctx.lib.zkhip_bases_download(self.ctx.h, self.h, ctypes.c_size_t(offset), ctypes.c_size_t(n), _p(out), _p(inf)),
                        "zkhip_bases_download")
        return out, inf

    def spread(self, n_total, d_rows=None, first=0):
        """a bases object of n_total points with these at the rows d_rows[j] (device u32 array; None: first + j), infinity elsewhere"""
        out = ctypes.c_void_p()
        self.ctx._check(self.ctx.lib.zkhip_bases_spread(self.ctx.h, self.h, ctypes.c_void_p(d_rows) if d_rows else None, ctypes.c_size_t(first),
                                                        ctypes.c_size_t(n_total), ctypes.byref(out)), "zkhip_bases_spread")
        return Bases(self.ctx, out, self.curve, self.group, n_total)

    def free(self):
        if self.h is not None and self.ctx.h:
            self.ctx.lib.zkhip_bases_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
