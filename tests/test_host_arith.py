"""The __host__ __device__ field / curve / recoding code of crypto3-zk_amd/csrc (fp.hpp, curve.hpp,
msm_recode.hpp), compiled for the CPU into libzkhip_hosttest.so, against the big-integer oracle.
This checks the exact arithmetic the GPU kernels run, without a GPU.  (The shim is test-only.)"""
import ctypes
import os
import random

import numpy as np
import pytest

import pyoracle as po
from util import CURVES, FQ_LIMBS, pt_from_limbs, pts_arr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "crypto3-zk_amd", "libzkhip_hosttest.so")


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SO):
        pytest.fail(f"{SO} missing: run __graft_entry__.build()")
    return ctypes.CDLL(SO)


def _u32(v, n32):
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(n32)], dtype=np.uint32)


def _int(a):
    return sum(int(x) << (32 * i) for i, x in enumerate(a))


# saturated reference types 0-3, lazy 29-bit-limb compute types 6-9 (same boundary form: canonical u32 limbs)
FIELDS = {0: (po.BLS12_381.p, 12), 1: (po.BLS12_381.r, 8), 2: (po.BN254.p, 8), 3: (po.BN254.r, 8),
          6: (po.BLS12_381.p, 12), 7: (po.BN254.p, 8), 8: (po.BLS12_381.r, 8), 9: (po.BN254.r, 8)}


@pytest.mark.parametrize("field", [0, 1, 2, 3, 6, 7, 8, 9])
def test_prime_field_ops(shim, field):
    p, nl = FIELDS[field]
    random.seed(field)
    edge = [0, 1, 2, p - 1, p - 2, (1 << (32 * nl - 1)) % p, (p + 1) // 2]
    vals = edge + [random.randrange(p) for _ in range(60)]
    out = np.zeros(nl, dtype=np.uint32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        A, B = _u32(a, nl), _u32(b, nl)
        X = (a * a - a * b - 2 * b * b) % p
        for op, fn in ((0, lambda: a * b % p), (1, lambda: (a + b) % p), (2, lambda: (a - b) % p), (4, lambda: a * a % p),
                       (5, lambda: (-a) % p), (6, lambda: 2 * a % p), (7, lambda: (a - b) % p),
                       (8, lambda: (a * b - X) * (b * b - X) % p), (9, lambda: (a + b) * (a + b) % p),
                       (10, lambda: (a * (a + b) + b * b) % p)):
            assert shim.zkt_field_op(field, op, P(A), P(B), P(out)) == 0
            assert _int(out) == fn(), (field, op, hex(a), hex(b))
        if a and i < 12:
            assert shim.zkt_field_op(field, 3, P(A), None, P(out)) == 0
            assert _int(out) == pow(a, -1, p)
        if a and field >= 6:    # the lazy-limb types' safegcd inverse (fu_safegcd.hpp: the one inversion of a grand-product call)
            assert shim.zkt_field_op(field, 11, P(A), None, P(out)) == 0
            assert _int(out) == pow(a, -1, p), hex(a)


@pytest.mark.parametrize("field,curve", [(4, 0), (5, 1), (10, 0), (11, 1)])
def test_fq2_ops(shim, field, curve):
    C = CURVES[curve]
    p, F = C.p, po.Fq2(C.p)
    nl = FQ_LIMBS[curve] * 2
    random.seed(11)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    out = np.zeros(2 * nl, dtype=np.uint32)
    pack = lambda x: np.concatenate([_u32(x[0], nl), _u32(x[1], nl)])
    unpack = lambda o: (_int(o[:nl]), _int(o[nl:]))
    vals = [(0, 0), (1, 0), (0, 1), (p - 1, p - 1)] + [(random.randrange(p), random.randrange(p)) for _ in range(12)]
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 2) % len(vals)]
        X = F.sub(F.sqr(a), F.add(F.mul(a, b), F.add(F.sqr(b), F.sqr(b))))
        for op, fn in ((0, lambda: F.mul(a, b)), (1, lambda: F.add(a, b)), (2, lambda: F.sub(a, b)), (4, lambda: F.sqr(a)),
                       (5, lambda: F.neg(a)), (6, lambda: F.add(a, a)), (7, lambda: F.sub(a, b)),
                       (8, lambda: F.mul(F.sub(F.mul(a, b), X), F.sub(F.sqr(b), X))),
                       (10, lambda: F.add(F.mul(a, F.add(a, b)), F.sqr(b)))):
            assert shim.zkt_field_op(field, op, P(pack(a)), P(pack(b)), P(out)) == 0
            assert unpack(out) == fn(), (op, a, b)
        if not F.is_zero(a) and i < 8:
            assert shim.zkt_field_op(field, 3, P(pack(a)), None, P(out)) == 0
            assert unpack(out) == F.inv(a)


def _chain(shim, field, curve, group, pts, infs, negs, mode, k=0):
    arr = pts_arr(curve, group, pts).view(np.uint32).reshape(len(pts), -1) if len(pts) else np.zeros((0, 1), dtype=np.uint32)
    arr = np.ascontiguousarray(arr)
    ncoord = 3 if mode == 3 else 2
    out = np.zeros(ncoord * FQ_LIMBS[curve] * group * 2, dtype=np.uint32)
    oinf = np.zeros(1, dtype=np.uint8)
    infa = np.array(infs, dtype=np.uint8)
    nega = np.array(negs, dtype=np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert shim.zkt_point_chain(field, P(arr), P(infa), P(nega), ctypes.c_size_t(len(pts)), mode, ctypes.c_uint32(k), P(out), P(oinf)) == 0
    return out.view(np.uint64), int(oinf[0])


COORD_FIELDS = {(0, 1): (0, 6), (0, 2): (4, 10), (1, 1): (2, 7), (1, 2): (5, 11)}  # (saturated, lazy) ids


@pytest.mark.parametrize("lazy", [0, 1])
@pytest.mark.parametrize("curve,group", [(0, 1), (0, 2), (1, 1), (1, 2)])
def test_xyzz_group_law(shim, curve, group, lazy):
    field = COORD_FIELDS[(curve, group)][lazy]
    """madd chains incl. the special cases the bucket method meets: P+P, P+(-P), infinity operands."""
    C = CURVES[curve]
    G = C.g1 if group == 1 else C.g2
    rng = po.SplitMix64(curve * 10 + group)
    ks = [rng.next_mod(C.r) for _ in range(6)]
    base = G.batch_mul_gen(ks)
    P0, P1, P2 = base[0], base[1], base[2]
    cases = [
        ([P0, P1, P2, base[3], base[4], base[5]], [0] * 6, [0, 1, 0, 1, 1, 0]),
        ([P0, P0], [0, 0], [0, 0]),            # doubling inside madd
        ([P0, P0, P0, P0], [0] * 4, [0] * 4),  # 4P via dbl then adds
        ([P0, P0], [0, 0], [0, 1]),            # P + (-P) = infinity
        ([P0, P0, P1], [0, 0, 0], [0, 1, 0]),  # infinity then restart
        ([P0, P1], [1, 0], [0, 0]),            # infinity operand
        ([], [], []),
    ]
    sgn = lambda P, n: G.neg(P) if n else P
    for pts, infs, negs in cases:
        exp = None
        for Pt, i, n in zip(pts, infs, negs):
            if not i:
                exp = G.add(exp, sgn(Pt, n))
        out, oinf = _chain(shim, field, curve, group, pts, infs, negs, 0)
        assert pt_from_limbs(curve, group, out, oinf) == exp
        out, oinf = _chain(shim, field, curve, group, pts, infs, negs, 4)  # through the device-buffer layout
        assert pt_from_limbs(curve, group, out, oinf) == exp
        if len(pts) >= 2:
            out, oinf = _chain(shim, field, curve, group, pts, infs, negs, 1)  # xyzz_add of the two halves
            assert pt_from_limbs(curve, group, out, oinf) == exp
            for k in (0, 1, 2, 37, 32768, 65535):
                out, oinf = _chain(shim, field, curve, group, pts, infs, negs, 2, k)
                assert pt_from_limbs(curve, group, out, oinf) == G.mul(exp, k)
        out, oinf = _chain(shim, field, curve, group, pts, infs, negs, 3)  # XYZZ -> Jacobian
        L = FQ_LIMBS[curve] * group
        if exp is None:
            assert oinf == 1 and po.from_limbs(out[2 * L:3 * L]) == 0
        else:
            from util import jac_to_affine_py
            assert jac_to_affine_py(curve, group, out.reshape(3, L)) == exp
    # xyzz_add doubling branch: same chain in both halves
    pts = [P0, P1, P0, P1]
    out, oinf = _chain(shim, field, curve, group, pts, [0] * 4, [0] * 4, 1)
    assert pt_from_limbs(curve, group, out, oinf) == G.mul(G.add(P0, P1), 2)
    # and the cancelling branch
    out, oinf = _chain(shim, field, curve, group, pts, [0] * 4, [0, 0, 1, 1], 1)
    assert oinf == 1


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 16])
def test_signed_digit_recoding(shim, c):
    r = po.BLS12_381.r
    random.seed(c)
    W = (256 + c - 1) // c
    B = 1 << (c - 1)
    vals = [0, 1, 2, B, B + 1, (1 << c) - 1, 1 << c, r - 1, r - 2, (1 << 255) - 1] + [random.randrange(r) for _ in range(50)]
    dig = np.zeros(W, dtype=np.int32)
    for v in vals:
        s = _u32(v, 8)
        carry = shim.zkt_recode(s.ctypes.data_as(ctypes.c_void_p), c, W, dig.ctypes.data_as(ctypes.c_void_p))
        assert carry == 0
        assert all(-B < int(d) <= B for d in dig)
        assert sum(int(d) << (c * w) for w, d in enumerate(dig)) == v


@pytest.mark.parametrize("curve_id,curve", [(0, po.BLS12_381), (1, po.BN254)])
@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 14, 15, 16])
def test_folded_recoding(shim, curve_id, curve, c):
    """msm_digits_only: scalars folded to |s| <= (r-1)/2 so no window size leaves a carry-only top window."""
    r = curve.r
    random.seed(100 + c)
    B = 1 << (c - 1)
    vals = [0, 1, 2, B, B + 1, (r - 1) // 2, (r + 1) // 2, r - 1, r - 2, r, r + 5, (1 << 256) - 1] + [random.randrange(r) for _ in range(60)]
    dig = np.zeros(140, dtype=np.int32)
    for v in vals:
        s = _u32(v, 8)
        W = shim.zkt_recode_folded(curve_id, s.ctypes.data_as(ctypes.c_void_p), c, dig.ctypes.data_as(ctypes.c_void_p))
        tb = r.bit_length()
        assert W == (tb + c - 1) // c
        off = [w * tb // W for w in range(W + 1)]  # balanced windows: widths floor / ceil of tb / W, never above c
        assert all(1 <= off[w + 1] - off[w] <= c for w in range(W))
        assert all(abs(int(d)) <= (1 << (off[w + 1] - off[w] - 1)) for w, d in enumerate(dig[:W]))
        got = sum(int(d) << off[w] for w, d in enumerate(dig[:W]))
        assert got % r == v % r and abs(got) <= (r - 1) // 2


def test_generated_asm_blocks_compute_montgomery_products():
    """The device products are single inline-asm blocks written by tools/gen_mont_asm.py (crypto3-zk_amd/csrc/mont_asm.hpp); the host
    build uses the C++ form, so here the emitted instructions themselves are interpreted on the CPU (tools/sim_mont_asm.py) and
    checked against the big-integer Montgomery product: mul, sqr, mul2 at the three limb counts shipped, and the Karatsuba block
    of round 5's experiment against the shipped 14-limb block, limb for limb."""
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import sim_mont_asm as sim
    from gen_mont_asm import Block, KBlock
    rng = random.Random(17)
    for L, modulus, bits in ((9, sim.BLS_R, 257), (10, sim.BN_Q, 262), (14, sim.BLS_Q, 386)):
        for kind in ("mul", "sqr", "mul2"):
            blk = Block(L, 29, False, kind)
            blk.build()
            bound = bits - 1 if kind == "mul2" else bits
            cases = [{k: rng.randrange(1 << bound) for k in "abcd"} for _ in range(6)]
            cases.append({k: (1 << bound) - 1 for k in "abcd"})
            cases.append({"a": 0, "b": modulus - 1, "c": 1, "d": 0})
            for ops in cases:
                r, ok = sim.montgomery_check(blk, modulus, ops)
                assert ok, (L, kind, ops)
    k, u = KBlock(), Block(14, 29, False, "mul")
    k.build()
    u.build()
    assert k.mads == 343 and len(k.lines) == 466 and len(u.lines) == 461
    for _ in range(20):
        ops = {"a": rng.randrange(1 << 386), "b": rng.randrange(1 << 386)}
        assert sim.montgomery_check(k, sim.BLS_Q, ops) == sim.montgomery_check(u, sim.BLS_Q, ops)
