"""Coefficient-form polynomial arithmetic behind the KZG opening proof (kzg_v2.hpp:236-305) through the C ABI,
against plain big-integer arithmetic: pointwise operators, batched evaluation, division by (X - z), and the
multi-polynomial accumulation f += theta_i (f_i - U_i) diffpoly_i."""
import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, fr_arr, fr_ints, limbs

pytestmark = pytest.mark.gpu


def _horner(c, z, r):
    acc = 0
    for x in reversed(c):
        acc = (acc * z + x) % r
    return acc


@pytest.mark.parametrize("curve", [0, 1])
def test_fr_vec_ops(ctx, curve):
    r = CURVES[curve].r
    n = 1003
    a = fr_ints(cp.random_fr(curve, 61, n))
    b = fr_ints(cp.random_fr(curve, 62, n))
    a[:4] = [0, r - 1, 1, r - 1]
    b[:4] = [0, r - 1, r - 1, 1]
    d_a, d_b, d_o = ctx.malloc(n * 32), ctx.malloc(n * 32), ctx.malloc(n * 32)
    ctx.h2d(d_a, fr_arr(a))
    ctx.h2d(d_b, fr_arr(b))
    got = np.zeros((n, 4), dtype=np.uint64)
    for op, f in ((0, lambda x, y: (x + y) % r), (1, lambda x, y: (x - y) % r), (2, lambda x, y: x * y % r)):
        ctx.fr_vec_op_dev(curve, op, d_a, d_b, d_o, n)
        ctx.d2h(got, d_o)
        assert fr_ints(got) == [f(x, y) for x, y in zip(a, b)], op
    ctx.fr_vec_op_dev(curve, 2, d_a, d_b, d_a, n)  # in place
    ctx.d2h(got, d_a)
    assert fr_ints(got) == [x * y % r for x, y in zip(a, b)]
    ctx.fr_vec_op_dev(curve, 0, d_a, d_b, d_o, 0)  # empty
    for p in (d_a, d_b, d_o):
        ctx.free(p)


@pytest.mark.parametrize("curve,n", [(0, 1), (0, 33), (0, 8192), (0, 8193), (1, 20011), (0, 70000)])
def test_poly_eval_and_div(ctx, curve, n):
    r = CURVES[curve].r
    batch = 3
    polys = [fr_ints(cp.random_fr(curve, 70 + b, n)) for b in range(batch)]
    polys[1][-1] = 0  # top coefficient zero
    pts = [0, 1, r - 1, po.SplitMix64(9).next_mod(r), po.SplitMix64(10).next_mod(r)]
    d = ctx.malloc(batch * n * 32)
    ctx.h2d(d, fr_arr([c for p in polys for c in p]))
    got = ctx.poly_eval_dev(curve, d, n, batch, fr_arr(pts))
    for b in range(batch):
        assert fr_ints(got[b]) == [_horner(polys[b], z, r) for z in pts], b
    # division of polynomial 0 by (X - z): quotient * (X - z) + remainder == f
    for z in (pts[3], 0, 1):
        d_q = ctx.malloc(n * 32)
        rem = ctx.poly_div_linear_dev(curve, d, n, limbs(z, 4), d_q)
        out = np.zeros((n, 4), dtype=np.uint64)
        ctx.d2h(out, d_q)
        g = fr_ints(out)
        assert po.from_limbs(rem) == g[0] == _horner(polys[0], z, r)
        q = g[1:]
        back = [((q[j - 1] if j >= 1 else 0) - z * (q[j] if j < n - 1 else 0)) % r for j in range(n)]
        back[0] = (back[0] + g[0]) % r
        assert back == polys[0]
        ctx.free(d_q)
    # in place, on an exactly divisible polynomial: f = (X - z) * polys[1]
    z = pts[4]
    f = [((polys[1][j - 1] if j >= 1 else 0) - z * (polys[1][j] if j < n else 0)) % r for j in range(n + 1)]
    d_f = ctx.malloc((n + 1) * 32)
    ctx.h2d(d_f, fr_arr(f))
    rem = ctx.poly_div_linear_dev(curve, d_f, n + 1, limbs(z, 4), d_f)
    out = np.zeros((n + 1, 4), dtype=np.uint64)
    ctx.d2h(out, d_f)
    assert po.from_limbs(rem) == 0 and fr_ints(out[1:]) == polys[1]
    ctx.free(d_f)
    ctx.free(d)


@pytest.mark.parametrize("curve", [0, 1])
def test_poly_lincomb(ctx, curve):
    r = CURVES[curve].r
    lens = [5000, 4096, 1, 4999]
    taps = 3
    polys = [fr_ints(cp.random_fr(curve, 80 + i, n)) for i, n in enumerate(lens)]
    coeffs = [[po.SplitMix64(100 + 10 * i + t).next_mod(r) for t in range(taps)] for i in range(len(lens))]
    coeffs[1][2] = 0
    coeffs[3] = [r - 1, 0, 1]
    acc_len = 5002
    ds = []
    for p in polys:
        d = ctx.malloc(len(p) * 32)
        ctx.h2d(d, fr_arr(p))
        ds.append(d)
    exp = [0] * acc_len
    for p, c in zip(polys, coeffs):
        for t in range(taps):
            for j, x in enumerate(p):
                if j + t < acc_len:
                    exp[j + t] = (exp[j + t] + c[t] * x) % r
    d_acc = ctx.malloc(acc_len * 32)
    flat = fr_arr([x for c in coeffs for x in c])
    ctx.poly_lincomb_dev(curve, ds, lens, flat, taps, d_acc, acc_len, False)
    out = np.zeros((acc_len, 4), dtype=np.uint64)
    ctx.d2h(out, d_acc)
    assert fr_ints(out) == exp
    ctx.poly_lincomb_dev(curve, ds, lens, flat, taps, d_acc, acc_len, True)  # accumulate: doubles
    ctx.d2h(out, d_acc)
    assert fr_ints(out) == [2 * x % r for x in exp]
    # many terms per output (the periodic fold of the lazy accumulator): 40 copies of one polynomial, one tap
    many = 40
    cs = [po.SplitMix64(300 + i).next_mod(r) for i in range(many)]
    ctx.poly_lincomb_dev(curve, [ds[0]] * many, [lens[0]] * many, fr_arr(cs), 1, d_acc, lens[0], False)
    out = np.zeros((lens[0], 4), dtype=np.uint64)
    ctx.d2h(out, d_acc)
    s = sum(cs) % r
    assert fr_ints(out) == [s * x % r for x in polys[0]]
    for d in ds + [d_acc]:
        ctx.free(d)
