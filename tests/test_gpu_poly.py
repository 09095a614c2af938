"""Coefficient-form polynomial arithmetic behind the KZG opening proof (kzg_v2.hpp:236-305) through the C ABI,
against plain big-integer arithmetic: pointwise operators, batched evaluation, division by (X - z), and the
multi-polynomial accumulation f += theta_i (f_i - U_i) diffpoly_i."""
from functools import reduce

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


@pytest.mark.parametrize("curve", [0, 1])
def test_fr_vec_affine_and_mul_div(ctx, curve):
    """zkhip_fr_vec_affine_dev (a x + b y + c, y optional) and zkhip_fr_vec_mul_div_dev (a b / c over the first `count` entries, rows in
    chunks of 8 sharing an inversion) through the C ABI against big-integer arithmetic: edge values, counts that are no multiple of the
    chunk, in place, entries behind `count` untouched, empty."""
    r = CURVES[curve].r
    n = 1003
    x = fr_ints(cp.random_fr(curve, 71, n))
    y = fr_ints(cp.random_fr(curve, 72, n))
    z = [v or 1 for v in fr_ints(cp.random_fr(curve, 73, n))]
    x[:4] = [0, r - 1, 1, r - 1]
    y[:4] = [0, r - 1, r - 1, 1]
    z[:4] = [1, r - 1, 2, r - 2]
    d_x, d_y, d_z, d_o = (ctx.malloc(n * 32) for _ in range(4))
    ctx.h2d(d_x, fr_arr(x))
    ctx.h2d(d_y, fr_arr(y))
    ctx.h2d(d_z, fr_arr(z))
    got = np.zeros((n, 4), dtype=np.uint64)
    for a, b, c in ((3, 5, 7), (r - 1, r - 1, r - 1), (0, 1, 0), (1, 0, 0), (0, 0, r - 1), (12345678901234567890 % r, r - 2, 1)):
        ctx.fr_vec_affine_dev(curve, d_x, d_y, limbs(a, 4), limbs(b, 4), limbs(c, 4), d_o, n)
        ctx.d2h(got, d_o)
        assert fr_ints(got) == [(a * u + b * v + c) % r for u, v in zip(x, y)], (a, b, c)
        ctx.fr_vec_affine_dev(curve, d_x, 0, limbs(a, 4), None, limbs(c, 4), d_o, n)
        ctx.d2h(got, d_o)
        assert fr_ints(got) == [(a * u + c) % r for u in x], (a, c)
    ctx.fr_vec_affine_dev(curve, d_x, 0, limbs(2, 4), None, limbs(1, 4), d_o, 0)     # empty
    for count in (n, 1000, 9, 8, 7, 1):
        ctx.h2d(d_o, fr_arr([7] * n))
        ctx.fr_vec_mul_div_dev(curve, d_x, d_y, d_z, d_o, count)
        ctx.d2h(got, d_o)
        assert fr_ints(got) == [u * v % r * pow(w, -1, r) % r for u, v, w in zip(x[:count], y[:count], z[:count])] + [7] * (n - count), count
    ctx.fr_vec_mul_div_dev(curve, d_x, d_y, d_z, d_x, n)                             # in place
    ctx.d2h(got, d_x)
    assert fr_ints(got) == [u * v % r * pow(w, -1, r) % r for u, v, w in zip(x, y, z)]
    ctx.fr_vec_mul_div_dev(curve, d_x, d_y, d_z, d_o, 0)                             # empty
    for p in (d_x, d_y, d_z, d_o):
        ctx.free(p)


@pytest.mark.parametrize("curve,n,usable,k_in,k_val", [(0, 64, 61, 1, 1), (1, 8, 4, 0, 2), (0, 2051, 2040, 2, 0), (0, 4096, 4095, 1, 2)])
def test_lookup_grand_product_abi(ctx, curve, n, usable, k_in, k_val):
    """zkhip_lookup_grand_product_dev through the C ABI on RANDOM vectors (no closing product needed: the recurrence is what is checked)
    against the oracle's row-by-row loop (compute_V_L, lookup_argument.hpp:375-409): sizes that are no power of two or multiple of the
    chunk, no inputs / no values, usable_rows = n - 1, zeros behind usable_rows; usable_rows >= n is refused."""
    r = CURVES[curve].r
    rng = po.SplitMix64(8300 + n + k_in)
    vec = lambda: [rng.next_mod(r) for _ in range(n)]
    inputs, values, sorted_ = [vec() for _ in range(k_in)], [vec() for _ in range(k_val)], [vec() for _ in range(k_in + k_val)]
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    exp = po.lookup_grand_product(inputs, values, sorted_, beta, gamma, usable, r)
    ptrs = []
    for v in inputs + values + sorted_:
        d = ctx.malloc(n * 32)
        ctx.h2d(d, fr_arr(v))
        ptrs.append(d)
    d_v = ctx.malloc(n * 32)
    ctx.lookup_grand_product_dev(curve, ptrs[:k_in], ptrs[k_in:k_in + k_val], ptrs[k_in + k_val:], n, usable, limbs(beta, 4), limbs(gamma, 4), d_v)
    got = np.zeros((n, 4), dtype=np.uint64)
    ctx.d2h(got, d_v)
    assert fr_ints(got) == exp
    with pytest.raises(Exception):
        ctx.lookup_grand_product_dev(curve, ptrs[:k_in], ptrs[k_in:k_in + k_val], ptrs[k_in + k_val:], n, n, limbs(beta, 4), limbs(gamma, 4), d_v)
    for p in ptrs + [d_v]:
        ctx.free(p)


@pytest.mark.parametrize("curve", [0, 1])
def test_grand_products_zero_denominator(ctx, curve):
    """one inversion per CALL must not change what a zero denominator does.  The reference inverts per row (0^-1 = 0 in
    its field type): the row's ratio is 0, V keeps its values up to that row and is zero behind it; a zero c[j] of a b / c zeroes out[j] alone.
    Forced here: a column entry chosen so that h = column + beta S_sigma + gamma vanishes at two rows (the FIRST decides), a sorted pair that
    kills the lookup denominator, zeros in c -- against the oracle's row-by-row loops with that inverse."""
    r = CURVES[curve].r
    n, k = 700, 2
    rng = po.SplitMix64(4400 + curve)
    vec = lambda: [rng.next_mod(r) for _ in range(n)]
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    cols, sid, ssig = [vec() for _ in range(k)], [vec() for _ in range(k)], [vec() for _ in range(k)]
    for z in (333, 520):
        cols[1][z] = (-(beta * ssig[1][z] + gamma)) % r         # h_1[z] = 0
    g, h, V = po.permutation_grand_product(cols, sid, ssig, beta, gamma, r)
    assert V[333] != 0 and V[334:] == [0] * (n - 334)
    ptrs = []
    for v in cols + sid + ssig:
        d = ctx.malloc(n * 32)
        ctx.h2d(d, fr_arr(v))
        ptrs.append(d)
    d_g, d_h, d_v = ctx.malloc(k * n * 32), ctx.malloc(k * n * 32), ctx.malloc(n * 32)
    ctx.perm_grand_product_dev(curve, ptrs[:k], ptrs[k:2 * k], ptrs[2 * k:], n, limbs(beta, 4), limbs(gamma, 4), d_g, d_h, d_v)
    got = np.zeros((n, 4), dtype=np.uint64)
    ctx.d2h(got, d_v)
    assert fr_ints(got) == V
    gh = np.zeros((k * n, 4), dtype=np.uint64)
    ctx.d2h(gh, d_h)
    assert fr_ints(gh) == [x for v in h for x in v]
    # the lookup product: sorted[0][j] + beta sorted[0][j + 1] + (1 + beta) gamma = 0 at row 100
    inputs, values, sorted_ = [vec()], [vec()], [vec(), vec()]
    usable = n - 5
    sorted_[0][100] = (-(beta * sorted_[0][101] + (1 + beta) * gamma)) % r
    VL = po.lookup_grand_product(inputs, values, sorted_, beta, gamma, usable, r)
    assert VL[100] != 0 and VL[101:] == [0] * (n - 101)
    lp = []
    for v in inputs + values + sorted_:
        d = ctx.malloc(n * 32)
        ctx.h2d(d, fr_arr(v))
        lp.append(d)
    ctx.lookup_grand_product_dev(curve, lp[:1], lp[1:2], lp[2:], n, usable, limbs(beta, 4), limbs(gamma, 4), d_v)
    ctx.d2h(got, d_v)
    assert fr_ints(got) == VL
    # a b / c with zeros in c (also the first and the last row, and a whole lane chunk)
    a, b, c = vec(), vec(), vec()
    for j in [0, 5, 64, 65, 66, 67, 68, 69, 70, 71, 72, n - 1]:
        c[j] = 0
    d_a, d_b, d_c = (ctx.malloc(n * 32) for _ in range(3))
    for d, v in ((d_a, a), (d_b, b), (d_c, c)):
        ctx.h2d(d, fr_arr(v))
    ctx.fr_vec_mul_div_dev(curve, d_a, d_b, d_c, d_v, n)
    ctx.d2h(got, d_v)
    assert fr_ints(got) == [x * y % r * po.inv0(w, r) % r for x, y, w in zip(a, b, c)]
    for p_ in ptrs + lp + [d_g, d_h, d_v, d_a, d_b, d_c]:
        ctx.free(p_)


def _lookup_sort_on_device(ctx, inputs, values, n, usable):
    ptrs = []
    for v in inputs + values:
        d = ctx.malloc(n * 32)
        ctx.h2d(d, fr_arr(v))
        ptrs.append(d)
    outs = [ctx.malloc(n * 32) for _ in range(len(inputs) + len(values))]
    junk = fr_arr([5] * n)
    for d in outs:
        ctx.h2d(d, junk)        # every entry of every vector must be written
    ctx.lookup_sort_dev(ptrs[:len(inputs)], ptrs[len(inputs):], n, usable, outs)
    got = []
    for d in outs:
        a = np.zeros((n, 4), dtype=np.uint64)
        ctx.d2h(a, d)
        got.append(fr_ints(a))
    for p in ptrs + outs:
        ctx.free(p)
    import ctypes
    flags = ctypes.c_uint32()
    ctx.lib.zkhip_device_status(ctx.h, ctypes.byref(flags))     # reads and clears; ZKHIP_ERR_RANGE when a flag was set
    return got, flags.value


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("log_n,k_in,k_val", [(6, 1, 1), (7, 2, 1), (8, 3, 2), (10, 2, 2), (12, 1, 1), (14, 2, 1), (14, 1, 3)])
def test_lookup_sort_polynomials_genuine(ctx, curve, log_n, k_in, k_val):
    """zkhip_lookup_sort_dev = sort_polynomials (lookup_argument.hpp:565-638) on GENUINE instances (tests/util.lookup_instance: every table
    column starts with a zero, holds distinct values with one repeated in adjacent rows, zeros behind; every input draws from the tables or
    is zero), 2^6 - 2^14 rows, both curves, against the oracle's restatement of the reference's map + walk: every entry of every vector,
    the stitch at usable_rows and the zero tails included; no status flag."""
    from util import lookup_instance
    C = CURVES[curve]
    rng = po.SplitMix64(9100 + log_n + 7 * k_in + curve)
    inputs, values, usable = lookup_instance(C, rng, log_n, k_in, k_val)
    n = 1 << log_n
    exp = po.lookup_sort_polynomials(inputs, values, n, usable)
    got, flags = _lookup_sort_on_device(ctx, inputs, values, n, usable)
    assert flags == 0
    assert got == exp


def test_lookup_sort_polynomials_edge_cases(ctx):
    """The corners of the reference's walk (lookup_argument.hpp:601-631), small enough to read: a table that starts on a non-zero value (the
    walk's virtual zero is emitted first), zero runs in the middle (one zero each) and at the end (nothing), a run of equal values that
    spans the boundary between two table columns, no inputs, no table; then what the reference only survives without its assertions --
    a looked-up value that is in no table (ignored, status bit 2), a value whose table entries are NOT adjacent (emitted `count` times per
    run), an emitted sequence that does not fit (status bit 3, the excess dropped)."""
    n, u = 16, 12
    pad = lambda v: v + [0] * (n - len(v))
    cases = [
        ([pad([7, 7, 9, 3, 3, 3, 0, 0, 9, 9, 7, 0])], [pad([7, 7, 0, 0, 9, 9, 3, 0, 0, 0, 0, 0])]),                      # leading zero, zero runs
        ([pad([5] * 12)], [pad([0, 5, 5, 5, 0, 0, 0, 0, 0, 0, 0, 0])]),                                                 # one hot value
        ([], [pad([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]), pad([11, 11, 12, 0, 0, 0, 0, 0, 0, 0, 0, 0])]),              # no inputs; a run across two columns
        ([pad([0] * 12), pad([2, 2, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0])], [pad([0, 0, 0, 2, 2, 0, 0, 0, 0, 0, 0, 0])]),     # zero inputs, zero runs around a value
        ([pad([1] * 12)], [pad([0] + [1] * 11)]),                                                                        # a table of one value: fills both vectors to the last entry
    ]
    for inputs, values in cases:
        exp = po.lookup_sort_polynomials(inputs, values, n, u)
        got, flags = _lookup_sort_on_device(ctx, inputs, values, n, u)
        assert flags == 0 and got == exp, (inputs, values)
    # no table at all: nothing is emitted, every looked-up value is foreign
    got, flags = _lookup_sort_on_device(ctx, [pad([0] * 12)], [], n, u)
    assert got == [[0] * n] and flags == 4
    # a foreign input: ignored (the reference without assertions counts a key it never emits), flagged
    inputs, values = [pad([4, 4, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0])], [pad([0, 4, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0])]
    got, flags = _lookup_sort_on_device(ctx, inputs, values, n, u)
    assert flags == 4 and got == po.lookup_sort_polynomials(inputs, values, n, u, strict=False)
    # the same value in two separate runs: `count` copies per run, as the walk does; fits here, no flag
    inputs, values = [pad([4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])], [pad([0, 4, 6, 4, 0, 0, 0, 0, 0, 0, 0, 0])]
    got, flags = _lookup_sort_on_device(ctx, inputs, values, n, u)
    exp = po.lookup_sort_polynomials(inputs, values, n, u, strict=False)
    assert flags == 0 and got == exp and exp[0][:9] == [0, 4, 4, 4, 6, 4, 4, 4, 0]
    # ... and does not fit here: flagged, the vectors hold the head of the sequence
    inputs, values = [pad([4] * 12)], [pad([4, 6, 4, 6, 4, 6, 4, 6, 4, 6, 4, 6])]
    got, flags = _lookup_sort_on_device(ctx, inputs, values, n, u)
    assert flags == 8
    flat = [0] + sum(([4] * 18 if i % 2 == 0 else [6] * 6 for i in range(12)), [])
    assert got[0][:u] == flat[:u] and got[1][:u] == flat[u:2 * u] and got[0][u] == flat[u] and got[1][u:] == [0] * (n - u)
    with pytest.raises(Exception):
        _lookup_sort_on_device(ctx, inputs, values, n, n)       # usable_rows >= n is refused


def test_lookup_sort_overflow_that_would_wrap_32_bits(ctx):
    """a table whose equal values are NOT adjacent emits `count` copies per run; 2^17 rows alternating two values are 2^17 runs
    of 2^16 copies each -- 2^33 + 1 emitted entries, which a u32 total wraps to 1: the overflow flag stayed down in exactly the malformed
    case it exists for, and the run offsets stopped being monotone.  The scans saturate now: the flag is raised and the vectors hold the
    HEAD of the sequence (a zero, then count(a) copies of a, count(b) copies of b, ...)."""
    u, n = 1 << 17, 1 << 18
    a, b = 4, 6
    values = np.zeros((n, 4), dtype=np.uint64)
    values[0:u:2, 0] = a
    values[1:u:2, 0] = b
    inputs = np.zeros((n, 4), dtype=np.uint64)
    inputs[:u, 0] = a                                        # every looked-up value is in the table: a occurs 2^16 + 2^17 times
    ptrs = [ctx.malloc(n * 32) for _ in range(4)]
    ctx.h2d(ptrs[0], inputs)
    ctx.h2d(ptrs[1], values)
    for d in ptrs[2:]:
        ctx.h2d(d, np.full((n, 4), 5, dtype=np.uint64))
    ctx.lookup_sort_dev(ptrs[:1], ptrs[1:2], n, u, ptrs[2:])
    import ctypes
    flags = ctypes.c_uint32()
    ctx.lib.zkhip_device_status(ctx.h, ctypes.byref(flags))
    assert flags.value == 8
    seq = np.concatenate([[0], np.repeat([a], 3 << 16), np.repeat([b], 1 << 16)])[: 2 * u]   # run 0: every copy of a; run 1: every copy of b; ...
    got = [np.zeros((n, 4), dtype=np.uint64) for _ in range(2)]
    for g, d in zip(got, ptrs[2:]):
        ctx.d2h(g, d)
    assert (got[0][:u, 0] == seq[:u]).all() and (got[1][:u, 0] == seq[u:]).all() and not got[0][:, 1:].any() and not got[1][:, 1:].any()
    assert got[0][u, 0] == seq[u] and not got[1][u:].any() and not got[0][u + 1:].any()
    for p in ptrs:
        ctx.free(p)


def test_block_cache_behind_malloc_free(ctx):
    """zkhip_malloc / zkhip_free keep freed blocks for the next request of their size class (option alloc_cache_mb): a block
    comes back for an equal or slightly smaller request, not for a much smaller or a larger one; its contents are whatever the last
    owner left (no clearing promised, none relied on: every kernel writes what it reads back); alloc_cache_mb = 0 empties the cache and
    turns it off; and the cache never hands out a block that is still live."""
    assert ctx.get_option("alloc_cache_mb") == 16384
    a = ctx.malloc(3 << 20)
    ctx.free(a)
    b = ctx.malloc(3 << 20)
    assert b == a                                            # the cached block
    c = ctx.malloc(3 << 20)
    assert c != b                                            # b is live: a second block
    ctx.free(b)
    d = ctx.malloc((3 << 20) - 70000)                        # within an eighth of the block: taken
    assert d == b
    ctx.free(d)
    e = ctx.malloc(1 << 20)                                  # a third of the block: not taken
    assert e != b
    f = ctx.malloc(4 << 20)                                  # larger: not taken
    assert f not in (b, c, e)
    data = np.arange(1 << 17, dtype=np.uint64)
    ctx.h2d(f, data)
    back = np.zeros_like(data)
    ctx.d2h(back, f)
    assert (back == data).all()
    ctx.set_option("alloc_cache_mb", 0)                      # flush + off
    assert ctx.get_option("alloc_cache_mb") == 0
    for p in (c, e, f):
        ctx.free(p)
    g = ctx.malloc(3 << 20)
    ctx.free(g)
    ctx.set_option("alloc_cache_mb", 16384)
    ctx.free(0)                                              # a null pointer is accepted
    # a block is settled with the context that ALLOCATED it, whichever context it is freed through -- no stale entry that the
    # owner's zkhip_destroy would free a second time; a block its caller still holds survives the owner's destruction
    import sys
    zk = sys.modules["crypto3_zk_amd"]
    other = zk.Context(0)
    p1, p2 = other.malloc(5 << 20), other.malloc(6 << 20)
    ctx.free(p1)                                             # through ANOTHER context: lands in `other`'s cache
    assert other.malloc(5 << 20) == p1                       # ... from where `other` hands it out again
    ctx.free(p1)
    with pytest.raises(Exception):
        ctx.free(p1)                                         # a double free is refused instead of corrupting the cache
    other.close(release_blocks=False)                        # p2 is still held: zkhip_destroy must not free it (the Python wrapper would)
    data = np.arange(1 << 10, dtype=np.uint64)
    ctx.h2d(p2, data)
    back = np.zeros_like(data)
    ctx.d2h(back, p2)
    assert (back == data).all()
    ctx.free(p2)                                             # ownerless now: handed to the driver
    # ... and the wrapper's default: close() hands back what its caller forgot, so nothing leaks for the life of the process
    third = zk.Context(0)
    q = third.malloc(7 << 20)
    assert q in third._live
    third.close()
    assert not third._live and q not in zk.zkhip._BLOCK_HOLDER


@pytest.mark.parametrize("curve,log_size", [(0, 6), (1, 9), (0, 14), (1, 16)])
def test_gate_eval_flat_program(zk, ctx, curve, log_size):
    """zkhip_gate_eval_dev: the gate argument's sum as ONE launch over a flat program (gates_argument.hpp:93-121, 203-216 once the
    expressions are monomials) against the same sum from big integers / the oracle's pointwise arithmetic: gates with and without a
    selector, shared factors, a repeated column, rotations of both signs that wrap, a constant term, more than 16 terms in one gate
    (the unreduced sums fold), an empty gate; with and without mask; accumulate."""
    C = CURVES[curve]
    r, size = C.r, 1 << log_size
    n_slots = 6
    cols = [cp.random_fr(curve, 700 + s, size) for s in range(n_slots)]
    cols[5][::3] = 0  # a selector-like column
    rng = po.SplitMix64(55 + log_size)
    big_gate = (None, [(rng.next_mod(r), [(t % 5, (t % 7) - 3), ((t + 1) % 5, 0)]) for t in range(37)])
    gates = [((5, 0), [(rng.next_mod(r), [(0, 0), (1, 1)]), (r - 1, [(2, -1)])]),
             ((5, 2), [(rng.next_mod(r), [(0, 2), (0, 0), (3, -2)]), (7, [])]),
             (None, [(rng.next_mod(r), [(4, 1), (4, 1), (4, size - 1)])]),
             big_gate, (None, []), ((3, -5), [(1, [])])]

    def expect(gs, mask=None, prev=None):
        acc = np.zeros((size, 4), dtype=np.uint64) if prev is None else prev
        for sel, terms in gs:
            g = np.zeros((size, 4), dtype=np.uint64)
            for c, factors in terms:
                t = np.repeat(fr_arr([c]), size, axis=0)
                for sl, rot in factors:
                    t = cp.fr_vec(curve, 2, t, np.roll(cols[sl], -rot, axis=0))
                g = cp.fr_vec(curve, 0, g, t)
            if sel is not None:
                g = cp.fr_vec(curve, 2, g, np.roll(cols[sel[0]], -sel[1], axis=0))
            acc = cp.fr_vec(curve, 0, acc, g)
        return cp.fr_vec(curve, 2, acc, mask) if mask is not None else acc

    d_slots = [ctx.malloc(size * 32) for _ in range(n_slots)]
    for p, c in zip(d_slots, cols):
        ctx.h2d(p, c)
    mask = cp.random_fr(curve, 799, size)
    mask[-5:] = 0
    d_mask, d_out = ctx.malloc(size * 32), ctx.malloc(size * 32)
    ctx.h2d(d_mask, mask)
    out = np.zeros((size, 4), dtype=np.uint64)
    ctx.gate_eval_dev(curve, gates, d_slots, log_size, d_out, d_mask)
    ctx.d2h(out, d_out)
    want = expect(gates, mask)
    assert (out == want).all()
    if log_size <= 9:  # and from big integers, row by row
        ci = [fr_ints(c) for c in cols]
        mi = fr_ints(mask)
        for j in (0, 1, size // 2, size - 1):
            tot = 0
            for sel, terms in gates:
                g = sum(c * reduce(lambda a, b: a * b % r, [ci[sl][(j + rot) % size] for sl, rot in fs], 1) for c, fs in terms) % r
                tot += g * (ci[sel[0]][(j + sel[1]) % size] if sel is not None else 1)
            assert po.from_limbs(out[j]) == tot * mi[j] % r
    # in two pieces: the first without mask, the second accumulates and carries the mask
    ctx.gate_eval_dev(curve, gates[:2], d_slots, log_size, d_out)
    ctx.d2h(out, d_out)
    assert (out == expect(gates[:2])).all()
    ctx.gate_eval_dev(curve, gates[2:], d_slots, log_size, d_out, d_mask, accumulate=True)
    ctx.d2h(out, d_out)
    assert (out == want).all()
    # a malformed program is refused on the host
    with pytest.raises(zk.ZkhipError):
        ctx.gate_eval_dev(curve, [(None, [(1, [(n_slots, 0)])])], d_slots, log_size, d_out)
    with pytest.raises(zk.ZkhipError):
        ctx.gate_eval_dev(curve, [((n_slots + 3, 0), [(1, [(0, 0)])])], d_slots, log_size, d_out)
    for p in d_slots + [d_mask, d_out]:
        ctx.free(p)
