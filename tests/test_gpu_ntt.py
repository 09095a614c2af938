"""Parity of the HIP NTT (through the C ABI) against the oracle: every size 2^0..2^13, batches,
forward / inverse / coset, the tunables that change the pass structure, and BASELINE config 3 (2^22 x 8)
through a bit-exact comparison plus round-trip and random-point (Horner) properties."""
import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, fr_arr, fr_ints, limbs, qap_domains

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", [0, 1])
def test_ntt_all_small_sizes(ctx, curve):
    C = CURVES[curve]
    g = limbs(C.fr_generator, 4)
    for log_m in range(0, 14):
        m = 1 << log_m
        batch = 3 if log_m < 12 else 1
        w = limbs(C.root_of_unity(log_m), 4)
        a = cp.random_fr(curve, 1000 + log_m, batch * m).reshape(batch, m, 4)
        exp = cp.ntt(curve, a, log_m, w)
        got = ctx.ntt(curve, a, log_m, w)
        assert (got == exp).all(), (curve, log_m)
        if log_m <= 6:
            assert fr_ints(got[0]) == po.dft_naive(fr_ints(a[0]), C.root_of_unity(log_m), C.r)
        assert (ctx.ntt(curve, got, log_m, w, inverse=True) == a).all()
        expc = cp.ntt(curve, a, log_m, w, coset=g)
        gotc = ctx.ntt(curve, a, log_m, w, coset=g)
        assert (gotc == expc).all()
        assert (ctx.ntt(curve, gotc, log_m, w, inverse=True, coset=g) == a).all()
        assert (ctx.ntt(curve, a, log_m, w, inverse=True, coset=g) == cp.ntt(curve, a, log_m, w, inverse=True, coset=g)).all()


def test_ntt_pass_structure_options(ctx):
    C = po.BLS12_381
    log_m = 11
    m = 1 << log_m
    w = limbs(C.root_of_unity(log_m), 4)
    a = cp.random_fr(0, 9, 2 * m).reshape(2, m, 4)
    exp = cp.ntt(0, a, log_m, w)
    try:
        for radix in (1, 2, 3, 4, 5, 6, 8, 10):
            for tile in (0, 1, 3, 4):
                for pair in (0, 1, 2):  # 1 / 2 / 4 polynomials of the (even) batch per workgroup
                    ctx.set_option("ntt_radix_log", radix)
                    ctx.set_option("ntt_tile_log", tile)
                    ctx.set_option("ntt_pair", pair)
                    assert (ctx.ntt(0, a, log_m, w) == exp).all(), (radix, tile, pair)
        a4 = cp.random_fr(0, 10, 4 * m).reshape(4, m, 4)
        ctx.set_option("ntt_radix_log", 6)
        ctx.set_option("ntt_pair", 2)
        assert (ctx.ntt(0, a4, log_m, w) == cp.ntt(0, a4, log_m, w)).all()
        assert (ctx.ntt(0, a4[:3], log_m, w) == cp.ntt(0, a4[:3], log_m, w)).all()  # odd batch: falls back to one per workgroup
    finally:
        ctx.set_option("ntt_radix_log", 8)
        ctx.set_option("ntt_tile_log", 3)
        ctx.set_option("ntt_pair", 1)


def test_ntt_edge_values(ctx):
    """all-zero, all-(r-1), delta and constant vectors; non-standard omega (another primitive root)."""
    C = po.BLS12_381
    log_m = 9
    m = 1 << log_m
    w0 = C.root_of_unity(log_m)
    w = limbs(pow(w0, 5, C.r), 4)  # any odd power is again a primitive m-th root: omega is an argument
    vecs = np.zeros((4, m, 4), dtype=np.uint64)
    vecs[1, :] = limbs(C.r - 1, 4)
    vecs[2, 0] = limbs(C.r - 1, 4)
    vecs[3, :] = limbs(7, 4)
    exp = cp.ntt(0, vecs, log_m, w)
    assert (ctx.ntt(0, vecs, log_m, w) == exp).all()
    assert fr_ints(exp[3])[0] == 7 * m % C.r and all(v == 0 for v in fr_ints(exp[3])[1:])


def test_ntt_full_size_2_22_batch_8(ctx):
    """BASELINE config 3.  Bit-exact against the oracle on every polynomial; round trip on the batch;
    out[i] = f(omega^i) checked by Horner evaluation at sampled points."""
    C = po.BLS12_381
    log_m, batch = 22, 8
    m = 1 << log_m
    w_int = C.root_of_unity(log_m)
    w = limbs(w_int, 4)
    a = cp.random_fr(0, 1, batch * m).reshape(batch, m, 4)
    got = ctx.ntt(0, a, log_m, w)
    exp = cp.ntt(0, a, log_m, w)
    assert (got == exp).all()
    for b, i in ((0, 0), (0, 1), (3, 123456), (7, m - 1), (5, m // 2)):
        x = limbs(pow(w_int, i, C.r), 4)
        assert (cp.fr_horner(0, a[b], x) == got[b, i]).all()
    assert (ctx.ntt(0, got, log_m, w, inverse=True) == a).all()
    g = limbs(C.fr_generator, 4)
    gotc = ctx.ntt(0, a[:2], log_m, w, coset=g)
    assert (gotc == cp.ntt(0, a[:2], log_m, w, coset=g)).all()
    assert (ctx.ntt(0, gotc, log_m, w, inverse=True, coset=g) == a[:2]).all()


@pytest.mark.parametrize("curve,log_m", [(0, 24), (1, 23), (0, 17)])
def test_ntt_large_single_polynomial(ctx, curve, log_m):
    """sizes beyond the BASELINE configs (four-pass structure at 2^24, BN254's 2^28-adic field): bit-exact against the
    oracle, out[i] = f(omega^i) by Horner at sampled points, round trip, coset round trip"""
    C = CURVES[curve]
    m = 1 << log_m
    w_int = C.root_of_unity(log_m)
    w = limbs(w_int, 4)
    a = cp.random_fr(curve, 7, m).reshape(1, m, 4)
    got = ctx.ntt(curve, a, log_m, w)
    assert (got == cp.ntt(curve, a, log_m, w)).all()
    for i in (0, 1, 54321, m - 1, m // 2 + 3):
        assert (cp.fr_horner(curve, a[0], limbs(pow(w_int, i, C.r), 4)) == got[0, i]).all()
    assert (ctx.ntt(curve, got, log_m, w, inverse=True) == a).all()
    g = limbs(C.fr_generator, 4)
    assert (ctx.ntt(curve, ctx.ntt(curve, a, log_m, w, coset=g), log_m, w, inverse=True, coset=g) == a).all()


@pytest.mark.parametrize("curve,log_n,expand", [(0, 8, 2), (1, 6, 1), (0, 12, 3)])
def test_lpc_resize_and_fold(ctx, zk, curve, log_n, expand):
    """polynomial_dfs::resize as precommit<FRI> uses it (basic_fri.hpp:452-455) and the DFS fold_polynomial
    (fold_polynomial.hpp:68-93), against the oracle / the formula on big integers."""
    import ctypes

    C = CURVES[curve]
    n, batch = 1 << log_n, 3
    log_out = log_n + expand
    m = 1 << log_out
    wn, wm = limbs(C.root_of_unity(log_n), 4), limbs(C.root_of_unity(log_out), 4)
    evals = cp.random_fr(curve, 77, batch * n).reshape(batch, n, 4)
    coeffs = cp.ntt(curve, evals, log_n, wn, inverse=True)
    padded = np.zeros((batch, m, 4), dtype=np.uint64)
    padded[:, :n] = coeffs
    exp = cp.ntt(curve, padded, log_out, wm)
    d_in, d_out = ctx.malloc(evals.nbytes), ctx.malloc(exp.nbytes)
    ctx.h2d(d_in, evals)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = ctx.lib.zkhip_poly_resize_dev(ctx.h, curve, ctypes.c_void_p(d_in), ctypes.c_size_t(log_n), ctypes.c_size_t(batch), P(wn),
                                       ctypes.c_void_p(d_out), ctypes.c_size_t(log_out), P(wm))
    assert rc == 0
    got = np.zeros_like(exp)
    ctx.d2h(got, d_out)
    assert (got == exp).all()
    # the extension agrees with the original evaluations on the sub-domain (omega_m^(2^expand) = omega_n)
    assert (got[:, :: 1 << expand] == evals).all()
    # fold the first extended polynomial with a random alpha
    r = C.r
    alpha = po.SplitMix64(5).next_mod(r)
    f = fr_ints(got[0])
    winv = pow(C.root_of_unity(log_out), -1, r)
    half, inv2 = m // 2, pow(2, -1, r)
    expf = [inv2 * ((1 + alpha * pow(winv, i, r)) * f[i] + (1 - alpha * pow(winv, i, r)) * f[half + i]) % r for i in range(half)]
    d_f = ctx.malloc(half * 32)
    al = limbs(alpha, 4)
    assert ctx.lib.zkhip_fri_fold_dev(ctx.h, curve, ctypes.c_void_p(d_out), ctypes.c_size_t(log_out), P(al), P(wm), ctypes.c_void_p(d_f)) == 0
    gotf = np.zeros((half, 4), dtype=np.uint64)
    ctx.d2h(gotf, d_f)
    assert fr_ints(gotf) == expf
    for p in (d_in, d_out, d_f):
        ctx.free(p)


@pytest.mark.parametrize("curve,log_n,expand,batch", [(0, 1, 1, 1), (0, 3, 4, 2), (1, 8, 1, 1), (0, 9, 2, 5), (1, 10, 3, 2), (0, 13, 3, 1), (0, 16, 2, 3), (1, 17, 1, 1),
                                                      (0, 17, 4, 1), (0, 18, 1, 2), (0, 5, 5, 1)])
def test_poly_resize_coset_extension(ctx, curve, log_n, expand, batch):
    """zkhip_poly_resize_dev growing n -> K n (round 5): the n known values copied to their places + the K - 1 new cosets by n-point
    transforms that store straight into theirs (K <= 16; K = 32 takes the old path) -- against the oracle's inverse transform, zero
    padding and K n-point transform; single- and multi-pass sizes, lone polynomials and batches, odd numbers of cosets x polynomials
    (the unpaired kernel), both curves; the option poly_coset_extend = 0 (one big transform) must give the same bits; d_in holds the
    coefficients afterwards either way."""
    import ctypes

    C = CURVES[curve]
    n = 1 << log_n
    log_out = log_n + expand
    m = 1 << log_out
    wn, wm = limbs(C.root_of_unity(log_n), 4), limbs(C.root_of_unity(log_out), 4)
    evals = cp.random_fr(curve, 770 + log_n, batch * n).reshape(batch, n, 4)
    coeffs = cp.ntt(curve, evals, log_n, wn, inverse=True)
    padded = np.zeros((batch, m, 4), dtype=np.uint64)
    padded[:, :n] = coeffs
    exp = cp.ntt(curve, padded, log_out, wm)
    d_in, d_out = ctx.malloc(evals.nbytes), ctx.malloc(exp.nbytes)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for mode in (1, 0):
        ctx.set_option("poly_coset_extend", mode)
        ctx.h2d(d_in, evals)
        ctx.h2d(d_out, np.full_like(exp, 7))
        rc = ctx.lib.zkhip_poly_resize_dev(ctx.h, curve, ctypes.c_void_p(d_in), ctypes.c_size_t(log_n), ctypes.c_size_t(batch), P(wn),
                                           ctypes.c_void_p(d_out), ctypes.c_size_t(log_out), P(wm))
        assert rc == 0
        got, back = np.zeros_like(exp), np.zeros_like(evals)
        ctx.d2h(got, d_out)
        ctx.d2h(back, d_in)
        assert (got == exp).all(), mode
        assert (back == coeffs).all(), mode
    ctx.set_option("poly_coset_extend", 1)
    # roots that are NOT nested (omega_out^K != omega_n: primitive roots of the right orders, from different generators) are legal
    # arguments of this entry point; the coset path would place the known values on the wrong points -- it must step aside for them
    if expand >= 2:
        wm3 = limbs(pow(C.root_of_unity(log_out), 3, C.r), 4)  # another primitive 2^log_out-th root; its K-th power is omega_n^3
        exp3 = cp.ntt(curve, padded, log_out, wm3)
        ctx.h2d(d_in, evals)
        rc = ctx.lib.zkhip_poly_resize_dev(ctx.h, curve, ctypes.c_void_p(d_in), ctypes.c_size_t(log_n), ctypes.c_size_t(batch), P(wn),
                                           ctypes.c_void_p(d_out), ctypes.c_size_t(log_out), P(wm3))
        assert rc == 0
        got = np.zeros_like(exp3)
        ctx.d2h(got, d_out)
        assert (got == exp3).all()
    ctx.free(d_in)
    ctx.free(d_out)


def test_ntt_rejects_a_root_of_the_wrong_order(zk, ctx):
    """omega must be a PRIMITIVE 2^log_m-th root of unity (a key generated over another domain must fail loudly,
    not transform over the wrong domain): the square of the right root (order m / 2), 1, and the root of the next size are refused."""
    C = po.BLS12_381
    log_m = 8
    a = cp.random_fr(0, 3, 1 << log_m).reshape(1, -1, 4)
    w = C.root_of_unity(log_m)
    assert (ctx.ntt(0, a, log_m, limbs(w, 4)) == cp.ntt(0, a, log_m, limbs(w, 4))).all()
    for bad in (w * w % C.r, 1, C.root_of_unity(log_m + 1), C.r - 1):
        with pytest.raises(zk.ZkhipError):
            ctx.ntt(0, a, log_m, limbs(bad, 4))


@pytest.mark.parametrize("curve", [0, 1])
def test_domain_fft_all_kinds(ctx, zk, curve):
    """evaluation_domain::fft / inverse_fft over the domains make_evaluation_domain returns for sizes that are no power of
    two (step radix-2; extended radix-2 over a pretended small two-adicity), against the oracle's domains -- which are pinned
    to the definitions over each point set (tests/test_oracle_kat.py) -- and, at the small sizes, against the definition
    directly; with and without the coset, batches, both directions, round trips."""
    C = CURVES[curve]
    r = C.r
    g = C.fr_generator
    gl = limbs(g, 4)
    # (min_size, pretended two-adicity): step domains with 1, 4, 16, 256, 512, 1024 columns and compr 2 .. 2^11; extended at 2^(s+1)
    cases = [(3, None), (5, None), (12, None), (20, None), (48, None), (65, None), (1040, None), (2048 + 256, None), (4096 + 512, None),
             (4096 + 2048, None), (8192 + 1024, None), (32768 + 16, None), (8, 2), (64, 5), (2048, 10), (32, None)]
    for n, s in cases:
        dom, zd = qap_domains(zk, curve, n, two_adicity=s)
        m = dom.m
        batch = 3 if m < 4096 else 2
        a = cp.random_fr(curve, 77 + n, batch * m).reshape(batch, m, 4)
        sh = limbs(dom.shift, 4)
        exp = np.stack([cp.domain_fft(curve, dom.kind, a[b], limbs(dom.omega, 4), sh) for b in range(batch)])
        got = ctx.domain_fft(curve, zd, a)
        assert (got == exp).all(), (curve, dom.describe())
        if m <= 64:
            xs = dom.elements()
            assert fr_ints(got[0]) == [po.poly_eval(fr_ints(a[0]), x, r) for x in xs]
        assert (ctx.domain_fft(curve, zd, got, inverse=True) == a).all()
        # coset: multiply_by_coset(g) then fft; inverse_fft then multiply_by_coset(g^-1)
        ac = np.stack([fr_arr(po.multiply_by_coset(fr_ints(a[b]), g, r)) for b in range(batch)]) if m <= 4096 else None
        gotc = ctx.domain_fft(curve, zd, a, coset=gl)
        if ac is not None:
            expc = np.stack([cp.domain_fft(curve, dom.kind, ac[b], limbs(dom.omega, 4), sh) for b in range(batch)])
            assert (gotc == expc).all(), (curve, dom.describe(), "coset")
        assert (ctx.domain_fft(curve, zd, gotc, inverse=True, coset=gl) == a).all()
    # a root of the wrong order is refused
    dom, zd = qap_domains(zk, curve, 20)
    bad = zk.zkhip.Domain.make(dom.kind, dom.m, limbs(C.root_of_unity(4), 4))
    with pytest.raises(zk.zkhip.ZkhipError):
        ctx.domain_fft(curve, bad, cp.random_fr(curve, 1, 20).reshape(1, 20, 4))


@pytest.mark.parametrize("curve", [0, 1])
def test_domain_lagrange_all_kinds(ctx, zk, curve):
    """evaluation_domain::evaluate_all_lagrange_polynomials(t) on the device (zkhip_domain_lagrange_dev: the key generator's QAP
    evaluation, r1cs_to_qap.hpp:152-153) against the oracle's domains -- pinned to the product definition of the Lagrange
    polynomials over each point set (tests/test_oracle_kat.py) --: basic, step (1 ... 1024 columns, every chunk boundary of the
    16-point batched inversion) and extended radix-2 (over a pretended small two-adicity); sum_i L_i(t) = 1; t on the domain refused."""
    C = CURVES[curve]
    r = C.r
    rng = po.SplitMix64(91 + curve)
    for n, s in [(2, None), (3, None), (5, None), (16, None), (17, None), (20, None), (33, None), (48, None), (65, None), (1040, None), (2048 + 256, None),
                 (4096 + 2048, None), (32768 + 16, None), (8, 2), (64, 5), (2048, 10), (1 << 13, None)]:
        dom, zd = qap_domains(zk, curve, n, two_adicity=s)
        t = rng.next_mod(r)
        got = fr_ints(ctx.domain_lagrange(curve, zd, limbs(t, 4)))
        assert got == dom.evaluate_all_lagrange_polynomials(t), (curve, dom.describe())
        assert sum(got) % r == 1
    dom, zd = qap_domains(zk, curve, 20)
    with pytest.raises(zk.zkhip.ZkhipError):
        ctx.domain_lagrange(curve, zd, limbs(dom.elements()[7], 4))
