"""Parity of the HIP Pippenger path (through the C ABI) against the oracle, the reference's own
known-answer vectors, and size-independent properties at the full BASELINE size.  Bit-exact (integer)."""
import json
import os

import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, fr_arr, fr_ints, jac_to_affine_py, limbs, pt_from_limbs, pt_limbs, pts_arr, qap_domains

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
h = lambda s: int(s, 16)


def gpu_affine(ctx, bases, scalars, **kw):
    """MSM on the GPU -> affine limbs via the checker's big-int inversion (and the device conversion)."""
    jac = ctx.msm(bases, scalars, **kw)
    P = jac_to_affine_py(bases.curve, bases.group, jac)
    dev_aff, dev_inf = ctx.jacobian_to_affine(bases.curve, bases.group, jac)
    assert pt_from_limbs(bases.curve, bases.group, dev_aff, dev_inf) == P
    return P


def test_reference_kat_vectors(ctx):
    """AGG:864-930 (bellperson): 2 x G2 MSM (n=8), 2 x G1 MSM (n=16); kzg.cpp:75-103 commit identity."""
    q = json.load(open(os.path.join(HERE, "golden", "ref_kat.json")))["prove_commitment_test"]
    C = po.BLS12_381
    tr = [h(x) for x in q["tr"]]
    qv, ha, hb = po.ipp2_prove_commitment_v(C, q["n"], h(q["alpha"]), h(q["beta"]), tr, h(q["kzg_challenge"]))
    qw, ga, gb = po.ipp2_prove_commitment_w(C, q["n"], h(q["alpha"]), h(q["beta"]), tr, h(q["r_shift"]), h(q["kzg_challenge"]))
    g2 = lambda e: ((h(e[0][0]), h(e[0][1])), (h(e[1][0]), h(e[1][1])))
    g1 = lambda e: (h(e[0]), h(e[1]))
    for group, pts, sc, exp in ((2, ha, qv, g2(q["comm_v"][0])), (2, hb, qv, g2(q["comm_v"][1])),
                                (1, ga, qw, g1(q["comm_w"][0])), (1, gb, qw, g1(q["comm_w"][1]))):
        b = ctx.upload_bases(0, group, pts_arr(0, group, pts))
        assert gpu_affine(ctx, b, fr_arr(sc)) == exp
        b.free()
    for curve in (0, 1):
        Cc = CURVES[curve]
        ck = po.structured_generators(Cc.g1, 4, 10)
        b = ctx.upload_bases(curve, 1, pts_arr(curve, 1, ck))
        assert gpu_affine(ctx, b, fr_arr([Cc.r - 1, 1, 2, 3])) == Cc.g1.mul(Cc.g1.gen, 3209)
        b.free()


@pytest.mark.parametrize("curve,group,sizes", [(0, 1, (1, 2, 17, 256, 4096)), (1, 1, (1, 17, 300, 2048)),
                                               (0, 2, (1, 2, 17, 200)), (1, 2, (1, 17, 200))])
def test_msm_matches_oracle(ctx, curve, group, sizes):
    C = CURVES[curve]
    nmax = max(sizes)
    ks = cp.random_fr(curve, 100 + curve * 10 + group, nmax)
    pts, inf = cp.batch_mul(curve, group, ks)
    bases = ctx.upload_bases(curve, group, pts)
    dl, dinf = bases.download()
    assert (dl == pts).all() and not dinf.any()
    for n in sizes:
        sc = cp.random_fr(curve, 200 + n, n)
        exp, einf = cp.msm(curve, group, pts[:n], sc, chunks=4)
        got = gpu_affine(ctx, bases, sc, n=n)
        assert got == pt_from_limbs(curve, group, exp, einf), (curve, group, n)
    bases.free()


@pytest.mark.parametrize("curve,group", [(0, 1), (1, 1), (0, 2)])
def test_msm_edge_cases(ctx, curve, group):
    """scalars {0, 1, r-1}, duplicate / negated / infinity bases, all-zero scalars, empty input, offsets."""
    C = CURVES[curve]
    G = C.g1 if group == 1 else C.g2
    n = 64
    ks = cp.random_fr(curve, 5, n)
    pts, _ = cp.batch_mul(curve, group, ks)
    P = [pt_from_limbs(curve, group, pts[i]) for i in range(n)]
    P[1] = P[0]                 # duplicate
    P[2] = G.neg(P[0])          # negation of a base that is also present
    P[3] = None                 # infinity base
    P[10] = P[11]
    arr = pts_arr(curve, group, P)
    infs = np.array([1 if p is None else 0 for p in P], dtype=np.uint8)
    bases = ctx.upload_bases(curve, group, arr, infs)
    sc = fr_ints(cp.random_fr(curve, 6, n))
    sc[0] = sc[1] = sc[2] = 12345  # P0*k + P0*k - P0*k
    sc[4] = 0
    sc[5] = 1
    sc[6] = C.r - 1
    sc[7] = 1
    sc[10] = 77
    sc[11] = C.r - 77            # cancels with the duplicate base
    exp = po.msm_naive(G, P, sc)
    assert gpu_affine(ctx, bases, fr_arr(sc)) == exp
    # every window size, with window tables (rebuilt for each c) ...
    for c in (2, 3, 5, 7, 12, 15, 16, 17, 20, 21):  # 3, 5, 15, 17 divide 255: the top window is full
        ctx.set_option("msm_window_bits", c)
        tb = ctx.upload_bases(curve, group, arr, infs)
        # all windows in one bucket set, a few sets folded afterwards, one set per window; both sort tile shapes
        for sets, tile_log in ((0, 14), (1, 14), (2, 12), (5, 14), (64, 12)):
            ctx.set_option("msm_sets", sets)
            ctx.set_option("msm_sort_tile_log", tile_log)
            assert gpu_affine(ctx, tb, fr_arr(sc)) == exp, (c, sets, tile_log)
        ctx.set_option("msm_sets", 0)
        ctx.set_option("msm_sort_tile_log", 14)
        tb.free()
    # ... and without them (the per-window sums are combined by the Horner pass of msm_final)
    ctx.set_option("msm_precompute", 0)
    ctx.set_option("msm_window_bits", 0)
    bases.free()
    bases = ctx.upload_bases(curve, group, arr, infs)
    ctx.set_option("msm_precompute", 1)
    for c in (0, 2, 3, 5, 7, 12, 15, 16, 19):
        ctx.set_option("msm_window_bits", c)
        assert gpu_affine(ctx, bases, fr_arr(sc)) == exp, c
    # scalars that are not canonical are taken mod r (the reference's field type cannot hold them)
    nc = list(sc)
    nc[8] = sc[8] + C.r if sc[8] + C.r < (1 << 256) else sc[8]
    nc[9] = (1 << 256) - 1
    sc9 = list(sc)
    sc9[9] = ((1 << 256) - 1) % C.r
    for c in (0, 15):
        ctx.set_option("msm_window_bits", c)
        assert gpu_affine(ctx, bases, fr_arr(nc)) == po.msm_naive(G, P, sc9)
    ctx.set_option("msm_window_bits", 0)
    zero = np.zeros((n, 4), dtype=np.uint64)
    assert gpu_affine(ctx, bases, zero) is None
    assert gpu_affine(ctx, bases, zero[:0], n=0) is None
    one = zero.copy()
    one[:, 0] = 1
    assert gpu_affine(ctx, bases, one) == po.msm_naive(G, P, [1] * n)
    # offset / sub-range (the reference passes iterator sub-ranges: prover.hpp:133-139)
    assert gpu_affine(ctx, bases, fr_arr(sc[20:50]), offset=20, n=30) == po.msm_naive(G, P[20:50], sc[20:50])
    with pytest.raises(Exception):
        ctx.msm(bases, fr_arr(sc), offset=10, n=n)
    bases.free()


@pytest.mark.parametrize("curve,group,n", [(0, 1, 20000), (1, 1, 9000), (0, 2, 9000)])
def test_msm_skewed_scalars(ctx, curve, group, n):
    """most scalars identical: every window has one bucket with thousands of entries, which is cut into several
    tasks (msm_bucket_large) whose partial sums are folded by msm_large_combine."""
    ks = cp.random_fr(curve, 21, n)
    bases = ctx.bases_from_scalars(curve, group, ks)
    pts, infs = bases.download()
    sc = cp.random_fr(curve, 22, n)
    sc[: (3 * n) // 4] = sc[0]
    exp, einf = cp.msm(curve, group, pts, sc, chunks=cp.num_threads())
    assert gpu_affine(ctx, bases, sc) == pt_from_limbs(curve, group, exp, einf)
    bases.free()


@pytest.mark.parametrize("curve,group,n", [(0, 2, 300), (1, 2, 1500), (0, 1, 1500)])
def test_msm_over_device_generated_bases(ctx, curve, group, n):
    """bases built on the device (zkhip_bases_from_scalars, and the window tables above 1024 points) hold lazily reduced
    coordinates -- for G2 up to 10p -- unlike uploaded ones (< 2p): the bucket kernels must accept both."""
    b = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, 21, n))
    pts, infs = b.download()
    sc = cp.random_fr(curve, 22, n)
    exp, einf = cp.msm(curve, group, pts, sc, chunks=cp.num_threads())
    for c in (0, 2, 6, 10):
        ctx.set_option("msm_window_bits", c)
        assert gpu_affine(ctx, b, sc) == pt_from_limbs(curve, group, exp, einf), c
    ctx.set_option("msm_window_bits", 0)
    b.free()


@pytest.mark.parametrize("curve,group,n", [(0, 1, 500), (1, 2, 60)])
def test_bases_from_scalars(ctx, curve, group, n):
    """device-side fixed-base batch exponentiation (generator.hpp:187-214) against the oracle"""
    ks = cp.random_fr(curve, 42, n)
    ks[0] = 0
    ks[1] = [1, 0, 0, 0]
    exp, einf = cp.batch_mul(curve, group, ks)
    b = ctx.bases_from_scalars(curve, group, ks)
    got, ginf = b.download()
    assert (ginf == einf).all() and (got == exp).all()
    b.free()


def test_msm_full_size_bls12_381_g1(ctx):
    """BASELINE config 2: 2^20 points, bit-exact against the oracle, plus linearity at full size."""
    n = 1 << 20
    ks = cp.random_fr(0, 1, n)
    bases = ctx.bases_from_scalars(0, 1, ks)
    # spot-check the device-generated bases against the oracle
    sample = np.array([0, 1, 77777, n - 1])
    exp_pts, _ = cp.batch_mul(0, 1, ks[sample])
    for j, i in enumerate(sample):
        got, _ = bases.download(int(i), 1)
        assert (got[0] == exp_pts[j]).all()
    s1 = cp.random_fr(0, 2, n)
    s2 = cp.random_fr(0, 3, n)
    G = po.BLS12_381.g1
    r1 = gpu_affine(ctx, bases, s1)
    # oracle on the same inputs (all host cores)
    allpts, _ = bases.download()
    exp, einf = cp.msm(0, 1, allpts, s1, chunks=cp.num_threads())
    assert r1 == pt_from_limbs(0, 1, exp, einf)
    # linearity: MSM(s1) + MSM(s2) == MSM(s1 + s2 mod r)
    r2 = gpu_affine(ctx, bases, s2)
    r = po.BLS12_381.r
    ssum = fr_arr([(a + b) % r for a, b in zip(fr_ints(s1), fr_ints(s2))])
    assert gpu_affine(ctx, bases, ssum) == G.add(r1, r2)
    # Groth16-like scalar distribution: half of the scalars in {0, 1}
    s3 = s1.copy()
    s3[::4] = 0
    s3[1::4] = [1, 0, 0, 0]
    exp, einf = cp.msm(0, 1, allpts, s3, chunks=cp.num_threads())
    assert gpu_affine(ctx, bases, s3) == pt_from_limbs(0, 1, exp, einf)
    bases.free()


@pytest.mark.parametrize("curve,group,n", [(0, 1, 3000), (0, 2, 700)])
def test_msm_repeated_calls_replay_graph(ctx, zk, curve, group, n):
    """the third identical call (same bases, same device buffers) is captured into a HIP graph and later ones replay it:
    the CONTENT of the scalar buffer changes between calls and every result must follow it; same through the batch
    entry point; growing the workspace in between (a larger MSM) must invalidate the graphs, not corrupt them."""
    ctx.set_option("msm_graphs", 1)  # off by default (no measured gain): exercised here
    b = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, 60, n))
    b2 = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, 61, n))
    pts, _ = b.download()
    pts2, _ = b2.download()
    jac = 3 * zk.coord_limbs(curve, group) * 8
    d_s, d_s2, d_o = ctx.malloc(n * 32), ctx.malloc(n * 32), ctx.malloc(2 * jac)
    res = np.zeros((2, jac // 8), dtype=np.uint64)

    def check(k, batch):
        s1, s2 = cp.random_fr(curve, 600 + k, n), cp.random_fr(curve, 700 + k, n)
        ctx.h2d(d_s, s1)
        ctx.h2d(d_s2, s2)
        if batch:
            ctx.msm_batch_dev([b, b2], [d_s, d_s2], [d_o, d_o + jac])
        else:
            ctx.msm_dev(b, d_s, d_o)
            ctx.msm_dev(b2, d_s2, d_o + jac)
        ctx.d2h(res, d_o)
        for r, p, s in ((res[0], pts, s1), (res[1], pts2, s2)):
            exp, einf = cp.msm(curve, group, p, s, chunks=4)
            assert jac_to_affine_py(curve, group, r) == pt_from_limbs(curve, group, exp, einf), (k, batch)

    for k in range(6):
        check(k, False)
    for k in range(6):
        check(10 + k, True)
    big = ctx.bases_from_scalars(curve, 1, cp.random_fr(curve, 62, 1 << 17))  # forces a larger workspace
    ctx.msm(big, cp.random_fr(curve, 63, 1 << 17))
    big.free()
    for k in range(4):
        check(20 + k, False)
        check(30 + k, True)
    ctx.set_option("msm_graphs", 0)
    check(40, False)
    for p in (d_s, d_s2, d_o):
        ctx.free(p)
    b.free()
    b2.free()


@pytest.mark.parametrize("curve,group,log_n", [(1, 1, 20), (0, 2, 18), (0, 1, 22)])
def test_msm_large_other_configs(ctx, zk, curve, group, log_n):
    """full-size MSMs outside the headline configuration (BN254 G1 2^20, BLS12-381 G2 2^18 on the lane-pair kernels,
    BLS12-381 G1 2^22): bit-exact against the oracle where it finishes in seconds, and the size-independent property
    MSM(s1) + MSM(s2) == MSM(s1 + s2 mod r) computed entirely on the device."""
    n = 1 << log_n
    b = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, 11, n))
    s1, s2 = cp.random_fr(curve, 12, n), cp.random_fr(curve, 13, n)
    jac = 3 * zk.coord_limbs(curve, group) * 8
    d1, d2, d3, d_out = ctx.malloc(n * 32), ctx.malloc(n * 32), ctx.malloc(n * 32), ctx.malloc(3 * jac)
    ctx.h2d(d1, s1)
    ctx.h2d(d2, s2)
    ctx.fr_vec_op_dev(curve, 0, d1, d2, d3, n)
    for i, d in enumerate((d1, d2, d3)):
        ctx.msm_dev(b, d, d_out + i * jac)
    res = np.zeros((3, jac // 8), dtype=np.uint64)
    ctx.d2h(res, d_out)
    ctx.jacobian_sum_dev(curve, group, d_out, 2, d_out)
    tot = np.zeros(jac // 8, dtype=np.uint64)
    ctx.d2h(tot, d_out)
    a_sum, i_sum = ctx.jacobian_to_affine(curve, group, tot)
    a_3, i_3 = ctx.jacobian_to_affine(curve, group, res[2])
    assert i_sum == i_3 and (a_sum == a_3).all()
    if log_n <= 20:
        pts, _ = b.download()
        exp, einf = cp.msm(curve, group, pts, s1, chunks=cp.num_threads())
        a_1, i_1 = ctx.jacobian_to_affine(curve, group, res[0])
        assert i_1 == einf and (a_1 == exp).all()
    for d in (d1, d2, d3, d_out):
        ctx.free(d)
    b.free()


@pytest.mark.parametrize("curve,group,log_n", [(0, 1, 15), (1, 1, 13), (0, 2, 13)])
def test_batch_members_over_the_same_scalars_share_one_sort(ctx, zk, curve, group, log_n):
    """Queries laid out over the same rows (zkhip_bases_spread) and multiplied by the same vector -- A, the dense B.h, the padded L of
    one proof -- share the digit extraction / sort / large-bucket plan of a batch ("msm_share_sort"): the results equal those of the
    original objects over their own (gathered / offset) scalars, with the option on and off."""
    n = 1 << log_n
    rng = np.random.default_rng(5)
    a = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, 21, n))
    rows = np.sort(rng.choice(n, size=n - n // 16, replace=False)).astype(np.uint32)    # a sparse query: its index list
    sparse = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, 22, len(rows)))
    first = 11
    tail = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, 23, n - first))     # the auxiliary part
    d_rows = ctx.malloc(rows.nbytes)
    ctx.h2d(d_rows, rows)
    sparse_rows, tail_rows = sparse.spread(n, d_rows=d_rows), tail.spread(n, first=first)
    # the spread objects hold the same points at the named rows and infinity elsewhere
    pts, inf = sparse_rows.download()
    src, _ = sparse.download()
    assert (pts[rows] == src).all() and not inf[rows].any() and inf.sum() == n - len(rows)
    pts, inf = tail_rows.download()
    src, _ = tail.download()
    assert (pts[first:] == src).all() and inf[:first].all() and not inf[first:].any()
    sc = cp.random_fr(curve, 24, n)
    sc[: n // 4, :] = 0
    sc[: n // 8, 0] = 1    # Groth16-like: zeros and ones (the large-bucket plan is shared, too)
    jac = 3 * zk.coord_limbs(curve, group) * 8
    d_s, d_g, d_out = ctx.malloc(n * 32), ctx.malloc(len(rows) * 32), ctx.malloc(6 * jac)
    ctx.h2d(d_s, sc)
    ctx.h2d(d_g, np.ascontiguousarray(sc[rows]))
    # the reference arrangement: every object over its own scalars
    ctx.msm_dev(a, d_s, d_out)
    ctx.msm_dev(sparse, d_g, d_out + jac)
    ctx.msm_dev(tail, d_s + 32 * first, d_out + 2 * jac)
    want = np.zeros((3, jac // 8), dtype=np.uint64)
    ctx.d2h(want, d_out)
    want = [ctx.jacobian_to_affine(curve, group, w) for w in want]
    for share in (1, 0):
        ctx.set_option("msm_share_sort", share)
        ctx.msm_batch_dev([a, sparse_rows, tail_rows], [d_s, d_s, d_s], [d_out + 3 * jac, d_out + 4 * jac, d_out + 5 * jac])
        got = np.zeros((6, jac // 8), dtype=np.uint64)
        ctx.d2h(got, d_out)
        for k in range(3):
            g_aff, g_inf = ctx.jacobian_to_affine(curve, group, got[3 + k])
            assert g_inf == want[k][1] and (g_aff == want[k][0]).all(), (share, k)
    ctx.set_option("msm_share_sort", 1)
    if log_n <= 13:    # and against the oracle
        p, _ = a.download()
        exp, einf = cp.msm(curve, group, p, sc, chunks=2)
        assert want[0][1] == einf and (want[0][0] == exp).all()
    for d in (d_rows, d_s, d_g, d_out):
        ctx.free(d)
    for b in (a, sparse, tail, sparse_rows, tail_rows):
        b.free()


# M + n + 1: 2^4+3 -> step(20); 2^10+11 -> step(2^10+16); 54 -> basic(64); 4096 -> basic; 65 -> step(64 + 1);
# 2^11+256 -> step, 256 columns; 2^12+1000 -> step(2^12 + 2^10): column tiles; 2^15+16 -> step
@pytest.mark.parametrize("curve,M,n", [(0, 16, 2), (1, 1024, 10), (0, 1024, 10), (0, 50, 3), (0, 4085, 10), (1, 60, 4), (0, 2048 + 250, 5),
                                       (0, 4096 + 990, 9), (1, 32768, 15), (0, 32768, 15)])
def test_groth16_witness_map(ctx, zk, curve, M, n):
    """r1cs_to_qap::witness_map on the device (sparse mat-vec, 7 transforms, pointwise) against the oracle, over the domain
    make_evaluation_domain(M + n + 1) picks -- a step radix-2 domain for most sizes -- and over the basic domain of the next
    power of two; the example system has two rows of ~M terms (long-row path) and M short rows."""
    C = CURVES[curve]
    g16 = cp.Groth16(curve, M, n, seed=3)
    assert g16.is_satisfied()
    gen = limbs(C.fr_generator, 4)
    r1cs = ctx.upload_r1cs(curve, g16.M, g16.n, g16.N, g16.csr(0), g16.csr(1), g16.csr(2))
    z = np.concatenate([np.array([[1, 0, 0, 0]], dtype=np.uint64), g16.assignment()])
    dom, zd = qap_domains(zk, curve, M + n + 1)
    assert (r1cs.kind, r1cs.m) == (dom.kind, dom.m) == zk.zkhip.domain_choice(curve, M + n + 1)
    # the basic domain of the next power of two first (the oracle's default), then the reference's choice
    m2 = 1 << (M + n).bit_length()
    for kind, m, w in ((0, m2, limbs(C.root_of_unity(m2.bit_length() - 1), 4)), (dom.kind, dom.m, limbs(dom.omega, 4))):
        r1cs.set_domain(kind, m)
        if kind != 0:
            g16.set_domain(kind, m, w)
        assert r1cs.m == g16.m == m
        exp = g16.witness_map(w, gen)
        got = ctx.groth16_witness_h(r1cs, z, w, gen)
        assert (got == exp).all()
        assert not got[m - 1].any() and not got[m].any()  # prover.hpp:88-89
        if kind != 0:    # the same through the whole domain description
            assert (ctx.groth16_witness_h(r1cs, z, zd, gen) == exp).all()
    if M <= 4096:
        # an EXTENDED radix-2 domain of the next power of two (a real field reaches that kind only at 2^(s+1) points; the kind is an
        # argument): <omega> and shift <omega>, omega of order m2 / 2, shift = multiplicative_generator^2 (detail::coset_shift)
        sh = limbs(pow(C.fr_generator, 2, C.r), 4)
        we = limbs(C.root_of_unity(m2.bit_length() - 2), 4)
        r1cs.set_domain(1, m2)
        g16.set_domain(1, m2, we, sh)
        exp = g16.witness_map(we, gen)
        got = ctx.groth16_witness_h(r1cs, z, zk.zkhip.Domain.make(1, m2, we, sh), gen)
        assert (got == exp).all() and not got[m2 - 1].any()
        with pytest.raises(zk.zkhip.ZkhipError):    # the extended kind needs its shift: the omega-only entry point refuses it
            ctx.groth16_witness_h(r1cs, z, we, gen)
    r1cs.free()


def test_tail_group_law_over_lane_pairs_and_quads(ctx):
    """the MSM tail's group law spread over lane pairs (fu_pair.hpp) and lane quads (fu_quad.hpp) against the one-lane formulas:
    addition, doubling, small multiples, both base fields (tests/cpp/quadtest.hip, a device-only unit test)"""
    import subprocess

    exe = os.path.join(os.path.dirname(HERE), "tests", "cpp", "quadtest")
    assert os.path.exists(exe), "tests/cpp/quadtest is missing: python -c 'import __graft_entry__ as g; g.build()'"
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, text=True)
    assert out.returncode == 0 and out.stdout.count("mismatch mask 0x0 ") == 2, out.stdout


def test_msm_batch(ctx, zk):
    """zkhip_msm_batch_dev: several MSMs with one shared bucket reduction == the same MSMs one by one;
    covers an empty member, a sub-range member and the mixed (fallback) case."""
    n = 3000
    ks = cp.random_fr(0, 31, n)
    b1 = ctx.bases_from_scalars(0, 1, ks)
    b2 = ctx.bases_from_scalars(0, 1, cp.random_fr(0, 32, n))
    ctx.set_option("msm_precompute", 0)
    small = ctx.bases_from_scalars(0, 1, cp.random_fr(0, 33, 40))  # no window tables: forces the fallback when mixed in
    ctx.set_option("msm_precompute", 1)
    scs = [cp.random_fr(0, 40 + i, n) for i in range(3)]
    scs[2][::3] = 0
    d_s = [ctx.malloc(s.nbytes) for s in scs]
    for d, s in zip(d_s, scs):
        ctx.h2d(d, s)
    d_o = [ctx.malloc(144) for _ in range(4)]
    singles = [jac_to_affine_py(0, 1, ctx.msm(b, s)) for b, s in ((b1, scs[0]), (b2, scs[1]), (b1, scs[2]))]
    sub = jac_to_affine_py(0, 1, ctx.msm(b2, scs[0][:1000], offset=500, n=1000))

    def fetch(k):
        out = np.zeros((3, 6), dtype=np.uint64)
        ctx.d2h(out, d_o[k])
        return jac_to_affine_py(0, 1, out)

    ctx.msm_batch_dev([b1, b2, b1, b2], [d_s[0], d_s[1], d_s[2], d_s[0]], d_o, offsets=[0, 0, 0, 500], ns=[n, n, n, 1000])
    assert [fetch(k) for k in range(4)] == singles + [sub]
    ctx.msm_batch_dev([b1, b2, b1], [d_s[0], d_s[1], d_s[2]], d_o[:3], ns=[n, 0, n])
    assert fetch(0) == singles[0] and fetch(1) is None and fetch(2) == singles[2]
    ctx.msm_batch_dev([b1, small], [d_s[0], d_s[1]], d_o[:2], ns=[n, 40])  # mixed: sequential fallback
    assert fetch(0) == singles[0] and fetch(1) == jac_to_affine_py(0, 1, ctx.msm(small, scs[1][:40]))
    for d in d_s + d_o:
        ctx.free(d)


@pytest.mark.parametrize("curve,group,n", [(0, 1, 5000), (0, 1, 20), (1, 1, 700), (0, 2, 300), (1, 2, 150)])
@pytest.mark.parametrize("world", [2, 4, 8, 19])
def test_msm_partitions_over_ranks(zk, ctx, curve, group, n, world):
    """SURVEY 8e's two ways to spread one MSM over `world` GPUs, emulated rank by rank on this GPU and folded with
    zkhip_jacobian_sum_dev exactly as the all-gather's receiver does:
      (i)  point-range partition: rank g runs the whole pipeline over points [lo_g, hi_g);
      (ii) window partition: rank g holds the tables of windows {w : w mod world == g} for ALL points
           (options msm_shard_rank / msm_shard_world at upload) and sums those windows only.
    Both must give the unsharded result, bit for bit in affine.  world = 19 exceeds the 16-17 windows: idle ranks."""
    from crypto3_zk_amd import dist as zd

    ks = cp.random_fr(curve, 900 + n, n)
    pts, _ = cp.batch_mul(curve, group, ks)
    sc = cp.random_fr(curve, 901 + n, n)
    sc[::7] = 0
    sc[1::11] = fr_arr([1])[0]
    sc[2::13] = fr_arr([CURVES[curve].r - 1])[0]
    exp, einf = cp.msm(curve, group, pts, sc, chunks=4)
    cl = zk.coord_limbs(curve, group)
    d_parts = ctx.malloc(world * 3 * cl * 8)
    d_tot = ctx.malloc(3 * cl * 8)
    d_sc = ctx.malloc(sc.nbytes)
    ctx.h2d(d_sc, sc)

    def folded():
        ctx.jacobian_sum_dev(curve, group, d_parts, world, d_tot)
        jac = np.zeros((3, cl), dtype=np.uint64)
        ctx.d2h(jac, d_tot)
        return ctx.jacobian_to_affine(curve, group, jac)

    try:
        # (i) point ranges
        full = ctx.upload_bases(curve, group, pts)
        for rank in range(world):
            lo, hi = zd.shard_range(n, rank, world)
            ctx.msm_dev(full, d_sc + 32 * lo, d_parts + rank * 3 * cl * 8, lo, hi - lo)
        aff, inf = folded()
        assert inf == einf and (aff == exp).all()
        full.free()
        # (ii) windows
        for rank in range(world):
            ctx.set_option("msm_shard_world", world)
            ctx.set_option("msm_shard_rank", rank)
            part = ctx.upload_bases(curve, group, pts)
            ctx.set_option("msm_shard_world", 1)
            dl, _ = part.download()
            assert (dl == pts).all()
            ctx.msm_dev(part, d_sc, d_parts + rank * 3 * cl * 8, 0, n)
            ctx.sync()
            part.free()
        aff, inf = folded()
        assert inf == einf and (aff == exp).all()
    finally:
        ctx.set_option("msm_shard_world", 1)
        for p in (d_parts, d_tot, d_sc):
            ctx.free(p)


def test_gather_bounds_and_device_status(zk, ctx):
    """zkhip_fr_gather_dev bounds every index by the source length: an index beyond it gathers the zero scalar and
    raises the sticky device status, which zkhip_device_status reports once and clears."""
    src = cp.random_fr(0, 7, 10)
    idx = np.array([0, 9, 3, 10, 4000000000, 2], dtype=np.uint32)
    d_src, d_idx, d_dst = ctx.malloc(src.nbytes), ctx.malloc(idx.nbytes), ctx.malloc(len(idx) * 32)
    ctx.h2d(d_src, src)
    ctx.h2d(d_idx, idx)
    assert ctx.device_status() == 0
    ctx.fr_gather_dev(d_src, 10, d_idx, len(idx), d_dst)
    out = np.zeros((len(idx), 4), dtype=np.uint64)
    ctx.d2h(out, d_dst)
    assert (out[[0, 1, 2, 5]] == src[[0, 9, 3, 2]]).all() and not out[[3, 4]].any()
    with pytest.raises(zk.ZkhipError):
        ctx.device_status()
    assert ctx.device_status() == 0  # cleared
    ctx.fr_gather_dev(d_src, 10, d_idx, 3, d_dst)  # in range: stays clean
    assert ctx.device_status() == 0
    for p in (d_src, d_idx, d_dst):
        ctx.free(p)


def test_msm_random_configurations(zk, ctx):
    """the same MSM under randomly drawn (window size, bucket sets, sort tile shape, tail segment, tables on / off, sub-range)
    combinations -- every path of the planner -- against the oracle."""
    rng = np.random.default_rng(2024)
    for curve, group, n in ((0, 1, 3000), (1, 1, 900), (0, 2, 500)):
        ks = cp.random_fr(curve, 300 + n, n)
        pts, _ = cp.batch_mul(curve, group, ks)
        sc = cp.random_fr(curve, 301 + n, n)
        sc[::9] = 0
        sc[1::17] = fr_arr([1])[0]
        try:
            for _ in range(10):
                c = int(rng.integers(2, 22))
                tables = bool(rng.integers(0, 4))
                ctx.set_option("msm_precompute", 1 if tables else 0)
                ctx.set_option("msm_window_bits", c)
                b = ctx.upload_bases(curve, group, pts)
                ctx.set_option("msm_sets", int(rng.integers(0, 9)))
                ctx.set_option("msm_sort_tile_log", int(rng.choice([12, 14])))
                ctx.set_option("msm_segment_log", int(rng.integers(-1, 6)))
                ctx.set_option("msm_tail_fold", int(rng.choice([0, 8, 12, 16])))  # two-level tail from 2^k buckets on (G1, tables)
                ctx.set_option("msm_fold_run", int(rng.choice([0, 1, 2, 4, 8, 16])))
                ctx.set_option("msm_tail_fold_g2", int(rng.integers(0, 2)))
                lo = int(rng.integers(0, n // 2))
                cnt = int(rng.integers(1, n - lo + 1))
                exp, einf = cp.msm(curve, group, pts[lo:lo + cnt], sc[lo:lo + cnt], chunks=4)
                got, ginf = ctx.msm_affine(b, sc[lo:lo + cnt], lo, cnt)
                assert ginf == einf and (got == exp).all(), (curve, group, c, tables, lo, cnt)
                b.free()
        finally:
            for name, v in (("msm_precompute", 1), ("msm_window_bits", 0), ("msm_sets", 0), ("msm_sort_tile_log", 14), ("msm_segment_log", -1),
                            ("msm_tail_fold", 16), ("msm_fold_run", 0), ("msm_tail_fold_g2", 1)):
                ctx.set_option(name, v)


@pytest.mark.parametrize("curve,group,n", [(0, 1, 1 << 17), (1, 1, 40000), (0, 2, 12000), (1, 2, 5000)])
def test_two_level_tail_equals_running_sums(zk, ctx, curve, group, n):
    """The bucket reduction in two levels (msm_core.hpp msm_fold: row and column sums of the bucket index, then the running sums over
    2 sets of ~sqrt(B) buckets) against the running sums over the whole set, for every bucket count 2^8 ... 2^20 (square and 2 : 1
    splits), every run length the kernel's geometry admits, a lone MSM and a batch with an empty and a sub-range member, G1 and G2;
    one case per group also against the oracle.  Bit-exact: both are sums of the same points."""
    ks = cp.random_fr(curve, 71, n)
    sc = cp.random_fr(curve, 72, n)
    sc[::7] = 0
    sc[3::11] = fr_arr([CURVES[curve].r - 1])[0]
    sc2 = cp.random_fr(curve, 73, n)
    d_s, d_s2 = ctx.malloc(sc.nbytes), ctx.malloc(sc2.nbytes)
    ctx.h2d(d_s, sc)
    ctx.h2d(d_s2, sc2)
    L = zk.coord_limbs(curve, group)
    d_o = [ctx.malloc(3 * L * 8) for _ in range(3)]

    def fetch(k):
        out = np.zeros((3, L), dtype=np.uint64)
        ctx.d2h(out, d_o[k])
        return jac_to_affine_py(curve, group, out)

    try:
        for c in ((9, 12, 13, 16, 17, 18, 19, 20, 21) if group == 1 else (9, 13, 16, 17, 20)):
            ctx.set_option("msm_window_bits", c)
            b = ctx.bases_from_scalars(curve, group, ks)
            results = []
            for fold, run in ((0, 0), (8, 0), (8, 1), (8, 2), (8, 4), (8, 8), (8, 16), (8, 32)):
                ctx.set_option("msm_tail_fold", fold)
                ctx.set_option("msm_fold_run", run)
                lone = jac_to_affine_py(curve, group, ctx.msm(b, sc))
                ctx.msm_batch_dev([b, b, b], [d_s, d_s2, d_s], d_o, offsets=[0, 0, 100], ns=[n, 0, n - 1000])
                results.append((lone, fetch(0), fetch(1), fetch(2)))
            assert all(r == results[0] for r in results[1:]), (curve, group, c)
            assert results[0][0] == results[0][1] and results[0][2] is None and results[0][0] is not None
            if c == 17 and curve == 1:
                exp, einf = cp.msm(curve, group, b.download()[0], sc, chunks=8)
                assert results[0][0] == pt_from_limbs(curve, group, exp, einf)
            b.free()
    finally:
        for name, v in (("msm_window_bits", 0), ("msm_tail_fold", 16), ("msm_fold_run", 0)):
            ctx.set_option(name, v)
        for d in [d_s, d_s2] + d_o:
            ctx.free(d)


@pytest.mark.parametrize("curve,group", [(0, 1), (1, 1), (0, 2)])
def test_two_level_tail_degenerate_buckets(zk, ctx, curve, group):
    """Row and column sums over buckets that hold THE SAME point (every base equal, scalar v once for v = 1 ... : each addition of the runs and of
    the trees is a doubling), its negative (scalars r - v: rows cancel to infinity), and mostly nothing (a handful of entries in 2^12 buckets):
    two levels == running sums == oracle."""
    r = CURVES[curve].r
    n, c = 6000, 13
    ks = fr_arr([5] * n)
    cases = {
        "doublings": fr_arr([(v % 4095) + 1 for v in range(n)]),
        "cancellations": fr_arr([(v % 2000) + 1 if v % 2 == 0 else r - ((v - 1) % 2000) - 1 for v in range(n)]),
        "sparse": fr_arr([0] * (n - 5) + [1, 4096, r - 4096, 77, 2 ** 40 + 3]),
    }
    try:
        ctx.set_option("msm_window_bits", c)
        b = ctx.bases_from_scalars(curve, group, ks)
        pts = b.download()[0]
        for name, sc in cases.items():
            got = []
            for fold, run in ((0, 0), (8, 0), (8, 2), (8, 8)):
                ctx.set_option("msm_tail_fold", fold)
                ctx.set_option("msm_fold_run", run)
                got.append(jac_to_affine_py(curve, group, ctx.msm(b, sc)))
            exp, einf = cp.msm(curve, group, pts, sc, chunks=4)
            assert all(g == got[0] for g in got) and got[0] == pt_from_limbs(curve, group, exp, einf), (curve, group, name)
        b.free()
    finally:
        for name, v in (("msm_window_bits", 0), ("msm_tail_fold", 16), ("msm_fold_run", 0)):
            ctx.set_option(name, v)
