"""End-to-end through the header-only C++ shim that mirrors the reference's interfaces
(crypto3-zk_amd/include/nil/crypto3/zk/hip/): r1cs_gg_ppzksnark_prover_hip::process and the batched KZG commit,
driven with inputs from the oracle and compared bit-for-bit with the oracle's proof / commitments.
BASELINE config 1 (Groth16 on alt_bn128, 2^10 constraints) runs here with the MSM / NTT on the GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, FQ_LIMBS, fr_arr, fr_ints, limbs, lookup_instance, permutation_instance, pt_limbs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(ROOT, "tests", "cpp", "libshimtest.so")
    srcs = [os.path.join(ROOT, "tests", "cpp", f) for f in ("shim_test.cpp", "arguments_test.cpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    return ctypes.CDLL(so)


_KEEP = []


def P(a):
    a = np.ascontiguousarray(a)
    _KEEP.append(a)  # ctypes pointers do not own the array: keep temporaries alive for the call
    return a.ctypes.data_as(ctypes.c_void_p)


def _oracle_domain(g, curve, M, n, domain):
    """the oracle instance on the domain the test names: "basic" = 2^ceil(log2(M + n + 1)) points (the oracle's default), "ref" = what
    make_evaluation_domain(M + n + 1) picks (r1cs_to_qap.hpp:229-230).  Both use the same root: the primitive 2^ceil(log2)-th one."""
    C = CURVES[curve]
    w = limbs(C.root_of_unity((M + n).bit_length()), 4)
    if domain == "ref":
        kind, m = cp.domain_choice(M + n + 1, C.two_adicity)
        g.set_domain(kind, m, w)
    elif domain == "extended":    # <omega> and shift <omega> over the next power of two: named explicitly (shim_set_domain)
        m = 1 << (M + n).bit_length()
        w = limbs(C.root_of_unity(m.bit_length() - 2), 4)
        g.set_domain(1, m, w, limbs(pow(C.fr_generator, 2, C.r), 4))
    return w


# M + n + 1 = 2^4+3 -> step(20), 2^10+11 -> step(2^10+16), 2^15+11 -> step(2^15+16), 111 -> basic(128)
@pytest.mark.parametrize("curve,M,n,world,domain", [(1, 1024, 10, 1, "basic"), (0, 1024, 10, 1, "basic"), (0, 100, 10, 1, "basic"), (0, 1 << 15, 10, 1, "basic"),
                                                    (0, 1024, 10, 2, "basic"), (1, 1024, 10, 3, "basic"), (0, 100, 10, 8, "basic"),
                                                    (0, 16, 2, 1, "ref"), (1, 1024, 10, 1, "ref"), (0, 1024, 10, 1, "ref"), (0, 1 << 15, 10, 1, "ref"),
                                                    (0, 1024, 10, 2, "ref"), (1, 1024, 10, 3, "ref"), (0, 100, 10, 1, "extended"), (1, 1024, 10, 2, "extended"),
                                                    (0, 1024, 10, 4, "ref"), (1, 1024, 10, 4, "basic"), (0, 1 << 15, 10, 2, "ref")])
def test_groth16_prover_shim(shim, curve, M, n, world, domain):
    """A key made by the oracle, over the basic domain of the next power of two or over the domain the reference's
    make_evaluation_domain picks (the shim tells them apart by the H query's size), proven with through the shim: bit-exact
    against the oracle's proof.  world > 1: the same proof sharded over `world` ranks (each holding a slice of every query;
    process_partial, the all-gather emulated by concatenation, finish) must equal the single-GPU proof -- SURVEY 8e's
    point-range partition -- and so must the proof of the DEVICE GROUP (r1cs_gg_ppzksnark_proving_key_group_hip: `world` contexts behind one
    caller, the 864-byte exchange inside the library, every transport the box offers; on a one-GPU box all members sit on device 0),
    through the group key and through the reference's static process(proving_key, x, w) over the default group."""
    import torch
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    if domain == "extended":
        C = CURVES[curve]
        shim.shim_set_domain(1, ctypes.c_size_t(1 << (M + n).bit_length()), P(limbs(pow(C.fr_generator, 2, C.r), 4)))
    try:
        _groth16_prover_shim(shim, curve, M, n, domain)
    finally:
        shim.shim_set_world(1)
        shim.shim_set_domain(-1, ctypes.c_size_t(0), None)


def _groth16_prover_shim(shim, curve, M, n, domain):
    C = CURVES[curve]
    g = cp.Groth16(curve, M, n, seed=1)
    w = _oracle_domain(g, curve, M, n, domain)
    gen = limbs(C.fr_generator, 4)
    rng = po.SplitMix64(2024)
    trap = fr_arr([rng.next_mod(C.r) for _ in range(5)])
    r_, s_ = limbs(rng.next_mod(C.r), 4), limbs(rng.next_mod(C.r), 4)
    g.keygen(trap, w)
    expected = g.prove(r_, s_, w, gen, chunks=cp.num_threads())
    csr = [g.csr(k) for k in range(3)]
    aq, ainf = g.query(0)
    bh, bhinf = g.query(1)
    bg, bginf = g.query(2)
    hq, _ = g.query(3)
    lq, _ = g.query(4)
    f1, _ = g.query(5)
    f2, _ = g.query(6)
    assert (bhinf == bginf).all()
    proof = np.zeros_like(expected)
    args = []
    for rp, cl, cf in csr:
        args += [P(rp), P(cl), P(cf)]
    rc = shim.shim_groth16_prove(curve, ctypes.c_size_t(g.M), ctypes.c_size_t(g.n), ctypes.c_size_t(g.N), *args, P(aq), P(ainf), P(bg), P(bh),
                                 P(bhinf), P(hq), ctypes.c_size_t(g.m - 1), P(lq), P(f1), P(f2), P(g.assignment()), P(w), P(gen), P(r_), P(s_),
                                 P(proof))
    assert rc == 0
    assert (proof == expected).all()
    if M == 100:
        # and against the trapdoor identity, independently of both implementations' MSM / NTT
        cs, prim, aux = po.r1cs_example_field_input(C.r, M, n, seed=1)
        rng = po.SplitMix64(2024)
        tr = [rng.next_mod(C.r) for _ in range(5)]
        rr, ss = rng.next_mod(C.r), rng.next_mod(C.r)
        wd = C.root_of_unity(g.log_m)
        if domain == "extended":
            wd = po.EvaluationDomain(po.EvaluationDomain.EXTENDED, g.m, C.root_of_unity(g.log_m - 1), C.r, pow(C.fr_generator, 2, C.r))
        eA, eB, eC = po.groth16_expected_in_exponent(C, cs, prim, aux, tr, rr, ss, wd)
        assert (proof == np.concatenate([pt_limbs(curve, 1, eA), pt_limbs(curve, 2, eB), pt_limbs(curve, 1, eC)])).all()


@pytest.mark.parametrize("curve,M,n,domain", [(0, 100, 10, "ref"), (0, 16, 2, "ref"), (1, 1024, 10, "ref"), (1, 1024, 10, "basic"), (0, (1 << 15) + 5, 3, "ref"),
                                              (0, (1 << 20) - 11, 10, "ref"), (0, 1 << 20, 10, "ref"), (0, 1 << 20, 10, "basic")])
def test_groth16_device_generated_key_matches_trapdoor(shim, curve, M, n, domain):
    """BASELINE config 4 AT ITS SIZE (M = 2^20, n = 10) over the domain the reference reduces over -- make_evaluation_domain(2^20 + 11)
    = the step radix-2 domain of 2^20 + 16 points -- and over the basic domain of 2^21 points, plus the m = 2^20 variant
    (M = 2^20 - 11): the key is generated on the device from a fixed trapdoor (r1cs_gg_ppzksnark_generator_hip =
    generator.hpp:240-377 with the batch exponentiations on the GPU), one proof is made with injected (r, s), and it must
    EQUAL the proof the trapdoor dictates (A = a G1, B = b G2, C = c G1; prover.hpp:141,145,151-153) -- a, b, c computed by
    the oracle without any MSM / NTT.  A wrong bucket, twiddle or query entry anywhere in the key or the transforms changes
    the result."""
    C = CURVES[curve]
    g = cp.Groth16(curve, M, n, seed=1)
    w = _oracle_domain(g, curve, M, n, domain)
    shim.shim_set_domain(0 if domain == "basic" else -1, ctypes.c_size_t(g.m if domain == "basic" else 0), None)
    gen = limbs(C.fr_generator, 4)
    rng = po.SplitMix64(4242)
    trap = fr_arr([rng.next_mod(C.r) for _ in range(5)])
    r_, s_ = limbs(rng.next_mod(C.r), 4), limbs(rng.next_mod(C.r), 4)
    args = []
    for k in range(3):
        rp, cl, cf = g.csr(k)
        args += [P(rp), P(cl), P(cf)]
    L1, L2 = 2 * FQ_LIMBS[curve], 4 * FQ_LIMBS[curve]
    proof = np.zeros(2 * L1 + L2, dtype=np.uint64)
    ms = np.zeros(2, dtype=np.float64)
    try:
        rc = shim.shim_groth16_generate_prove(curve, ctypes.c_size_t(g.M), ctypes.c_size_t(g.n), ctypes.c_size_t(g.N), *args, P(g.assignment()), P(trap), P(w),
                                              P(gen), P(r_), P(s_), P(proof), P(ms))
    finally:
        shim.shim_set_domain(-1, ctypes.c_size_t(0), None)
    assert rc == 0
    a, b, c = g.expected_exponents(trap, w, r_, s_)
    eA, _ = cp.batch_mul(curve, 1, a.reshape(1, 4))
    eB, _ = cp.batch_mul(curve, 2, b.reshape(1, 4))
    eC, _ = cp.batch_mul(curve, 1, c.reshape(1, 4))
    assert (proof == np.concatenate([eA[0], eB[0], eC[0]])).all()
    print("M = %d, domain of %d points: key generated on the device in %.0f ms, first proof %.1f ms" % (M, g.m, ms[0], ms[1]))


@pytest.mark.parametrize("curve,log_n,batch", [(0, 10, 3), (1, 8, 2), (0, 19, 2)])
def test_kzg_commit_shim(shim, zk, ctx, curve, log_n, batch):
    """commit(batch) = per column iNTT + MSM against {alpha^i G} (kzg.hpp:427-435); alpha = 7 as placeholder.cpp:175.
    2^19 rows: columns longer than one staging slice of upload_scalars (2^18 elements)."""
    C = CURVES[curve]
    n = 1 << log_n
    alpha = 7
    big = log_n >= 16
    if big:  # the SRS by the device's fixed-base kernel (itself checked against the oracle in test_gpu_msm.py)
        x, pw = 1, np.empty((n, 4), dtype=np.uint64)
        for i in range(n):
            pw[i] = limbs(x, 4)
            x = x * alpha % C.r
        b = ctx.bases_from_scalars(curve, 1, pw)
        srs, _ = b.download()
        b.free()
    else:
        srs, _ = cp.batch_mul(curve, 1, fr_arr([pow(alpha, i, C.r) for i in range(n)]))
    w = limbs(C.root_of_unity(log_n), 4)
    evals = cp.random_fr(curve, 5, batch * n).reshape(batch, n, 4)
    coeffs = cp.ntt(curve, evals, log_n, w, inverse=True)
    out = np.zeros((batch, srs.shape[1]), dtype=np.uint64)
    oinf = np.zeros(batch, dtype=np.uint8)
    assert shim.shim_kzg_commit(curve, P(srs), ctypes.c_size_t(n), P(evals), ctypes.c_size_t(log_n), ctypes.c_size_t(batch), P(w), P(out), P(oinf)) == 0
    for b in range(batch):
        if not big:
            exp, einf = cp.msm(curve, 1, srs, coeffs[b], chunks=4)
            assert oinf[b] == einf and (out[b] == exp).all()
        # the commitment is f(alpha) * G: Horner on the coefficients
        fa = cp.fr_horner(curve, coeffs[b], limbs(alpha, 4))
        pt, _ = cp.batch_mul(curve, 1, fa.reshape(1, 4))
        assert oinf[b] == 0 and (out[b] == pt[0]).all()
    # the same batch through an adapter whose scalar values are NOT canonical limbs in memory (the stand-in for crypto3-algebra's
    # Montgomery-form field type): upload_scalars converts on host threads instead of copying the vectors as they lie
    out2 = np.zeros_like(out)
    oinf2 = np.zeros_like(oinf)
    assert shim.shim_kzg_commit_foreign(curve, P(srs), ctypes.c_size_t(n), P(evals), ctypes.c_size_t(log_n), ctypes.c_size_t(batch), P(w), P(out2), P(oinf2)) == 0
    assert (out2 == out).all() and (oinf2 == oinf).all()


def _srs(curve, alpha, n):
    C = CURVES[curve]
    pts, _ = cp.batch_mul(curve, 1, fr_arr([pow(alpha, i, C.r) for i in range(n)]))
    return pts


@pytest.mark.parametrize("curve", [0, 1])
def test_kzg_basic_proof_eval_shim(shim, curve):
    """kzg_basic_test (t/commitment/kzg.cpp:75-103) on this curve: f = {-1, 1, 2, 3}, alpha = 10, z = 2.
    commit = 3209 G (the literal of :97); proof = commit((f - f(2)) / (X - 2)) = commit(3X^2 + 8X + 17) = 397 G."""
    C = CURVES[curve]
    srs = _srs(curve, 10, 16)
    f = fr_arr([C.r - 1, 1, 2, 3])
    out = np.zeros(srs.shape[1], dtype=np.uint64)
    assert shim.shim_kzg_basic_proof(curve, P(srs), ctypes.c_size_t(16), P(f), ctypes.c_size_t(4), P(limbs(2, 4)), P(out)) == 0
    exp, _ = cp.batch_mul(curve, 1, fr_arr([397]))
    assert (out == exp[0]).all()
    # a random polynomial longer than one Horner workgroup chunk, random point
    n = 300
    coeffs = cp.random_fr(curve, 9, n)
    z = po.SplitMix64(4).next_mod(C.r)
    srs = _srs(curve, 10, n)
    assert shim.shim_kzg_basic_proof(curve, P(srs), ctypes.c_size_t(n), P(coeffs), ctypes.c_size_t(n), P(limbs(z, 4)), P(out)) == 0
    fi = [po.from_limbs(x) for x in coeffs]
    q, rem = po.poly_divmod(po.poly_sub(fi, [po.poly_eval(fi, z, C.r)], C.r), [(-z) % C.r, 1], C.r)
    assert not rem
    exp, _ = cp.batch_mul(curve, 1, fr_arr([po.poly_eval(q, 10, C.r)]))
    assert (out == exp[0]).all()


def _vk(curve, alpha, n):
    C = CURVES[curve]
    pts, _ = cp.batch_mul(curve, 2, fr_arr([pow(alpha, i, C.r) for i in range(n)]))
    return pts


@pytest.mark.parametrize("curve", [0, 1])
def test_kzg_batched_free_functions_shim(shim, curve):
    """batched_kzg's algorithms (kzg.hpp:322-630) as free functions over coefficient-form polynomials: first the reference's own
    batched_kzg_basic_test inputs (test/commitment/kzg.cpp:535-572: f = 1 + 2X + ... + 8X^7, alpha = 7, S = {101, 2, 3}), then a
    ragged batch (different lengths and point sets, a polynomial shorter than its point set, shared points): commitments,
    merged points, the transcript walk and the single quotient commitment against the oracle's restatement of proof_eval."""
    C = CURVES[curve]
    r = C.r
    alpha = 7
    rng = po.SplitMix64(311 + curve)
    x1, x2, x3 = (rng.next_mod(r) for _ in range(3))
    cases = [[([1, 2, 3, 4, 5, 6, 7, 8], [101, 2, 3])],
             [([rng.next_mod(r) for _ in range(300)], [x1, x2]), ([rng.next_mod(r) for _ in range(77)], [x2]),
              ([rng.next_mod(r) for _ in range(2)], [x3, x1, x2]), ([rng.next_mod(r) for _ in range(1024)], [x1, x2, x3]),
              ([rng.next_mod(r) for _ in range(5)], [])]]
    g = lambda v: cp.batch_mul(curve, 1, fr_arr([v % r]))[0][0]
    u64 = lambda v: np.array(v, dtype=np.uint64)
    for case in cases:
        npolys = len(case)
        n_srs = max(len(f) for f, _ in case)
        srs = _srs(curve, alpha, n_srs)
        gamma = rng.next_mod(r)
        _, accum = po.kzg_v1_proof_eval(r, {0: [f for f, _ in case]}, {0: [pts for _, pts in case]}, gamma)
        merged_exp = sorted({x for _, pts in case for x in pts})
        commits = np.zeros((npolys, srs.shape[1]), dtype=np.uint64)
        merged = np.zeros((max(1, len(merged_exp)) + 4, 4), dtype=np.uint64)
        n_merged = np.zeros(1, dtype=np.uint64)
        proof = np.zeros(srs.shape[1], dtype=np.uint64)
        absorbed = np.zeros(2, dtype=np.uint64)
        allpts = fr_arr([x for _, pts in case for x in pts]) if any(pts for _, pts in case) else np.zeros((1, 4), dtype=np.uint64)
        rc = shim.shim_kzg_batched(curve, P(srs), ctypes.c_size_t(n_srs), ctypes.c_size_t(npolys), P(u64([len(f) for f, _ in case])),
                                   P(fr_arr([c for f, _ in case for c in f])), P(u64([len(pts) for _, pts in case])), P(allpts), P(limbs(gamma, 4)),
                                   P(commits), P(merged), P(n_merged), P(proof), P(absorbed))
        assert rc == 0
        for p, (f, _) in enumerate(case):
            assert (commits[p] == g(po.poly_eval(f, alpha, r))).all(), p
        assert int(n_merged[0]) == len(merged_exp) and [po.from_limbs(x) for x in merged[: len(merged_exp)]] == merged_exp
        # update_transcript: every commitment, every point of every S, every coefficient of every r (:323-372)
        # (r_i has |S_i| coefficients)
        assert int(absorbed[0]) == npolys and int(absorbed[1]) == 2 * sum(len(pts) for _, pts in case)
        assert any(accum)
        e1, i1 = cp.msm(curve, 1, srs[: len(accum)], fr_arr(accum), chunks=2)
        assert i1 == 0 and (proof == e1).all()
        assert (proof == g(po.poly_eval(accum, alpha, r))).all()


@pytest.mark.parametrize("curve,world", [(0, 1), (1, 1), (0, 3), (1, 2)])
def test_kzg_v1_proof_eval_shim(shim, curve, world):
    """kzg_commitment_scheme (the first batched scheme, kzg.hpp:636-873): commit + proof_eval through the shim class against the
    oracle's restatement of :782-807 -- evaluations z, the single quotient commitment kzg_proof, the verifier's equation in
    the exponent with alpha known -- and commit_g2 (:497-510, 659-664) against the G2 MSM oracle.
    world > 1: additionally the scheme over a device group (columns dealt, the quotient commitment cut by point range over the members'
    key replicas) must give the same commitments, evaluations and proof."""
    import torch
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        _kzg_v1_proof_eval_shim(shim, curve)
    finally:
        shim.shim_set_world(1)


def _kzg_v1_proof_eval_shim(shim, curve):
    C = CURVES[curve]
    r = C.r
    alpha = 7
    rng = po.SplitMix64(177 + curve)
    x1, x2, x3 = (rng.next_mod(r) for _ in range(3))
    # three polynomials sharing {x1, x2}, one with the same set in another order, ragged sets, and one without points
    layout = [(0, 6, [x1, x2]), (0, 6, [x1, x2]), (0, 7, [x2, x1]), (2, 6, [x2]), (2, 7, [x1, x3, x2]), (2, 5, [])]
    npolys = len(layout)
    evals, coeffs = [], []
    for p, (_, log_n, _) in enumerate(layout):
        e = cp.random_fr(curve, 900 + p, 1 << log_n)
        evals.append(e)
        c = cp.ntt(curve, e.reshape(1, -1, 4), log_n, limbs(C.root_of_unity(log_n), 4), inverse=True)[0]
        coeffs.append([po.from_limbs(x) for x in c])
    n_srs, n_vk = 128, 6
    srs, vk = _srs(curve, alpha, n_srs), _vk(curve, alpha, n_vk)
    gamma = rng.next_mod(r)
    polys, points = {}, {}
    for p, (k, _, pts) in enumerate(layout):
        polys.setdefault(k, []).append(coeffs[p])
        points.setdefault(k, []).append(pts)
    z, accum = po.kzg_v1_proof_eval(r, polys, points, gamma)
    merged = sorted({x for _, _, pts in layout for x in pts})
    g2_poly = po.vanishing_poly(merged, r)        # what verify_eval commits in G2: V(T) (:862-865)

    u64 = lambda v: np.array(v, dtype=np.uint64)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(8)])
    allpts = fr_arr([x for _, _, pts in layout for x in pts])
    commits = np.zeros((npolys, srs.shape[1]), dtype=np.uint64)
    zvals = np.zeros((len(allpts), 4), dtype=np.uint64)
    proof = np.zeros(srs.shape[1], dtype=np.uint64)
    g2_out = np.zeros((2, vk.shape[1]), dtype=np.uint64)
    absorbed = np.zeros(2, dtype=np.uint64)
    rc = shim.shim_kzg_v1_proof_eval(curve, P(srs), ctypes.c_size_t(n_srs), P(vk), ctypes.c_size_t(n_vk), ctypes.c_size_t(npolys),
                                     P(u64([k for k, _, _ in layout])), P(u64([l for _, l, _ in layout])), P(np.concatenate(evals)),
                                     P(u64([len(p) for _, _, p in layout])), P(allpts), P(roots), P(limbs(gamma, 4)), P(fr_arr(g2_poly)),
                                     ctypes.c_size_t(len(g2_poly)), P(commits), P(zvals), P(proof), P(g2_out), P(absorbed))
    assert rc == 0
    g = lambda v: cp.batch_mul(curve, 1, fr_arr([v % r]))[0][0]
    for p in range(npolys):
        assert (commits[p] == g(po.poly_eval(coeffs[p], alpha, r))).all(), p
    exp_z = [v for k in sorted(z) for zl in z[k] for v in zl]
    assert [po.from_limbs(x) for x in zvals] == exp_z
    # kzg_proof against the oracle's accum: through the MSM oracle, and in the exponent
    e1, i1 = cp.msm(curve, 1, srs[: len(accum)], fr_arr(accum), chunks=2)
    assert i1 == 0 and (proof == e1).all()
    assert (proof == g(po.poly_eval(accum, alpha, r))).all()
    # the verifier's equation (:809-868) with alpha in the clear: sum_j gamma^j (f_j(alpha) - U_j(alpha)) Z_{T \ S_j}(alpha) == accum(alpha) V_T(alpha)
    lhs, fac = 0, 1
    for k in sorted(polys):
        for i, c in enumerate(polys[k]):
            U = po.lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            diff = po.vanishing_poly([x for x in merged if x not in points[k][i]], r)
            lhs = (lhs + fac * (po.poly_eval(c, alpha, r) - po.poly_eval(U, alpha, r)) * po.poly_eval(diff, alpha, r)) % r
            fac = fac * gamma % r
    assert lhs == po.poly_eval(accum, alpha, r) * po.poly_eval(g2_poly, alpha, r) % r
    # commit_g2(V(T)): the member and the free function, against the G2 MSM oracle and in the exponent
    e2, i2 = cp.msm(curve, 2, vk[: len(g2_poly)], fr_arr(g2_poly), chunks=1)
    assert i2 == 0 and (g2_out[0] == e2).all() and (g2_out[1] == e2).all()
    assert (g2_out[0] == cp.batch_mul(curve, 2, fr_arr([po.poly_eval(g2_poly, alpha, r)]))[0][0]).all()
    # transcript traffic: 6 commitments; 8 evaluations + 8 U coefficients
    assert absorbed[0] == npolys and absorbed[1] == 2 * len(allpts)


@pytest.mark.parametrize("curve", [0, 1])
def test_multiexp_reference_arity_shim(shim, curve):
    """A KZG parameter struct declared like the reference's (kzg.hpp:76-135) that only shadows `multiexp_method` with the device
    policy: its call sites -- multiexp<typename KZG::multiexp_method>(b0, b1, s0, s1, 1), no context argument (kzg.hpp:146, 417,
    505) -- compile against the shim and run on the thread's default context (or the caller's): commit == f(alpha) G1,
    commit_g2 == g(alpha) G2."""
    C = CURVES[curve]
    r, alpha = C.r, 10
    srs, vk = _srs(curve, alpha, 300), _vk(curve, alpha, 5)
    f = cp.random_fr(curve, 31, 300)
    f[:4] = fr_arr([r - 1, 1, 0, 3])      # scalars 0, 1, r - 1 among them (multiexp_with_mixed_addition's special cases)
    gpoly = cp.random_fr(curve, 32, 5)
    for own in (0, 1):
        o1 = np.zeros(srs.shape[1], dtype=np.uint64)
        o2 = np.zeros(vk.shape[1], dtype=np.uint64)
        assert shim.shim_kzg_reference_arity(curve, P(srs), ctypes.c_size_t(300), P(vk), ctypes.c_size_t(5), P(f), ctypes.c_size_t(300), P(gpoly),
                                             ctypes.c_size_t(5), own, P(o1), P(o2)) == 0
        assert (o1 == cp.batch_mul(curve, 1, fr_arr([po.poly_eval(fr_ints(f), alpha, r)]))[0][0]).all()
        assert (o2 == cp.batch_mul(curve, 2, fr_arr([po.poly_eval(fr_ints(gpoly), alpha, r)]))[0][0]).all()
    # kzg_basic_test's literal through this path (kzg.cpp:83-97): commit({-1, 1, 2, 3}, alpha = 10) = 3209 G
    o1 = np.zeros(srs.shape[1], dtype=np.uint64)
    o2 = np.zeros(vk.shape[1], dtype=np.uint64)
    assert shim.shim_kzg_reference_arity(curve, P(srs), ctypes.c_size_t(300), P(vk), ctypes.c_size_t(5), P(fr_arr([r - 1, 1, 2, 3])), ctypes.c_size_t(4),
                                         P(gpoly), ctypes.c_size_t(5), 0, P(o1), P(o2)) == 0
    assert (o1 == cp.batch_mul(curve, 1, fr_arr([3209]))[0][0]).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_placeholder_call_sequence_with_foreign_polynomial_type(shim, curve):
    """VERDICT r3 missing #2: the scheme classes are templated on PolynomialType like the reference's polys_evaluator
    (batched_commitment.hpp:56-64).  tests/cpp/shim_test.cpp drives placeholder's exact call sequence (preprocessor.hpp:481-489,
    prover.hpp:129-141, 170, 202-213, 314-317, 363-410; permutation_argument.hpp:137) against the placeholder-facing KZG scheme and
    the LPC scheme, once with the shim's own polynomial_dfs and once with a FOREIGN class whose storage is private (begin / end /
    size / operator[] only, like math::polynomial_dfs): both must give the same bytes (checked in C++), and the KZG run is held
    against the oracle here -- commitments = f(alpha) G, evaluations, pi_1, pi_2 = po.kzg_v2_proof_eval."""
    C = CURVES[curve]
    r, alpha, log_n, nw = C.r, 7, 6, 3
    n = 1 << log_n
    npolys = 8 + nw
    rng = po.SplitMix64(4100 + curve)
    ch, theta, theta2 = (rng.next_mod(r) for _ in range(3))
    omega = C.root_of_unity(log_n)
    evals, coeffs = [], []
    for p in range(npolys):
        e = cp.random_fr(curve, 4200 + p, n)
        evals.append(e)
        c = cp.ntt(curve, e.reshape(1, -1, 4), log_n, limbs(omega, 4), inverse=True)[0]
        coeffs.append([po.from_limbs(x) for x in c])
    srs = _srs(curve, alpha, n)
    L1 = srs.shape[1]
    polys = {0: coeffs[0:4], 1: coeffs[4:5 + nw], 2: [coeffs[5 + nw]], 3: coeffs[6 + nw:]}
    points = {0: [[ch], [ch], [ch, ch * omega % r], [ch, ch * omega % r]],
              1: [[ch, ch * (omega * omega if i & 1 else omega) % r] for i in range(nw + 1)],
              2: [[ch, ch * omega % r]], 3: [[ch], [ch]]}
    z, f, Lq = po.kzg_v2_proof_eval(r, polys, points, theta, theta2)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 2)])
    out = np.zeros(4096, dtype=np.uint64)
    out_len = np.zeros(1, dtype=np.uint64)
    rc = shim.shim_placeholder_sequence(curve, P(srs), ctypes.c_size_t(n), P(np.concatenate(evals)), ctypes.c_size_t(npolys), ctypes.c_size_t(log_n),
                                        ctypes.c_size_t(nw), P(roots), P(limbs(ch, 4)), P(np.concatenate([limbs(theta, 4), limbs(theta2, 4)])), P(out),
                                        ctypes.c_size_t(len(out)), P(out_len))
    assert rc == 0
    out = out[: int(out_len[0])]
    sizes = [int(x) for x in out[:4]]
    assert sizes == [len(polys[b]) * L1 * 8 for b in range(4)]      # byte blobs: one packed point per polynomial (kzg_v2.hpp:208-226)
    at = 4
    g = lambda v: cp.batch_mul(curve, 1, fr_arr([v % r]))[0][0]
    for b in range(4):
        for c in polys[b]:
            assert (out[at:at + L1] == g(po.poly_eval(c, alpha, r))).all(), b
            at += L1
    exp_z = [v for k in sorted(z) for zl in z[k] for v in zl]
    got_z = [po.from_limbs(out[at + 4 * i: at + 4 * i + 4]) for i in range(len(exp_z))]
    assert got_z == exp_z
    at += 4 * len(exp_z)
    assert (out[at:at + L1] == g(po.poly_eval(f, alpha, r))).all() and (out[at + L1:at + 2 * L1] == g(po.poly_eval(Lq, alpha, r))).all()
    assert at + 2 * L1 + 1 == len(out)


@pytest.mark.parametrize("curve,log_n,k,chunks", [(0, 6, 3, 0), (1, 6, 2, 0), (0, 8, 4, 0), (0, 5, 1, 0), (0, 6, 3, 2), (1, 5, 4, 3), (0, 6, 5, 3), (0, 5, 2, 8)])
def test_placeholder_permutation_argument_shim(shim, curve, log_n, k, chunks):
    """placeholder's permutation argument, prover side, on the device (hip/placeholder_permutation.hpp; permutation_argument.hpp:70-224),
    one-part (max_quotient_chunks = 0) and multi-part (the reference's tests run 8, 10, 30, 50): the grand product V_P -- a serial loop
    with one inversion per row in the reference; chunks sharing an inversion + a three-level prefix-product scan on the device --
    EXACTLY the oracle's row-by-row recurrence, the intermediate polynomials of the multi-part form value by value, and the three
    constraint polynomials F_0, F_1, F_2 equal to the oracle's dense coefficient-form arithmetic.  A genuine copy-constraint instance
    (tests/util.py permutation_instance): the product closes at usable_rows."""
    C = CURVES[curve]
    r = C.r
    n = 1 << log_n
    usable = n - 3
    rng = po.SplitMix64(6100 + 10 * curve + log_n + k + 100 * chunks)
    cols, S_id, S_sigma = permutation_instance(C, rng, log_n, k, usable)
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    parts = 1 if chunks == 0 else -(-k // (chunks - 1))
    alphas = [rng.next_mod(r) for _ in range(parts - 1)]
    res = po.permutation_argument(cols, S_id, S_sigma, q_last, q_blind, L0, beta, gamma, C.root_of_unity, r, chunks, alphas, usable)
    V, F = res[0], res[1]
    currents = res[2] if chunks else []
    assert V[usable] == 1                                   # the permutation is satisfied: the grand product closes
    evals = fr_arr([x for v in cols + S_id + S_sigma + [q_last, q_blind, L0] for x in v])
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 4)])
    out_vp = np.zeros((n, 4), dtype=np.uint64)
    out_F = np.zeros((3, 8 * n, 4), dtype=np.uint64)
    sizes = np.zeros(3, dtype=np.uint64)
    out_parts = np.zeros((max(1, parts - 1), n, 4), dtype=np.uint64)
    al = np.stack([limbs(a, 4) for a in alphas] + [limbs(0, 4)])
    rc = shim.shim_placeholder_permutation(curve, P(evals), ctypes.c_size_t(k), ctypes.c_size_t(log_n), P(roots), P(limbs(beta, 4)), P(limbs(gamma, 4)),
                                           ctypes.c_size_t(chunks), P(al), ctypes.c_size_t(len(alphas)), ctypes.c_size_t(usable), P(out_vp), P(out_F), P(sizes),
                                           P(out_parts))
    assert rc == 0
    assert fr_ints(out_vp) == V
    for i, c in enumerate(currents):
        assert fr_ints(out_parts[i]) == c, i
    for f in range(3):
        got = po.poly_trim(fr_ints(out_F[f][: int(sizes[f])]))
        assert got == F[f], f


@pytest.mark.parametrize("curve,log_n,k_in,k_val,big,part_sizes", [(0, 6, 1, 1, (), None), (1, 6, 2, 1, (1,), None), (0, 7, 3, 2, (0,), None), (0, 5, 1, 2, (), None),
                                                                   (0, 6, 1, 1, (), [1, 1]), (1, 5, 2, 1, (1,), [2, 1]), (0, 6, 3, 2, (0,), [2, 1, 2])])
def test_placeholder_lookup_argument_shim(shim, curve, log_n, k_in, k_val, big, part_sizes):
    """placeholder's lookup argument, prover side, from the sorted vectors on (hip/placeholder_lookup.hpp; lookup_argument.hpp:153-296, one part):
    V_L -- a serial loop with one inversion per row in the reference (compute_V_L, :375-409); the permutation argument's scan on the device --
    EXACTLY the oracle's row-by-row recurrence (zero behind usable_rows), and the four constraint polynomials equal to the oracle's dense
    coefficient-form arithmetic.  A genuine instance (inputs drawn from the tables, `sorted` by the oracle's restatement of
    sort_polynomials, :565-638): the product closes, V_L[usable_rows] = 1 -- the reference's own BOOST_CHECK (:217)."""
    C = CURVES[curve]
    r = C.r
    n = 1 << log_n
    rng = po.SplitMix64(6400 + 10 * curve + log_n + k_in + 3 * k_val)
    inputs, values, usable = lookup_instance(C, rng, log_n, k_in, k_val, big)
    red_in = [po.reduce_dfs_polynomial_domain(f, n) for f in inputs]
    sorted_ = po.lookup_sort_polynomials(red_in, values, n, usable)
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    alphas = [rng.next_mod(r) for _ in range(k_in + k_val - 1)]
    part_alphas = [rng.next_mod(r) for _ in range(len(part_sizes) - 1)] if part_sizes else []
    res = po.lookup_argument(inputs, values, sorted_, q_last, q_blind, L0, beta, gamma, alphas, usable, C.root_of_unity, r, part_sizes, part_alphas)
    V, F = res[0], res[1]
    currents = res[2] if part_sizes else []
    assert V[usable] == 1 and all(v == 0 for v in V[usable + 1:])
    evals = fr_arr([x for v in inputs + values + sorted_ + [q_last, q_blind, L0] for x in v])
    in_logs = np.array([len(f).bit_length() - 1 for f in inputs], dtype=np.uint64)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 5)])
    out_vl = np.zeros((n, 4), dtype=np.uint64)
    out_F = np.zeros((4, 16 * n, 4), dtype=np.uint64)
    sizes = np.zeros(4, dtype=np.uint64)
    al = np.stack([limbs(a, 4) for a in alphas] + [limbs(0, 4)])
    ps = np.array(part_sizes or [], dtype=np.uint64)
    pa = np.stack([limbs(a, 4) for a in part_alphas] + [limbs(0, 4)])
    out_parts = np.zeros((max(1, len(currents)), n, 4), dtype=np.uint64)
    rc = shim.shim_placeholder_lookup(curve, P(evals), ctypes.c_size_t(k_in), P(in_logs), ctypes.c_size_t(k_val), ctypes.c_size_t(log_n), ctypes.c_size_t(usable),
                                      P(roots), P(limbs(beta, 4)), P(limbs(gamma, 4)), P(al), P(ps) if len(ps) else None, ctypes.c_size_t(len(ps)), P(pa), P(out_vl),
                                      P(out_F), P(sizes), P(out_parts))
    assert rc == 0
    assert fr_ints(out_vl) == V
    for i, c in enumerate(currents):
        assert fr_ints(out_parts[i]) == c, i
    for f in range(4):
        got = po.poly_trim(fr_ints(out_F[f][: int(sizes[f])]))
        assert got == F[f], f


def argument_instances(C, rng, log_n, k, options, columns, k_in, big):
    """a genuine permutation instance over k witness columns and a genuine lookup instance in the constraint system's OWN terms: a tag
    selector (1 on rows 1 .. T), `options * columns` constant columns (two adjacent rows equal in every column: a repeated table value),
    inputs that take values of the theta-compressed table (or zero) on the usable rows.  -> everything shim_placeholder_arguments_transcript takes"""
    r, n = C.r, 1 << log_n
    usable = n - 3
    cols, S_id, S_sigma = permutation_instance(C, rng, log_n, k, usable)
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    T = usable // 2
    tag = [1 if 1 <= j <= T else 0 for j in range(n)]
    consts = []
    for _ in range(options * columns):
        c = [rng.next_mod(r) for _ in range(n)]
        consts.append(c)
    for o in range(options):            # rows 3 and 4 hold the same table entry in every option
        for i in range(columns):
            consts[o * columns + i][4] = consts[o * columns + i][3]
    return usable, cols, S_id, S_sigma, q_last, q_blind, L0, tag, consts


@pytest.mark.parametrize("curve,log_n,k,chunks,options,columns,k_in,big,lookup_parts", [(0, 5, 2, 0, 1, 2, 1, (), None), (1, 5, 3, 3, 2, 1, 1, (0,), [2, 1]),
                                                                                      (0, 6, 4, 3, 1, 3, 2, (1,), [1, 1, 1]), (1, 6, 1, 0, 2, 2, 2, (), None)])
def test_placeholder_arguments_reference_entry_points(shim, curve, log_n, k, chunks, options, columns, k_in, big, lookup_parts):
    """placeholder's permutation and lookup arguments through the REFERENCE'S entry points (hip/placeholder_arguments.hpp):
    prove_eval(constraint_system, preprocessed_data, table_description, column_polynomials, commitment_scheme, transcript) and
    placeholder_lookup_argument_prover(constraint_system, preprocessed_data, plonk_columns, commitment_scheme, transcript).prove_eval(),
    driven in tests/cpp/arguments_test.cpp by stand-ins declared like the reference's classes, with a transcript and a commitment scheme
    that RECORD.  Held here against the oracle:
      * the recorded ORDER of challenge draws, append_to_batch, commit and absorb == the oracle's replay of permutation_argument.hpp:95-97,
        139, 181-183, 200 and lookup_argument.hpp:150, 192-206, 213, 267, 282-283 (BLS12-381 through transcript.challenge<FieldType>());
      * prepare_lookup_value on the device (the table columns compressed with theta, tagged, masked: a polynomial on 4n points),
        sort_polynomials on the device, V_P, V_L, every intermediate polynomial value by value, all seven F polynomials coefficient by
        coefficient == the oracle's;
      * (inside the harness) the explicit-challenge overloads give the same bits."""
    C = CURVES[curve]
    r, n = C.r, 1 << log_n
    rng = po.SplitMix64(7700 + 13 * curve + log_n + k + 5 * options + columns)
    usable, cols, S_id, S_sigma, q_last, q_blind, L0, tag, consts = argument_instances(C, rng, log_n, k, options, columns, k_in, big)
    perm_parts = 1 if chunks == 0 else -(-k // (chunks - 1))
    total = k_in + options
    lparts = lookup_parts or [total]
    assert sum(lparts) == total
    ch = [rng.next_mod(r) for _ in range(2 + perm_parts - 1 + 3 + len(lparts) - 1 + total - 1)]
    beta_p, gamma_p, alphas_p = ch[0], ch[1], ch[2:2 + perm_parts - 1]
    c0 = 2 + perm_parts - 1
    theta, beta_l, gamma_l = ch[c0], ch[c0 + 1], ch[c0 + 2]
    part_alphas = ch[c0 + 3:c0 + 3 + len(lparts) - 1]
    alphas_l = ch[c0 + 3 + len(lparts) - 1:]
    # the oracle's side of the lookup argument, from the constraint system's terms on
    mask = [(1 - a - b) % r for a, b in zip(q_last, q_blind)]
    tables = [(0, columns, [[o * columns + i for i in range(columns)] for o in range(options)])]
    values = po.lookup_prepare_value(tables, [tag], consts, theta, mask, C.root_of_unity, r)
    assert all(len(v) == 4 * n for v in values)
    red_val = [po.reduce_dfs_polynomial_domain(v, n) for v in values]
    for v in red_val:       # the definition on the rows: mask tag ((t + 1) + sum theta^(i + 1) constant_i)
        o = red_val.index(v)
        assert v == [mask[j] * tag[j] * (1 + sum(pow(theta, i + 1, r) * consts[o * columns + i][j] for i in range(columns))) % r for j in range(n)]
    pool = [0] + [x for v in red_val for x in v[:usable] if x]
    inputs = []
    for i in range(k_in):
        f = [pool[rng.next_mod(len(pool))] for _ in range(usable)] + [rng.next_mod(r) for _ in range(n - usable)]
        if i in big:
            bigf = po.dfs_resize(f, 2 * n, C.root_of_unity, r)
            c = rng.next_mod(r)
            f = [(v - 2 * c * (j & 1)) % r for j, v in enumerate(bigf)]
        inputs.append(f)
    red_in = [po.reduce_dfs_polynomial_domain(f, n) for f in inputs]
    sorted_ = po.lookup_sort_polynomials(red_in, red_val, n, usable)
    lres = po.lookup_argument(inputs, values, sorted_, q_last, q_blind, L0, beta_l, gamma_l, alphas_l, usable, C.root_of_unity, r, lookup_parts, part_alphas)
    VL, FL = lres[0], lres[1]
    l_currents = lres[2] if lookup_parts else []
    assert VL[usable] == 1
    pres = po.permutation_argument(cols, S_id, S_sigma, q_last, q_blind, L0, beta_p, gamma_p, C.root_of_unity, r, chunks, alphas_p, usable)
    VP, FP = pres[0], pres[1]
    p_currents = pres[2] if chunks else []
    assert VP[usable] == 1
    # through the harness
    alpha = 7
    srs = cp.batch_mul(curve, 1, fr_arr([pow(alpha, i, r) for i in range(n)]))[0]
    evals = fr_arr([x for v in cols + S_id + S_sigma + [q_last, q_blind, L0, tag] + consts + inputs for x in v])
    in_logs = np.array([len(f).bit_length() - 1 for f in inputs], dtype=np.uint64)
    lp = np.array(lparts, dtype=np.uint64)
    events = np.zeros(256, dtype=np.uint64)
    n_events = np.zeros(1, dtype=np.uint64)
    out_perm = np.zeros(((1 + len(p_currents)) * n + 3 * 8 * n, 4), dtype=np.uint64)
    perm_sizes = np.zeros(3, dtype=np.uint64)
    out_look = np.zeros(((1 + len(l_currents) + total) * n + 4 * 16 * n, 4), dtype=np.uint64)
    look_sizes = np.zeros(4, dtype=np.uint64)
    rc = shim.shim_placeholder_arguments_transcript(curve, P(srs), ctypes.c_size_t(n), P(evals), ctypes.c_size_t(k), ctypes.c_size_t(log_n), ctypes.c_size_t(usable),
                                                    ctypes.c_size_t(chunks), ctypes.c_size_t(perm_parts), ctypes.c_size_t(options), ctypes.c_size_t(columns),
                                                    ctypes.c_size_t(k_in), P(in_logs), P(lp), ctypes.c_size_t(len(lparts)), P(np.stack([limbs(c, 4) for c in ch])),
                                                    ctypes.c_size_t(len(ch)), P(events), ctypes.c_size_t(len(events)), P(n_events), P(out_perm), P(perm_sizes),
                                                    P(out_look), P(look_sizes))
    assert rc == 0
    ev = [int(x) for x in events[: int(n_events[0])]]
    cut = ev.index(0)
    assert ev[:cut] == po.permutation_argument_events(perm_parts)
    assert ev[cut + 1:] == po.lookup_argument_events(total, len(lparts))
    # the permutation argument's polynomials
    at = 0
    assert fr_ints(out_perm[at:at + n]) == VP
    at += n
    for c in p_currents:
        assert fr_ints(out_perm[at:at + n]) == c
        at += n
    for f in range(3):
        got = po.poly_trim(fr_ints(out_perm[at + f * 8 * n: at + f * 8 * n + int(perm_sizes[f])]))
        assert got == FP[f], f
    # the lookup argument's
    at = 0
    assert fr_ints(out_look[at:at + n]) == VL
    at += n
    for c in l_currents:
        assert fr_ints(out_look[at:at + n]) == c
        at += n
    for sv in sorted_:
        assert fr_ints(out_look[at:at + n]) == sv
        at += n
    for f in range(4):
        got = po.poly_trim(fr_ints(out_look[at + f * 16 * n: at + f * 16 * n + int(look_sizes[f])]))
        assert got == FL[f], f


@pytest.mark.parametrize("curve,n", [(0, 64), (1, 48), (0, 1040), (1, 4096)])
def test_column_range_polynomials_shim(shim, curve, n):
    """detail::column_range_polynomials / column_polynomial (arithmetization/plonk/detail/column_polynomial.hpp:43-72): the columns of a
    table to coefficient form by ONE batched inverse transform over the domain make_evaluation_domain returns for n rows (basic and
    step radix-2 here), against the oracle's inverse transform over the same domain, column by column."""
    C = CURVES[curve]
    dom = po.make_evaluation_domain(C, n)
    m, count = dom.m, 5
    a = cp.random_fr(curve, 9100 + n, count * m).reshape(count, m, 4)
    w, sh = limbs(dom.omega, 4), limbs(dom.shift, 4)
    exp = np.stack([cp.domain_fft(curve, dom.kind, a[c], w, sh, inverse=True) for c in range(count)])
    out = np.zeros((count, m, 4), dtype=np.uint64)
    rc = shim.shim_column_polynomials(curve, P(a), ctypes.c_size_t(count), dom.kind, ctypes.c_size_t(m), P(w), P(sh), P(out))
    assert rc == 0
    assert (out == exp).all()
    if m <= 64:
        xs = dom.elements()
        assert [po.poly_eval(fr_ints(out[0]), x, C.r) for x in xs] == fr_ints(a[0])


@pytest.mark.parametrize("curve,log_n,world", [(0, 6, 1), (1, 5, 1), (0, 6, 3), (1, 5, 2)])
def test_placeholder_round_composed_shim(shim, curve, log_n, world):
    """The pieces composed as placeholder_prover::process strings them (prover.hpp:170-218, 262-277, 220-259, 314-317): permutation
    argument, lookup argument and a gate argument over GENUINE instances (copy constraints closed inside the usable rows, inputs drawn
    from the table, w2 = w0 w1 where the selector is on), their eight constraint polynomials -- living on different domains -- consolidated
    with eight alphas, divided by X^n - 1 (exactly: every part vanishes on the rows), split and committed; V_P, V_L, the sorted vectors
    and the quotient parts go to the KZG scheme as device polynomials.  Against the oracle: T coefficient by coefficient, every
    commitment = polynomial(alpha) G.
    world > 1: the round's KZG scheme over a device group of that many members (the arguments' kernels on member 0; commit(batch) deals the
    resident polynomials over the members device to device): the same T and commitments."""
    import torch
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        _placeholder_round_composed_shim(shim, curve, log_n)
    finally:
        shim.shim_set_world(1)


def _placeholder_round_composed_shim(shim, curve, log_n):
    C = CURVES[curve]
    r, alpha, k = C.r, 7, 2
    n = 1 << log_n
    usable = n - 3
    root = C.root_of_unity
    rng = po.SplitMix64(8800 + curve + log_n)
    cols, S_id, S_sigma = permutation_instance(C, rng, log_n, k, usable)
    inputs, values, _ = lookup_instance(C, rng, log_n, 1, 1)
    sorted_ = po.lookup_sort_polynomials(inputs, values, n, usable)
    q = [1 if (j % 3 == 0 and j < usable) else 0 for j in range(n)]
    w0, w1 = ([rng.next_mod(r) for _ in range(n)] for _ in range(2))
    w2 = [w0[j] * w1[j] % r if q[j] else rng.next_mod(r) for j in range(n)]
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    ch = [rng.next_mod(r) for _ in range(13)]
    bp, gp, bl, gl, al = ch[:5]
    alphas = ch[5:]
    VP, Fp = po.permutation_argument(cols, S_id, S_sigma, q_last, q_blind, L0, bp, gp, root, r)
    VL, Fl = po.lookup_argument(inputs, values, sorted_, q_last, q_blind, L0, bl, gl, [al], usable, root, r)
    assert VP[usable] == 1 and VL[usable] == 1
    mask = [(1 - a - b) % r for a, b in zip(q_last, q_blind)]
    G = po.gate_argument_dfs([(1, [(q, 0), (w0, 0), (w1, 0)]), (r - 1, [(q, 0), (w2, 0)])], mask, 4 * n, root, r)

    def to_dfs(c, size=4 * n):
        assert len(c) <= size
        return po.ntt(list(c) + [0] * (size - len(c)), root(size.bit_length() - 1), r)
    F = [to_dfs(f) for f in Fp + Fl] + [G]
    T = po.quotient_polynomial(F, alphas, n, root, r)
    parts = po.quotient_polynomial_split_dfs(T, n, 4, n, root, r)
    srs = _srs(curve, alpha, n)
    L1 = srs.shape[1]
    evals = fr_arr([x for v in cols + S_id + S_sigma + inputs + values + sorted_ + [q, w0, w1, w2, q_last, q_blind, L0] for x in v])
    roots = np.stack([limbs(root(l), 4) for l in range(log_n + 4)])
    out_T = np.zeros((3 * n, 4), dtype=np.uint64)
    out_commits = np.zeros((8, L1), dtype=np.uint64)
    rc = shim.shim_placeholder_round(curve, P(srs), ctypes.c_size_t(n), P(evals), ctypes.c_size_t(k), ctypes.c_size_t(log_n), ctypes.c_size_t(usable), P(roots),
                                     P(np.stack([limbs(c, 4) for c in ch])), P(out_T), P(out_commits))
    assert rc == 0
    assert fr_ints(out_T) == T
    co = lambda e: po.intt(list(e), root(log_n), r)
    g = lambda c: cp.batch_mul(curve, 1, fr_arr([po.poly_eval(c, alpha, r)]))[0][0]
    expected = [co(VP), co(VL), co(sorted_[0]), co(sorted_[1])] + [T[i * n:(i + 1) * n] for i in range(4)]
    for i, c in enumerate(expected):
        assert (out_commits[i] == g(c)).all(), i


def _selectors(n, usable):
    return [1 if j == usable else 0 for j in range(n)], [1 if j > usable else 0 for j in range(n)], [1] + [0] * (n - 1)


@pytest.mark.parametrize("curve,log_n", [(0, 12), (1, 12), (0, 16), (1, 16)])
def test_placeholder_arguments_at_multipass_sizes(shim, curve, log_n):
    """the arguments' parity beyond toy sizes.  At 2^12 and 2^16 rows the device runs what the 2^20 legs run -- multi-pass
    transforms, subsampled extensions inside polynomial_product, the block cache's reuse -- and every output is held against the
    C++-backed oracle (oracle/cport.py permutation_argument / lookup_argument / quotient_polynomial, pinned to pyoracle's dense arithmetic
    at <= 2^8 rows in tests/test_oracle_kat.py), bit for bit:
      * permutation argument, 4 columns, one part and max_quotient_chunks = 3 (two parts): V_P, the intermediate polynomial, F_0 .. F_2;
      * lookup argument, 2 inputs (one on the 2n-point domain) over 1 table, one part and parts [2, 1]: sorted (oracle's map + walk),
        V_L, the intermediate polynomial, F_0 .. F_3;
      * the composed round (permutation + lookup + gate arguments, quotient of the eight parts, split, commitments): T coefficient by
        coefficient, every commitment = polynomial(alpha) G."""
    C = CURVES[curve]
    r, n = C.r, 1 << log_n
    usable = n - 3
    root = C.root_of_unity
    A = fr_arr
    rng = po.SplitMix64(9900 + curve + log_n)
    q_last, q_blind, L0 = _selectors(n, usable)
    roots = np.stack([limbs(root(l), 4) for l in range(log_n + 5)])
    # ---- permutation argument
    k = 4
    cols, S_id, S_sigma = permutation_instance(C, rng, log_n, k, usable)
    e_perm = fr_arr([x for v in cols + S_id + S_sigma + [q_last, q_blind, L0] for x in v])
    a_cols, a_sid, a_ssig = [A(c) for c in cols], [A(c) for c in S_id], [A(c) for c in S_sigma]
    for chunks in (0, 3):
        parts = 1 if chunks == 0 else -(-k // (chunks - 1))
        beta, gamma = rng.next_mod(r), rng.next_mod(r)
        alphas = [rng.next_mod(r) for _ in range(parts - 1)]
        want = cp.permutation_argument(curve, a_cols, a_sid, a_ssig, A(q_last), A(q_blind), A(L0), beta, gamma, chunks, alphas, usable)
        assert fr_ints(want[0][usable:usable + 1]) == [1]
        out_vp = np.zeros((n, 4), dtype=np.uint64)
        out_F = np.zeros((3, 8 * n, 4), dtype=np.uint64)
        sizes = np.zeros(3, dtype=np.uint64)
        out_parts = np.zeros((max(1, parts - 1), n, 4), dtype=np.uint64)
        al = np.stack([limbs(a, 4) for a in alphas] + [limbs(0, 4)])
        rc = shim.shim_placeholder_permutation(curve, P(e_perm), ctypes.c_size_t(k), ctypes.c_size_t(log_n), P(roots), P(limbs(beta, 4)), P(limbs(gamma, 4)),
                                               ctypes.c_size_t(chunks), P(al), ctypes.c_size_t(len(alphas)), ctypes.c_size_t(usable), P(out_vp), P(out_F), P(sizes),
                                               P(out_parts))
        assert rc == 0
        assert (out_vp == want[0]).all()
        for i, c in enumerate(want[2] if chunks else []):
            assert (out_parts[i] == c).all(), i
        for f in range(3):
            got = cp.poly_trim(out_F[f][: int(sizes[f])])
            assert got.shape == want[1][f].shape and (got == want[1][f]).all(), (chunks, f)
    # ---- lookup argument
    k_in, k_val = 2, 1
    inputs, values, usable_l = lookup_instance(C, rng, log_n, k_in, k_val, (1,))
    assert usable_l == usable
    a_in, a_val = [A(f) for f in inputs], [A(v) for v in values]
    red_in = [cp.reduce_dfs_polynomial_domain(f, n) for f in a_in]
    sorted_ = cp.lookup_sort_polynomials(red_in, a_val, n, usable)
    e_look = np.concatenate(a_in + a_val + sorted_ + [A(q_last), A(q_blind), A(L0)])
    in_logs = np.array([len(f).bit_length() - 1 for f in inputs], dtype=np.uint64)
    for part_sizes in (None, [2, 1]):
        beta, gamma = rng.next_mod(r), rng.next_mod(r)
        alphas = [rng.next_mod(r) for _ in range(k_in + k_val - 1)]
        part_alphas = [rng.next_mod(r) for _ in range(len(part_sizes) - 1)] if part_sizes else []
        want = cp.lookup_argument(curve, a_in, a_val, sorted_, A(q_last), A(q_blind), A(L0), beta, gamma, alphas, usable, part_sizes, part_alphas)
        assert fr_ints(want[0][usable:usable + 1]) == [1]
        currents = want[2] if part_sizes else []
        out_vl = np.zeros((n, 4), dtype=np.uint64)
        out_F = np.zeros((4, 16 * n, 4), dtype=np.uint64)
        sizes = np.zeros(4, dtype=np.uint64)
        al = np.stack([limbs(a, 4) for a in alphas] + [limbs(0, 4)])
        ps = np.array(part_sizes or [], dtype=np.uint64)
        pa = np.stack([limbs(a, 4) for a in part_alphas] + [limbs(0, 4)])
        out_parts = np.zeros((max(1, len(currents)), n, 4), dtype=np.uint64)
        rc = shim.shim_placeholder_lookup(curve, P(e_look), ctypes.c_size_t(k_in), P(in_logs), ctypes.c_size_t(k_val), ctypes.c_size_t(log_n), ctypes.c_size_t(usable),
                                          P(roots), P(limbs(beta, 4)), P(limbs(gamma, 4)), P(al), P(ps) if len(ps) else None, ctypes.c_size_t(len(ps)), P(pa), P(out_vl),
                                          P(out_F), P(sizes), P(out_parts))
        assert rc == 0
        assert (out_vl == want[0]).all()
        for i, c in enumerate(currents):
            assert (out_parts[i] == c).all(), i
        for f in range(4):
            got = cp.poly_trim(out_F[f][: int(sizes[f])])
            assert got.shape == want[1][f].shape and (got == want[1][f]).all(), (part_sizes, f)
    # ---- the composed round: 2 permuted columns, 1 input over 1 table, the gate q (w0 w1 - w2)
    k, alpha = 2, 7
    inputs1, values1, _ = lookup_instance(C, rng, log_n, 1, 1)
    a_in1, a_val1 = [A(f) for f in inputs1], [A(v) for v in values1]
    sorted1 = cp.lookup_sort_polynomials(a_in1, a_val1, n, usable)
    q = [1 if (j % 3 == 0 and j < usable) else 0 for j in range(n)]
    w0, w1 = ([rng.next_mod(r) for _ in range(n)] for _ in range(2))
    w2 = [w0[j] * w1[j] % r if q[j] else rng.next_mod(r) for j in range(n)]
    ch = [rng.next_mod(r) for _ in range(13)]
    bp, gp, bl, gl, al1 = ch[:5]
    alphas = ch[5:]
    VL, Fl = cp.lookup_argument(curve, a_in1, a_val1, sorted1, A(q_last), A(q_blind), A(L0), bl, gl, [al1], usable)[:2]
    co = lambda e: cp.dfs_coefficients(curve, A(e) if isinstance(e, list) else e)
    mask = [(1 - a - b) % r for a, b in zip(q_last, q_blind)]
    # gate argument (gates_argument.hpp:203-216): mask (q w0 w1 - q w2)
    cq = co(q)
    G = cp.poly_mul(curve, co(mask), cp.poly_sub(curve, cp.poly_mul(curve, cp.poly_mul(curve, cq, co(w0)), co(w1)), cp.poly_mul(curve, cq, co(w2))))
    # a 2-column copy-constraint instance of its own (the cycles of the 4-column one cross into columns 2 and 3)
    cols2, S_id2, S_sigma2 = permutation_instance(C, rng, log_n, k, usable)
    a2 = [A(c) for c in cols2], [A(c) for c in S_id2], [A(c) for c in S_sigma2]
    VP, Fp = cp.permutation_argument(curve, a2[0], a2[1], a2[2], A(q_last), A(q_blind), A(L0), bp, gp)[:2]
    T = cp.quotient_polynomial(curve, list(Fp) + list(Fl) + [G], alphas, n)
    assert len(T) <= 3 * n
    srs = _srs(curve, alpha, n)
    L1 = srs.shape[1]
    evals = np.concatenate(a2[0] + a2[1] + a2[2] + a_in1 + a_val1 + sorted1 + [A(q), A(w0), A(w1), A(w2), A(q_last), A(q_blind), A(L0)])
    out_T = np.zeros((3 * n, 4), dtype=np.uint64)
    out_commits = np.zeros((8, L1), dtype=np.uint64)
    rc = shim.shim_placeholder_round(curve, P(srs), ctypes.c_size_t(n), P(evals), ctypes.c_size_t(k), ctypes.c_size_t(log_n), ctypes.c_size_t(usable), P(roots),
                                     P(np.stack([limbs(c, 4) for c in ch])), P(out_T), P(out_commits))
    assert rc == 0
    Tp = np.concatenate([T, np.zeros((3 * n - len(T), 4), dtype=np.uint64)])
    assert (out_T == Tp).all()
    at_alpha = lambda c: cp.fr_horner(curve, c, limbs(alpha, 4)) if len(c) else limbs(0, 4)
    g = lambda c: cp.batch_mul(curve, 1, at_alpha(c).reshape(1, 4))[0][0]
    expected = [co(VP), co(VL), co(sorted1[0]), co(sorted1[1])] + [Tp[i * n:(i + 1) * n] for i in range(4)]
    for i, c in enumerate(expected):
        assert (out_commits[i] == g(c)).all(), i


def test_placeholder_transcript_bytes_bls12_381(shim):
    """a placeholder proof is bit-exact with the reference's only if the TRANSCRIPT absorbs the same bytes.  The
    placeholder-facing KZG scheme with the reference's encodings -- commitments as 48-byte compressed BLS12-381 points (the
    encoding pinned by the reference's own byte vectors, aggregation test :932-1010, through pyoracle.bls12_381_compress),
    scalars as 32 big-endian bytes -- must hand the transcript exactly what kzg_v2.hpp does: the prover's transcript(commitment
    blob) per batch (prover.hpp:142, 171, 207), then inside proof_eval per batch the blob in ONE call, every evaluation, every U
    coefficient (:150-190), then pi_1 (:265-272) and pi_2 (:296-304)."""
    curve = 0
    C = CURVES[curve]
    r, alpha, log_n, nw = C.r, 7, 5, 2
    n = 1 << log_n
    npolys = 8 + nw
    rng = po.SplitMix64(4300)
    ch, theta, theta2 = (rng.next_mod(r) for _ in range(3))
    omega = C.root_of_unity(log_n)
    evals, coeffs = [], []
    for p in range(npolys):
        e = cp.random_fr(curve, 4400 + p, n)
        evals.append(e)
        c = cp.ntt(curve, e.reshape(1, -1, 4), log_n, limbs(omega, 4), inverse=True)[0]
        coeffs.append([po.from_limbs(x) for x in c])
    srs = _srs(curve, alpha, n)
    polys = {0: coeffs[0:4], 1: coeffs[4:5 + nw], 2: [coeffs[5 + nw]], 3: coeffs[6 + nw:]}
    points = {0: [[ch], [ch], [ch, ch * omega % r], [ch, ch * omega % r]],
              1: [[ch, ch * (omega * omega if i & 1 else omega) % r] for i in range(nw + 1)],
              2: [[ch, ch * omega % r]], 3: [[ch], [ch]]}
    z, f, Lq = po.kzg_v2_proof_eval(r, polys, points, theta, theta2)
    G = C.g1
    def commit_bytes(c):
        return po.bls12_381_compress(1, G.mul(G.gen, po.poly_eval(c, alpha, r) % r))
    be = lambda v: int(v).to_bytes(32, "big")
    blobs = {b: b"".join(commit_bytes(c) for c in polys[b]) for b in range(4)}
    expected = [blobs[0], blobs[1], blobs[2], blobs[3]]            # the prover's transcript(commitments[batch])
    for b in range(4):                                             # update_transcript, batches ascending
        expected.append(blobs[b])
        expected += [be(v) for zl in z[b] for v in zl]
        for i in range(len(polys[b])):
            U = po.lagrange_interpolation(list(zip(points[b][i], z[b][i])), r)
            expected += [be(c) for c in U]
    expected += [commit_bytes(f), commit_bytes(Lq)]               # pi_1, pi_2
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 2)])
    out = np.zeros(8192, dtype=np.uint64)
    out_len = np.zeros(1, dtype=np.uint64)
    rc = shim.shim_placeholder_transcript_bls(P(srs), ctypes.c_size_t(n), P(np.concatenate(evals)), ctypes.c_size_t(npolys), ctypes.c_size_t(log_n),
                                              ctypes.c_size_t(nw), P(roots), P(limbs(ch, 4)), P(np.concatenate([limbs(theta, 4), limbs(theta2, 4)])), P(out),
                                              ctypes.c_size_t(len(out)), P(out_len))
    assert rc == 0
    words = out[: int(out_len[0])]
    got, at = [], 1
    for _ in range(int(words[0])):
        ln = int(words[at])
        nwords = (ln + 7) // 8
        got.append(words[at + 1: at + 1 + nwords].tobytes()[:ln])
        at += 1 + nwords
    assert len(got) == len(expected)
    for i, (g, e) in enumerate(zip(got, expected)):
        assert g == e, i


@pytest.mark.parametrize("curve,log_n", [(0, 8), (1, 8), (0, 12)])
def test_placeholder_quotient_chain_shim(shim, curve, log_n):
    """placeholder's quotient chain on the device (hip/placeholder_quotient.hpp; prover.hpp:220-277, 314-317, gates_argument.hpp:
    203-216) against the oracle's restatement (itself pinned to the reference's split_polynomial test and to dense polynomial
    arithmetic, tests/test_oracle_kat.py): gate product over the 4x extended domain with a rotated column, a second part on the 2x
    domain, F / (X^n - 1), the split into parts, and the QUOTIENT batch committed from device-resident parts = part(alpha) G."""
    C = CURVES[curve]
    r, alpha = C.r, 7
    n = 1 << log_n
    rng = po.SplitMix64(5100 + curve + log_n)
    cols = [[rng.next_mod(r) for _ in range(n)] for _ in range(3)]
    q = [1 + rng.next_mod(r - 1) if i % 2 == 0 else 0 for i in range(n)]
    cols[0] = [0 if q[i] else cols[0][i] for i in range(n)]
    w3 = [cols[1][i] * cols[2][i] % r for i in range(n)]
    mask = [1] * n
    theta, a0, a1 = (rng.next_mod(r) for _ in range(3))
    root = C.root_of_unity
    G = po.gate_argument_dfs([(theta, [(q, 0), (cols[0], 0), (cols[1], 1), (cols[2], 0)])], mask, 4 * n, root, r)
    e1, e2, e3 = (po.dfs_resize(x, 2 * n, root, r) for x in (cols[1], cols[2], w3))
    F1 = [(x * y - z) % r for x, y, z in zip(e1, e2, e3)]
    T = po.quotient_polynomial([G, F1], [a0, a1], n, root, r)
    parts = po.quotient_polynomial_split_dfs(T, n, 4, n, root, r)
    srs = _srs(curve, alpha, n)
    L1 = srs.shape[1]
    evals = fr_arr(cols[0] + cols[1] + cols[2] + w3 + q + mask)
    roots = np.stack([limbs(root(l), 4) for l in range(log_n + 3)])
    out_T = np.zeros((3 * n, 4), dtype=np.uint64)
    out_parts = np.zeros((4, n, 4), dtype=np.uint64)
    out_commits = np.zeros((4, L1), dtype=np.uint64)
    rc = shim.shim_placeholder_quotient(curve, P(srs), ctypes.c_size_t(n), P(evals), ctypes.c_size_t(log_n), P(roots), P(limbs(theta, 4)),
                                        P(np.concatenate([limbs(a0, 4), limbs(a1, 4)])), P(out_T), P(out_parts), P(out_commits))
    assert rc == 0
    assert fr_ints(out_T) == T
    for k in range(4):
        assert fr_ints(out_parts[k]) == parts[k], k
        chunk = T[k * n:(k + 1) * n]
        assert (out_commits[k] == cp.batch_mul(curve, 1, fr_arr([po.poly_eval(chunk, alpha, r)]))[0][0]).all(), k


@pytest.mark.parametrize("curve,world", [(0, 1), (1, 1), (0, 2), (1, 4), (0, 7)])
def test_kzg_v2_proof_eval_shim(shim, curve, world):
    """kzg_commitment_scheme_v2::commit + proof_eval (kzg_v2.hpp:208-305) through the shim class against the oracle's
    restatement: evaluations z, both quotient commitments, and the verifier's equation in the exponent (alpha known).
    world > 1: additionally the scheme over a DEVICE GROUP of that many members (kzg_params_group_hip: the key replicated, the columns of
    commit(batch) dealt, the coefficient forms gathered on member 0) must give the same commitments and the same opening proof."""
    import torch
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        _kzg_v2_proof_eval_shim(shim, curve)
    finally:
        shim.shim_set_world(1)


def _kzg_v2_proof_eval_shim(shim, curve):
    C = CURVES[curve]
    r = C.r
    alpha = 7  # placeholder.cpp:175
    rng = po.SplitMix64(77 + curve)
    x1, x2, x3 = (rng.next_mod(r) for _ in range(3))
    # batch 0: three polynomials of 64 evaluations opened at {x1, x2}; batch 2: sizes 64 and 128 with ragged point sets
    layout = [(0, 6, [x1, x2]), (0, 6, [x1, x2]), (0, 6, [x1, x2]), (2, 6, [x2]), (2, 7, [x1, x3])]
    npolys = len(layout)
    evals, coeffs = [], []
    for p, (_, log_n, _) in enumerate(layout):
        e = cp.random_fr(curve, 500 + p, 1 << log_n)
        evals.append(e)
        c = cp.ntt(curve, e.reshape(1, -1, 4), log_n, limbs(C.root_of_unity(log_n), 4), inverse=True)[0]
        coeffs.append([po.from_limbs(x) for x in c])
    n_srs = 128
    srs = _srs(curve, alpha, n_srs)
    theta, theta2 = rng.next_mod(r), rng.next_mod(r)
    polys, points = {}, {}
    for p, (k, _, pts) in enumerate(layout):
        polys.setdefault(k, []).append(coeffs[p])
        points.setdefault(k, []).append(pts)
    z, f, L = po.kzg_v2_proof_eval(r, polys, points, theta, theta2)

    u64 = lambda v: np.array(v, dtype=np.uint64)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(8)])
    allpts = fr_arr([x for _, _, pts in layout for x in pts])
    commits = np.zeros((npolys, srs.shape[1]), dtype=np.uint64)
    zvals = np.zeros((len(allpts), 4), dtype=np.uint64)
    pi = np.zeros((2, srs.shape[1]), dtype=np.uint64)
    absorbed = np.zeros(2, dtype=np.uint64)
    rc = shim.shim_kzg_v2_proof_eval(curve, P(srs), ctypes.c_size_t(n_srs), ctypes.c_size_t(npolys), P(u64([k for k, _, _ in layout])),
                                     P(u64([l for _, l, _ in layout])), P(np.concatenate(evals)), P(u64([len(p) for _, _, p in layout])), P(allpts),
                                     P(roots), P(limbs(theta, 4)), P(limbs(theta2, 4)), P(commits), P(zvals), P(pi), P(absorbed))
    assert rc == 0
    g = lambda v: cp.batch_mul(curve, 1, fr_arr([v % r]))[0][0]
    # commitments: f_i(alpha) G
    for p in range(npolys):
        assert (commits[p] == g(po.poly_eval(coeffs[p], alpha, r))).all(), p
    # evaluations in the reference's order (batch, polynomial, point)
    exp_z = [v for k in sorted(z) for zl in z[k] for v in zl]
    assert [po.from_limbs(x) for x in zvals] == exp_z
    # pi_1, pi_2 against the oracle's quotients, once through the MSM oracle and once in the exponent
    e1, i1 = cp.msm(curve, 1, srs[: len(f)], fr_arr(f), chunks=2)
    assert i1 == 0 and (pi[0] == e1).all()
    assert (pi[0] == g(po.poly_eval(f, alpha, r))).all() and (pi[1] == g(po.poly_eval(L, alpha, r))).all()
    # the verifier's equation (kzg_v2.hpp verify_eval) with alpha in the clear:
    #   sum_i theta^i Z_{T\S_i}(theta2) (f_i(alpha) - U_i(theta2)) - V(theta2) f(alpha) == (alpha - theta2) L(alpha)
    merged = sorted({x for _, _, pts in layout for x in pts})
    lhs, th = 0, 1
    for k in sorted(polys):
        for i, c in enumerate(polys[k]):
            U = po.lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            zts = po.poly_eval(po.vanishing_poly([x for x in merged if x not in points[k][i]], r), theta2, r)
            lhs = (lhs + th * zts * (po.poly_eval(c, alpha, r) - po.poly_eval(U, theta2, r))) % r
            th = th * theta % r
    lhs = (lhs - po.poly_eval(po.vanishing_poly(merged, r), theta2, r) * po.poly_eval(f, alpha, r)) % r
    assert lhs == (alpha - theta2) * po.poly_eval(L, alpha, r) % r
    # transcript traffic: 5 commitments + pi_1 + pi_2; 8 evaluations + 8 U coefficients
    assert list(absorbed) == [npolys + 2, 2 * len(allpts)]


def _extend(curve, evals, log_n, log_big):
    """polynomial_dfs::resize on the oracle: coefficients, zero-extend, evaluate on the larger domain"""
    C = CURVES[curve]
    c = cp.ntt(curve, evals.reshape(1, -1, 4), log_n, limbs(C.root_of_unity(log_n), 4), inverse=True)[0]
    big = np.zeros((1 << log_big, 4), dtype=np.uint64)
    big[: 1 << log_n] = c
    return cp.ntt(curve, big.reshape(1, -1, 4), log_big, limbs(C.root_of_unity(log_big), 4))[0]


@pytest.mark.parametrize("curve,log_domain,fri_step", [(0, 8, 1), (0, 8, 3), (1, 7, 2)])
def test_precommit_leaves_shim(shim, curve, log_domain, fri_step):
    """precommit<FRI> up to the Merkle tree (basic_fri.hpp:433-496): resize to D, then the coset-ordered leaves.
    The expected layout replays the reference's s_indices loop (:456-492) on the oracle's extended polynomials."""
    C = CURVES[curve]
    logs = [log_domain - 2, log_domain - 2, log_domain, log_domain - 1]
    evals = [cp.random_fr(curve, 900 + i, 1 << l) for i, l in enumerate(logs)]
    ext = [e if l == log_domain else _extend(curve, e, l, log_domain) for e, l in zip(evals, logs)]
    D, m = 1 << log_domain, 2
    coset = 1 << fri_step
    exp = []
    for x in range(D // coset):
        for f in ext:
            s = [[0, 0] for _ in range(coset // m)]
            s[0] = [x, (x + D // 2) % D]
            exp += [f[s[0][0]], f[s[0][1]]]
            base, prev_half, i = D // (m * m), 1, 1
            while i < coset // m:
                for j in range(prev_half):
                    s[i][0] = (base + s[j][0]) % D
                    s[i][1] = (s[i][0] + D // 2) % D
                    exp += [f[s[i][0]], f[s[i][1]]]
                    i += 1
                base //= m
                prev_half <<= 1
    exp = np.array(exp, dtype=np.uint64)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_domain + 1)])
    out = np.zeros((len(logs) * D, 4), dtype=np.uint64)
    rc = shim.shim_precommit_leaves(curve, P(np.concatenate(evals)), ctypes.c_size_t(len(logs)), P(np.array(logs, dtype=np.uint64)),
                                    ctypes.c_size_t(log_domain), ctypes.c_size_t(fri_step), P(roots), P(out))
    assert rc == 0
    assert (out == exp).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_polynomial_dfs_ops_shim(shim, curve):
    """device polynomial_dfs: resize, *=, coefficients / from_coefficients, +=, -=, and fold_polynomial of the product"""
    C = CURVES[curve]
    r = C.r
    log_n, log_big = 6, 8
    a, b = cp.random_fr(curve, 950, 1 << log_n), cp.random_fr(curve, 951, 1 << log_n)
    ea, eb = _extend(curve, a, log_n, log_big), _extend(curve, b, log_n, log_big)
    prod = [x * y % r for x, y in zip((po.from_limbs(v) for v in ea), (po.from_limbs(v) for v in eb))]
    alpha = po.SplitMix64(12).next_mod(r)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_big + 1)])
    big = 1 << log_big
    o_prod, o_round, o_as = (np.zeros((big, 4), dtype=np.uint64) for _ in range(3))
    o_fold = np.zeros((big // 2, 4), dtype=np.uint64)
    rc = shim.shim_dfs_ops(curve, P(a), P(b), ctypes.c_size_t(log_n), ctypes.c_size_t(log_big), P(roots), P(limbs(alpha, 4)), P(o_prod), P(o_round),
                           P(o_as), P(o_fold))
    assert rc == 0
    assert [po.from_limbs(v) for v in o_prod] == prod
    assert (o_round == o_prod).all() and (o_as == o_prod).all()
    winv, inv2, half = pow(C.root_of_unity(log_big), -1, r), pow(2, -1, r), big // 2
    assert [po.from_limbs(v) for v in o_fold] == [
        inv2 * ((1 + alpha * pow(winv, i, r)) * prod[i] + (1 - alpha * pow(winv, i, r)) * prod[half + i]) % r for i in range(half)]


@pytest.mark.parametrize("curve", [0, 1])
def test_kc_multiexp_shim(shim, curve):
    """kc_multiexp_with_mixed_addition (knowledge_commitment_multiexp.hpp:57-108): sparse (G2, G1) pairs, index window,
    zero / one scalars, against the oracle's restatement."""
    C = CURVES[curve]
    domain = 60
    indices = [0, 1, 3, 4, 7, 10, 11, 20, 21, 22, 35, 36, 50, 59]
    ks = cp.random_fr(curve, 970, len(indices))
    g_pts, _ = cp.batch_mul(curve, 2, ks)
    h_pts, _ = cp.batch_mul(curve, 1, ks)
    from util import pt_from_limbs
    values = [(pt_from_limbs(curve, 2, g_pts[i]), pt_from_limbs(curve, 1, h_pts[i])) for i in range(len(indices))]
    for min_idx, max_idx in ((0, 60), (3, 36), (5, 6), (11, 51)):
        n = max_idx - min_idx
        sc = [po.from_limbs(v) for v in cp.random_fr(curve, 980 + min_idx, n)]
        for j in range(0, n, 3):
            sc[j] = 0
        for j in range(1, n, 5):
            sc[j] = 1
        eg, eh = po.kc_multiexp(C.g2, C.g1, indices, values, min_idx, max_idx, sc)
        og, oh = np.zeros(g_pts.shape[1], dtype=np.uint64), np.zeros(h_pts.shape[1], dtype=np.uint64)
        oinf = np.zeros(2, dtype=np.uint8)
        rc = shim.shim_kc_multiexp(curve, P(g_pts), P(h_pts), P(np.array(indices, dtype=np.uint64)), ctypes.c_size_t(len(indices)),
                                   ctypes.c_size_t(domain), ctypes.c_size_t(min_idx), ctypes.c_size_t(max_idx), P(fr_arr(sc)), ctypes.c_size_t(n),
                                   P(og), P(oh), P(oinf))
        assert rc == 0
        assert pt_from_limbs(curve, 2, og, oinf[0]) == eg and pt_from_limbs(curve, 1, oh, oinf[1]) == eh, (min_idx, max_idx)


@pytest.mark.parametrize("curve,log_m", [(0, 5), (1, 4)])
def test_powers_of_tau_lagrange_shim(shim, curve, log_m):
    """evaluate_all_lagrange_polynomials over group elements (powers_of_tau/result.hpp:81-94): [tau^i] G -> [L_j(tau)] G"""
    C = CURVES[curve]
    r, m = C.r, 1 << log_m
    w, tau = C.root_of_unity(log_m), po.SplitMix64(55).next_mod(C.r)
    powers, _ = cp.batch_mul(curve, 1, fr_arr([pow(tau, i, r) for i in range(m)]))
    out = np.zeros_like(powers)
    oinf = np.zeros(m, dtype=np.uint8)
    assert shim.shim_lagrange_g1(curve, P(powers), ctypes.c_size_t(m), P(limbs(w, 4)), P(out), P(oinf)) == 0
    exp, einf = cp.batch_mul(curve, 1, fr_arr(po.lagrange_at(m, w, tau, r)))
    assert (oinf == einf).all() and (out == exp).all()


def _key_blob(g, C):
    """serialise the oracle's proving key in the reference's wire format (g16/marshalling.hpp:203-492, 656-760):
    4-byte big-endian counts, compressed points (po.bls12_381_compress, pinned to AGG:932-1010), little-endian Fr"""
    u32 = lambda v: int(v).to_bytes(4, "big")
    aq, ainf = g.query(0)
    bh, bhinf = g.query(1)
    bg, _ = g.query(2)
    hq, hinf = g.query(3)
    lq, linf = g.query(4)
    f1, _ = g.query(5)
    f2, _ = g.query(6)
    pt = lambda grp, arr, inf=0: po.bls12_381_compress(grp, pt_from_limbs_(grp, arr, inf))
    out = pt(1, f1[0]) + pt(1, f1[1]) + pt(2, f2[0]) + pt(1, f1[2]) + pt(2, f2[1])           # alpha_g1 beta_g1 beta_g2 delta_g1 delta_g2
    out += u32(len(aq)) + b"".join(pt(1, aq[i], ainf[i]) for i in range(len(aq)))
    idx = [i for i in range(len(bh)) if not bhinf[i]]                                          # sparse over the non-zero B_i(t)
    body = u32(len(idx)) + b"".join(u32(i) for i in idx) + b"".join(pt(2, bg[i]) + pt(1, bh[i]) for i in idx) + u32(len(bh))
    out += u32(len(body)) + body
    out += u32(g.m - 1) + b"".join(pt(1, hq[i], hinf[i]) for i in range(g.m - 1))
    out += u32(len(lq)) + b"".join(pt(1, lq[i], linf[i]) for i in range(len(lq)))
    out += u32(g.n) + u32(g.N - g.n) + u32(g.M)
    csr = [g.csr(k) for k in range(3)]
    for row in range(g.M):
        c = b""
        for rp, cl, cf in csr:
            lo, hi = int(rp[row]), int(rp[row + 1])
            c += u32(hi - lo) + b"".join(u32(cl[j]) + po.from_limbs(cf[j]).to_bytes(32, "little") for j in range(lo, hi))
        out += u32(len(c)) + c
    return out


def pt_from_limbs_(grp, arr, inf):
    from util import pt_from_limbs
    return pt_from_limbs(0, grp, arr, inf)


def test_groth16_from_serialised_key(shim):
    """the proving key arrives in the reference's wire format, its point blobs are decoded on the device,
    and the proof equals the oracle's."""
    C = CURVES[0]
    M, n = 100, 10
    g = cp.Groth16(0, M, n, seed=1)
    w = limbs(C.root_of_unity(g.log_m), 4)
    gen = limbs(C.fr_generator, 4)
    rng = po.SplitMix64(2024)
    trap = fr_arr([rng.next_mod(C.r) for _ in range(5)])
    r_, s_ = limbs(rng.next_mod(C.r), 4), limbs(rng.next_mod(C.r), 4)
    g.keygen(trap, w)
    expected = g.prove(r_, s_, w, gen, chunks=4)
    blob = np.frombuffer(_key_blob(g, C), dtype=np.uint8).copy()
    proof = np.zeros_like(expected)
    rc = shim.shim_groth16_prove_from_bytes(P(blob), ctypes.c_size_t(len(blob)), P(g.assignment()), ctypes.c_size_t(g.n), ctypes.c_size_t(g.N), P(w),
                                            P(gen), P(r_), P(s_), P(proof))
    assert rc == 0
    assert (proof == expected).all()
    # a truncated blob is reported, not read past
    assert shim.shim_groth16_prove_from_bytes(P(blob), ctypes.c_size_t(len(blob) - 7), P(g.assignment()), ctypes.c_size_t(g.n),
                                              ctypes.c_size_t(g.N), P(w), P(gen), P(r_), P(s_), P(proof)) == -1


@pytest.mark.parametrize("curve", [0, 1])
def test_polynomial_product_shift_shim(shim, curve):
    """math::polynomial_product (k factors, each resized to the product's domain, one k-way pointwise pass),
    math::polynomial_shift (rotation of the evaluation vector, also over an extension of the shift's domain) and
    polynomial_dfs::resize to a smaller domain -- against the oracle's coefficient-form arithmetic."""
    C = CURVES[curve]
    r = C.r
    logs, degs = [4, 5, 4], [9, 31, 15]
    root = lambda l: C.root_of_unity(l)
    coeffs = [[po.SplitMix64(70 + k).next_mod(r) for _ in range(d + 1)] for k, d in enumerate(degs)]
    evals = [po.ntt(c + [0] * ((1 << l) - len(c)), root(l), r) for c, l in zip(coeffs, logs)]
    prod_c = [1]
    for c in coeffs:
        prod_c = po.poly_mul(prod_c, c, r)
    size = 1
    while size < sum(degs) + 1:
        size <<= 1
    prod_e = po.ntt(prod_c + [0] * (size - len(prod_c)), root(size.bit_length() - 1), r)
    roots = np.stack([limbs(root(l), 4) for l in range(10)])
    o_prod, o_shift = np.zeros((size, 4), dtype=np.uint64), np.zeros((size, 4), dtype=np.uint64)
    o_size = np.zeros(2, dtype=np.uint64)
    n0 = 1 << logs[0]
    o_small = np.zeros((2 * n0, 4), dtype=np.uint64)
    for shift, dom in ((1, 0), (-3, 0), (1, 16), (-1, 8)):
        rc = shim.shim_dfs_product_shift(curve, P(fr_arr(sum(evals, []))), ctypes.c_size_t(3), P(np.array(logs, dtype=np.uint64)),
                                         P(np.array(degs, dtype=np.uint64)), P(roots), ctypes.c_int64(shift), ctypes.c_size_t(dom), P(o_prod), P(o_size),
                                         P(o_shift), P(o_small))
        assert rc == 0
        assert list(o_size) == [size, sum(degs)]
        assert fr_ints(o_prod) == prod_e
        assert fr_ints(o_shift) == po.polynomial_shift(prod_e, shift, dom)
        assert fr_ints(o_small[:n0]) == evals[0] and fr_ints(o_small[n0:]) == evals[0]  # factors untouched; grow-then-shrink round trip
    # the shift really is f(omega^shift X): check one case in coefficient form
    w = root(size.bit_length() - 1)
    assert po.polynomial_shift(prod_e, 1) == po.ntt([c * pow(w, i, r) % r for i, c in enumerate(prod_c + [0] * (size - len(prod_c)))], w, r)


def _toy_root(r):
    return lambda leaves, per_leaf: (per_leaf + sum((i + 1) * v for i, v in enumerate(leaves))) % r


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("builder", ["vector", "span", "streaming"])
@pytest.mark.parametrize("curve,log_domain,steps", [(0, 8, [1, 2]), (1, 7, [2, 1, 1]), (0, 7, [3])])
def test_lpc_scheme_shim(shim, curve, log_domain, steps, builder, world):
    """lpc_commitment_scheme_hip driven through the consumer contract placeholder has with its commitment scheme (fixed batch,
    preprocess / setup, two batches, ragged point sets): commit roots, evaluations, FRI round roots and the final polynomial
    against po.lpc_proof_eval (lpc.hpp:101-200 + basic_fri.hpp:433-496, 705-742); the Merkle tree is a toy functor on both sides.
    builder: the three shapes of the caller's tree builder -- a std::vector of leaves, a span over page-locked memory, and slices
    of whole leaves absorbed while the next slice is in flight (64-element slices, one polynomial per upload chunk, the second
    batch LENT to the scheme instead of copied).
    world > 1: the scheme over a DEVICE GROUP of that many members (hip/lpc.hpp commit_group: the polynomials dealt, the leaves cut by range
    over the leaf owners -- the first 1, 2 or 4 members --, the coefficient forms gathered on member 0): the same roots, evaluations, rounds.
    Two polynomials per batch: at world 3 and 8 some members hold leaves but no polynomial."""
    import torch
    shim.shim_set_lpc_builder({"vector": 0, "span": 1, "streaming": 2}[builder])
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        _lpc_scheme_shim(shim, curve, log_domain, steps)
    finally:
        shim.shim_set_lpc_builder(0)
        shim.shim_set_world(1)


def _lpc_scheme_shim(shim, curve, log_domain, steps):
    C = CURVES[curve]
    r = C.r
    logs = [log_domain - 3, log_domain - 3, log_domain - 2, log_domain - 3]
    evals = [fr_ints(cp.random_fr(curve, 1300 + i, 1 << l)) for i, l in enumerate(logs)]
    rng = po.SplitMix64(55 + curve)
    p0, p1, p2 = (rng.next_mod(r) for _ in range(3))
    etha, theta = rng.next_mod(r), rng.next_mod(r)
    alphas = [rng.next_mod(r) for _ in range(sum(steps))]
    challenges = [etha, etha, theta] + alphas
    batches = {0: evals[:2], 1: evals[2:]}
    points = {0: [[p0], [p0, p2]], 1: [[p0, p1], [p0]]}
    e_roots, e_z, e_fri, e_final = po.lpc_proof_eval(r, batches, points, [0], log_domain, steps, C.root_of_unity, challenges, _toy_root(r))
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_domain + 1)])
    o_roots, o_z, o_fri = np.zeros((2, 4), dtype=np.uint64), np.zeros((6, 4), dtype=np.uint64), np.zeros((len(steps), 4), dtype=np.uint64)
    nfinal = 1 << (log_domain - sum(steps))
    o_final, o_counts = np.zeros((nfinal, 4), dtype=np.uint64), np.zeros(6, dtype=np.uint64)
    rc = shim.shim_lpc_scheme(curve, P(fr_arr(sum(evals, []))), ctypes.c_size_t(4), P(np.array(logs, dtype=np.uint64)), ctypes.c_size_t(log_domain),
                              P(np.array(steps, dtype=np.uint64)), ctypes.c_size_t(len(steps)), P(roots), P(fr_arr([p0, p1, p2])), P(fr_arr(challenges)),
                              ctypes.c_size_t(len(challenges)), P(o_roots), P(o_z), P(o_fri), P(o_final), P(o_counts))
    assert rc == 0
    assert fr_ints(o_roots) == [e_roots[0], e_roots[1]]
    assert fr_ints(o_z) == [v for k in (0, 1) for pl in e_z[k] for v in pl]
    assert fr_ints(o_fri) == e_fri
    assert fr_ints(o_final) == (e_final + [0] * nfinal)[:nfinal]
    # 6 evaluations, one root per step, all challenges drawn, 2 batch roots + the round roots absorbed, and what the caller's
    # query phase needs is kept (round trees, alphas, batch trees)
    assert list(o_counts) == [6, len(steps), nfinal, len(challenges), 2 + len(steps), len(steps) + 100 * sum(steps) + 10000 * 2]


def test_precommit_leaves_full_size_16_columns(shim):
    """BASELINE cfg 5's LPC shape AT ITS SIZE: 16 polynomial_dfs of 2^20 rows extended to D[0] = 2^21 and laid out as coset-ordered leaves
    (precommit<FRI>, basic_fri.hpp:433-496) -- 2^20 leaves of 32 elements, 1.07 GB -- bit-exact against the oracle (cport: inverse
    transform, zero padding, forward transform over 2^21 points, then fri_leaves' index computation, pinned to pyoracle's replay of the
    reference's s_indices loop at <= 2^8).  cfg 2 and cfg 3 are checked this way at their full sizes; this is the LPC row's turn."""
    curve, log_n, cols, fri_step = 0, 20, 16, 1
    C = CURVES[curve]
    evals = cp.random_fr(curve, 4100, cols << log_n).reshape(cols, 1 << log_n, 4)
    coeffs = cp.ntt(curve, evals, log_n, limbs(C.root_of_unity(log_n), 4), inverse=True)
    ext = np.zeros((cols, 2 << log_n, 4), dtype=np.uint64)
    ext[:, : 1 << log_n] = coeffs
    del coeffs
    ext = cp.ntt(curve, ext, log_n + 1, limbs(C.root_of_unity(log_n + 1), 4))
    want = cp.fri_leaves(list(ext), fri_step)
    del ext
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 2)])
    out = np.zeros((cols * (2 << log_n), 4), dtype=np.uint64)
    rc = shim.shim_precommit_leaves(curve, P(evals.reshape(-1, 4)), ctypes.c_size_t(cols), P(np.full(cols, log_n, dtype=np.uint64)), ctypes.c_size_t(log_n + 1),
                                    ctypes.c_size_t(fri_step), P(roots), P(out))
    assert rc == 0
    assert out.shape == want.shape and np.array_equal(out, want)
    # the same leaves through the SCHEME's commit over a device group of four members (16 polynomials dealt four each, four leaf owners,
    # slices of 2^21 elements absorbed in order) -- and over one context, for the record
    import torch
    for world in (4, 1):
        out[:] = 0
        shim.shim_set_world(world)
        shim.shim_set_gpus(max(1, torch.cuda.device_count()))
        try:
            rc = shim.shim_lpc_commit_leaves(curve, P(evals.reshape(-1, 4)), ctypes.c_size_t(cols), P(np.full(cols, log_n, dtype=np.uint64)), ctypes.c_size_t(log_n + 1),
                                             ctypes.c_size_t(fri_step), ctypes.c_size_t(1 << 21), P(out))
        finally:
            shim.shim_set_world(1)
        assert rc == 0, world
        assert np.array_equal(out, want), world


@pytest.mark.parametrize("curve,logs,log_domain,fri_step,world,slice_elems", [
    (0, [6, 6, 7, 7, 7], 9, 1, 2, 40), (0, [6, 6, 7, 7, 7], 9, 3, 4, 1 << 20), (1, [8] * 7, 10, 2, 3, 7 * 4 * 5), (1, [5], 8, 4, 8, 16),
    (0, [10] * 9 + [11] * 4, 12, 1, 8, 13 * 2 * 100), (0, [4, 4, 4], 4, 4, 4, 1), (1, [13] * 6, 15, 2, 5, 1 << 14), (0, [7, 7], 7, 1, 1, 64)])
def test_lpc_commit_leaves_over_group(shim, curve, logs, log_domain, fri_step, world, slice_elems):
    """The leaves lpc_commitment_scheme_hip::commit hands to a streaming tree builder -- over a device group of `world` members (hip/lpc.hpp
    commit_group: polynomials dealt, segments packed and pushed to the leaf owners, each owner's range laid out by the ordinary leaf kernel on its
    compact domain) -- against the oracle's precommit leaves (inverse transform, zero padding, forward transform on D[0], fri_leaves): ragged
    batches (runs of two sizes), more members than polynomials, a single leaf (log_domain == fri_step: one owner), no extension (size == D),
    slices that cut the owners' ranges unevenly.  Twice per scheme: the second batch reuses every kept buffer."""
    import torch
    C = CURVES[curve]
    D = 1 << log_domain
    evals = [cp.random_fr(curve, 5200 + i + log_domain, 1 << l) for i, l in enumerate(logs)]
    ext = []
    for e, l in zip(evals, logs):
        c = cp.ntt(curve, e.reshape(1, -1, 4), l, limbs(C.root_of_unity(l), 4), inverse=True)[0]
        big = np.zeros((1, D, 4), dtype=np.uint64)
        big[0, : 1 << l] = c
        ext.append(cp.ntt(curve, big, log_domain, limbs(C.root_of_unity(log_domain), 4))[0])
    want = cp.fri_leaves(ext, fri_step)
    out = np.zeros((len(logs) * D, 4), dtype=np.uint64)
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        rc = shim.shim_lpc_commit_leaves(curve, P(np.concatenate(evals)), ctypes.c_size_t(len(logs)), P(np.array(logs, dtype=np.uint64)), ctypes.c_size_t(log_domain),
                                         ctypes.c_size_t(fri_step), ctypes.c_size_t(slice_elems), P(out))
    finally:
        shim.shim_set_world(1)
    assert rc == 0
    assert np.array_equal(out, want)


@pytest.mark.parametrize("world", [1, 4])
@pytest.mark.parametrize("builder", ["vector", "span", "streaming"])
@pytest.mark.parametrize("curve,log_domain,steps", [(0, 15, [3, 2, 2]), (1, 15, [2, 3, 1]), (0, 19, [3, 3, 2]), (1, 19, [4, 2, 2])])
def test_lpc_scheme_at_multipass_sizes(shim, curve, log_domain, steps, builder, world):
    """the LPC scheme at sizes where every transform is multi-pass, uploads are chunked and the streaming leaf builder
    wraps -- polynomials of 2^12 / 2^13 evaluations on a 2^15-point domain and of 2^16 / 2^17 on a 2^19-point one, both curves, all three
    tree-builder shapes -- against the C++ oracle's restatement of lpc.hpp:101-200 + basic_fri.hpp:433-496, 705-742 (cport.lpc_proof_eval,
    pinned to pyoracle at <= 2^8): batch roots over the coset-ordered leaves, every evaluation, every FRI round root, the final polynomial.
    world 4: the same through the scheme over a device group of four members."""
    import torch
    C = CURVES[curve]
    r = C.r
    logs = [log_domain - 3, log_domain - 3, log_domain - 2, log_domain - 3]
    evals = [cp.random_fr(curve, 2300 + i + log_domain, 1 << l) for i, l in enumerate(logs)]
    rng = po.SplitMix64(155 + curve + log_domain)
    p0, p1, p2 = (rng.next_mod(r) for _ in range(3))
    etha, theta = rng.next_mod(r), rng.next_mod(r)
    challenges = [etha, etha, theta] + [rng.next_mod(r) for _ in range(sum(steps))]
    points = {0: [[p0], [p0, p2]], 1: [[p0, p1], [p0]]}
    e_roots, e_z, e_fri, e_final = cp.lpc_proof_eval(curve, {0: evals[:2], 1: evals[2:]}, points, [0], log_domain, steps, challenges, cp.toy_root(curve))
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_domain + 1)])
    o_roots, o_z, o_fri = np.zeros((2, 4), dtype=np.uint64), np.zeros((6, 4), dtype=np.uint64), np.zeros((len(steps), 4), dtype=np.uint64)
    nfinal = 1 << (log_domain - sum(steps))
    o_final, o_counts = np.zeros((nfinal, 4), dtype=np.uint64), np.zeros(6, dtype=np.uint64)
    shim.shim_set_lpc_builder({"vector": 0, "span": 1, "streaming": 2}[builder])
    shim.shim_set_lpc_slice(ctypes.c_size_t(3 << (log_domain - 4)))   # slices of whole leaves that do not divide the leaf count evenly
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        rc = shim.shim_lpc_scheme(curve, P(np.concatenate(evals)), ctypes.c_size_t(4), P(np.array(logs, dtype=np.uint64)), ctypes.c_size_t(log_domain),
                                  P(np.array(steps, dtype=np.uint64)), ctypes.c_size_t(len(steps)), P(roots), P(fr_arr([p0, p1, p2])), P(fr_arr(challenges)),
                                  ctypes.c_size_t(len(challenges)), P(o_roots), P(o_z), P(o_fri), P(o_final), P(o_counts))
    finally:
        shim.shim_set_lpc_builder(0)
        shim.shim_set_lpc_slice(ctypes.c_size_t(64))
        shim.shim_set_world(1)
    assert rc == 0
    assert fr_ints(o_roots) == [e_roots[0], e_roots[1]]
    assert fr_ints(o_z) == [v for k in (0, 1) for pl in e_z[k] for v in pl]
    assert fr_ints(o_fri) == e_fri
    assert (o_final == cp._pad(e_final, nfinal)[:nfinal]).all()
    assert list(o_counts) == [6, len(steps), nfinal, len(challenges), 2 + len(steps), len(steps) + 100 * sum(steps) + 10000 * 2]


def _kzg_layout_at(curve, log_n, layout=None, seed=0):
    """five polynomials in two batches, sizes 2^log_n and 2^(log_n + 1), ragged point sets (the shape of test_kzg_v2_proof_eval_shim) -- or the
    layout given: [(batch id ascending, log2 size, points)]"""
    C = CURVES[curve]
    r = C.r
    rng = po.SplitMix64(177 + curve + log_n + seed)
    x1, x2, x3 = (rng.next_mod(r) for _ in range(3))
    if layout is None:
        layout = [(0, log_n, [x1, x2]), (0, log_n, [x1, x2]), (0, log_n, [x1, x2]), (2, log_n, [x2]), (2, log_n + 1, [x1, x3])]
    evals, polys, points = [], {}, {}
    for p, (k, l, pts) in enumerate(layout):
        e = cp.random_fr(curve, 2500 + p + log_n + 7 * seed, 1 << l)
        evals.append(e)
        polys.setdefault(k, []).append(cp.ntt(curve, e.reshape(1, -1, 4), l, limbs(C.root_of_unity(l), 4), inverse=True)[0])
        points.setdefault(k, []).append(pts)
    return layout, evals, polys, points, rng.next_mod(r), rng.next_mod(r)


@pytest.mark.parametrize("curve,log_n,world", [(0, 12, 1), (1, 12, 4), (0, 16, 3), (1, 16, 1)])
def test_kzg_proof_eval_at_multipass_sizes(shim, curve, log_n, world):
    """both batched KZG opening proofs (kzg_v2.hpp:236-305, kzg.hpp:782-807) through the shim classes with polynomials
    of 2^12 / 2^13 and 2^16 / 2^17 evaluations, both curves, against the C++ oracle's restatements (cport.kzg_v2_proof_eval /
    kzg_v1_proof_eval, pinned to pyoracle at <= 2^7): every evaluation, and the quotient commitments pi_1, pi_2 / kzg_proof equal to the
    oracle's MSM of the oracle's quotient polynomials over the same SRS (and to quotient(alpha) G in the exponent).
    world > 1: both schemes additionally over a device group of that many members (inside the harness: commitments, evaluations and the
    quotient commitments -- cut by point range over the members -- equal to the single-device scheme's)."""
    import torch
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        _kzg_proof_eval_at_multipass_sizes(shim, curve, log_n)
    finally:
        shim.shim_set_world(1)


def _kzg_proof_eval_at_multipass_sizes(shim, curve, log_n, layout=None, seed=0):
    C = CURVES[curve]
    r, alpha = C.r, 7
    layout, evals, polys, points, theta, theta2 = _kzg_layout_at(curve, log_n, layout, seed)
    npolys = len(layout)
    n_srs = 2 << log_n
    srs = _srs(curve, alpha, n_srs)
    u64 = lambda v: np.array(v, dtype=np.uint64)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 2)])
    allpts = fr_arr([x for _, _, pts in layout for x in pts])
    g = lambda v: cp.batch_mul(curve, 1, fr_arr([v % r]))[0][0]
    z, f, L = cp.kzg_v2_proof_eval(curve, polys, points, theta, theta2)
    commits = np.zeros((npolys, srs.shape[1]), dtype=np.uint64)
    zvals = np.zeros((len(allpts), 4), dtype=np.uint64)
    pi = np.zeros((2, srs.shape[1]), dtype=np.uint64)
    absorbed = np.zeros(2, dtype=np.uint64)
    rc = shim.shim_kzg_v2_proof_eval(curve, P(srs), ctypes.c_size_t(n_srs), ctypes.c_size_t(npolys), P(u64([k for k, _, _ in layout])),
                                     P(u64([l for _, l, _ in layout])), P(np.concatenate(evals)), P(u64([len(p) for _, _, p in layout])), P(allpts),
                                     P(roots), P(limbs(theta, 4)), P(limbs(theta2, 4)), P(commits), P(zvals), P(pi), P(absorbed))
    assert rc == 0
    flat = [c for k in sorted(polys) for c in polys[k]]
    for p in range(npolys):
        assert (commits[p] == g(cp.poly_eval(curve, flat[p], alpha))).all(), p
    assert [po.from_limbs(x) for x in zvals] == [v for k in sorted(z) for zl in z[k] for v in zl]
    e1, i1 = cp.msm(curve, 1, srs[: len(f)], f, chunks=cp.num_threads())
    e2, i2 = cp.msm(curve, 1, srs[: len(L)], L, chunks=cp.num_threads())
    assert i1 == 0 and i2 == 0 and (pi[0] == e1).all() and (pi[1] == e2).all()
    assert (pi[0] == g(cp.poly_eval(curve, f, alpha))).all() and (pi[1] == g(cp.poly_eval(curve, L, alpha))).all()
    # the first batched scheme: one quotient commitment
    z1, acc = cp.kzg_v1_proof_eval(curve, polys, points, theta)
    vk = cp.batch_mul(curve, 2, fr_arr([pow(alpha, i, r) for i in range(3)]))[0]
    proof = np.zeros(srs.shape[1], dtype=np.uint64)
    g2_out = np.zeros((2, vk.shape[1]), dtype=np.uint64)
    g2_poly = fr_arr([3, 5, 11])
    rc = shim.shim_kzg_v1_proof_eval(curve, P(srs), ctypes.c_size_t(n_srs), P(vk), ctypes.c_size_t(3), ctypes.c_size_t(npolys), P(u64([k for k, _, _ in layout])),
                                     P(u64([l for _, l, _ in layout])), P(np.concatenate(evals)), P(u64([len(p) for _, _, p in layout])), P(allpts), P(roots),
                                     P(limbs(theta, 4)), P(g2_poly), ctypes.c_size_t(3), P(commits), P(zvals), P(proof), P(g2_out), P(absorbed))
    assert rc == 0
    assert [po.from_limbs(x) for x in zvals] == [v for k in sorted(z1) for zl in z1[k] for v in zl]
    e3, i3 = cp.msm(curve, 1, srs[: len(acc)], acc, chunks=cp.num_threads())
    assert i3 == 0 and (proof == e3).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_kzg_placeholder_contract_shim(shim, curve):
    """kzg_commitment_scheme_v2_placeholder_hip under the same consumer: byte-blob commitments through the caller's packer
    (kzg_v2.hpp:208-226), verify_eval through the caller's hook (:312), and the opening proof equal to the oracle's."""
    C = CURVES[curve]
    r = C.r
    log_n, npolys, alpha = 5, 4, 7
    n = 1 << log_n
    srs = _srs(curve, alpha, n + 4)
    evals = [fr_ints(cp.random_fr(curve, 1400 + i, n)) for i in range(npolys)]
    rng = po.SplitMix64(91 + curve)
    p0, p1, p2 = (rng.next_mod(r) for _ in range(3))
    theta, theta2 = rng.next_mod(r), rng.next_mod(r)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 1)])
    blob, pi = np.zeros(2, dtype=np.uint64), np.zeros((2, srs.shape[1]), dtype=np.uint64)
    rc = shim.shim_kzg_placeholder_contract(curve, P(srs), ctypes.c_size_t(n + 4), P(fr_arr(sum(evals, []))), ctypes.c_size_t(npolys), ctypes.c_size_t(log_n),
                                            P(roots), P(fr_arr([p0, p1, p2])), P(fr_arr([theta, theta2])), ctypes.c_size_t(2), P(blob), P(pi))
    assert rc == 0
    point_bytes = srs.shape[1] * 8
    assert list(blob) == [2 * point_bytes, 2 * point_bytes]  # two polynomials per batch, one packed point each
    w = C.root_of_unity(log_n)
    polys = {0: [po.intt(e, w, r) for e in evals[:2]], 1: [po.intt(e, w, r) for e in evals[2:]]}
    points = {0: [[p0], [p0, p2]], 1: [[p0, p1], [p0]]}
    _, f, L = po.kzg_v2_proof_eval(r, polys, points, theta, theta2)
    g = lambda e: cp.batch_mul(curve, 1, fr_arr([e]))[0][0]
    assert (pi[0] == g(po.poly_eval(f, alpha, r))).all() and (pi[1] == g(po.poly_eval(L, alpha, r))).all()


@pytest.mark.parametrize("curve,log_n", [(0, 6), (1, 8), (0, 12), (1, 14), (0, 16)])
def test_gate_argument_fused_flat_program(shim, curve, log_n):
    """placeholder_quotient_hip::gate_argument as ONE launch over a flat program (zkhip_gate_eval_dev; gates_argument.hpp:93-121,
    203-216) against the C++ oracle's statement-by-statement restatement (cport.gate_argument_dfs, pinned to pyoracle at <= 2^8 rows) on
    a gate set with shared factors, rotations +-1 / +-2, a repeated column, a selector used with a rotation, a one-factor product, and a
    mask that is not all ones -- 2^6 ... 2^16 rows, extended domain 4 n, both curves.  Four evaluations must agree with the oracle bit
    for bit: fused; round 5's two launches per product; fused in several accumulating groups; fused from extension caches (twice)."""
    C = CURVES[curve]
    r, n, log_ext = C.r, 1 << log_n, log_n + 2
    ncols = 7
    cols = [cp.random_fr(curve, 1200 + c, n) for c in range(ncols)]
    cols[0][1::2] = 0                                           # selectors: zero on half / two thirds of the rows
    cols[1][::3] = 0
    mask = np.repeat(fr_arr([1]), n, axis=0)
    mask[-4:] = 0
    rng = po.SplitMix64(60 + log_n)
    # (coefficient, [(column, rotation), ...]); the first factor is the selector.  Degrees: every product has <= 4 factors of degree n - 1
    products = [(rng.next_mod(r), [(0, 0), (2, 0), (3, 1)]), (r - 1, [(0, 0), (4, -1)]), (rng.next_mod(r), [(0, 0), (2, 2), (2, 0)]),
                (rng.next_mod(r), [(1, 0), (3, -2), (5, 0)]), (5, [(1, 0)]), (rng.next_mod(r), [(1, 1), (6, 0), (6, 1)]),
                (rng.next_mod(r), [(4, 0), (5, 1)]), (rng.next_mod(r), [(0, 0), (6, -1), (3, 1)])]
    want = cp.gate_argument_dfs(curve, [(c, [(cols[k], rot) for k, rot in fs]) for c, fs in products], mask, 4 * n)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_ext + 1)])
    u64 = lambda v: np.array(v, dtype=np.uint64)
    fac = np.array([x for _, fs in products for k, rot in fs for x in (k, rot)], dtype=np.int64)
    for variant in (0, 1, 2, 3):
        out = np.zeros((4 * n, 4), dtype=np.uint64)
        deg = np.zeros(1, dtype=np.uint64)
        rc = shim.shim_gate_argument(curve, P(np.concatenate(cols)), ctypes.c_size_t(ncols), ctypes.c_size_t(log_n), P(u64([n - 1] * ncols)), P(mask),
                                     ctypes.c_size_t(n - 1), P(roots), ctypes.c_size_t(len(products)), P(fr_arr([c for c, _ in products])),
                                     P(u64([len(fs) for _, fs in products])), P(fac), ctypes.c_size_t(log_ext), variant, P(out), P(deg))
        assert rc == 0, variant
        assert (out == want).all(), variant
        assert int(deg[0]) == 3 * (n - 1) + (n - 1), variant   # the largest product (three factors) times the mask


@pytest.mark.parametrize("curve,log_n", [(0, 6), (1, 9), (0, 13), (1, 16)])
def test_prepare_lookup_input_flat(shim, curve, log_n):
    """prepare_lookup_input's numeric side (lookup_argument.hpp:435-496) over flattened expressions: l = selector * (table_id +
    sum_k theta^(k + 1) expression_k), every constraint ONE gate of the flat-program kernel, on the domain polynomial_dfs arithmetic ends
    up on (the smallest power of two that holds the degree).  Against the oracle's dense evaluation of the same sum (cport.gate_argument_dfs
    with an all-ones mask): a constraint of two linear expressions, one with a product of two columns and rotations (degree 3: the 4 n
    domain), one whose expression is a constant."""
    C = CURVES[curve]
    r, n = C.r, 1 << log_n
    ncols = 5
    cols = [cp.random_fr(curve, 3300 + c + log_n, n) for c in range(ncols)]
    cols[0][::2] = 0
    rng = po.SplitMix64(88 + log_n)
    theta = rng.next_mod(r)
    c1, c2, c3 = rng.next_mod(r), rng.next_mod(r), rng.next_mod(r)
    # constraints: (table_id, [expression = [(coefficient, [(column, rotation), ...]), ...], ...])
    constraints = [(1, [[(1, [(1, 0)])], [(c1, [(2, 1)]), (r - 1, [(3, 0)])]]),
                   (2, [[(c2, [(1, 0), (4, -1)]), (c3, [(2, 2)]), (5, [])]]),
                   (7, [[(9, [])]])]
    ones = np.repeat(fr_arr([1]), n, axis=0)
    want = []
    for tid, exprs in constraints:
        products, th, deg = [(tid, [(cols[0], 0)])], theta, 1
        for e in exprs:
            for coeff, fs in e:
                products.append((th * coeff % r, [(cols[0], 0)] + [(cols[k], rot) for k, rot in fs]))
                deg = max(deg, 1 + len(fs))
            th = th * theta % r
        size = n
        while size < deg * (n - 1) + 1:
            size <<= 1
        want.append(cp.gate_argument_dfs(curve, products, ones, size))
    u64 = lambda v: np.array(v, dtype=np.uint64)
    monos = [m for _, exprs in constraints for e in exprs for m in e]
    fac = np.array([x for _, fs in monos for k, rot in fs for x in (k, rot)] or [0, 0], dtype=np.int64)
    roots = np.stack([limbs(C.root_of_unity(l), 4) for l in range(log_n + 3)])
    out = np.zeros((sum(len(w) for w in want), 4), dtype=np.uint64)
    sizes = np.zeros(2 * len(constraints), dtype=np.uint64)
    rc = shim.shim_lookup_input_flat(curve, P(np.concatenate(cols)), ctypes.c_size_t(ncols), ctypes.c_size_t(log_n), P(u64([n - 1] * ncols)), P(roots),
                                     ctypes.c_size_t(len(constraints)), P(u64([x for tid, exprs in constraints for x in (tid, len(exprs))])),
                                     P(u64([len(e) for _, exprs in constraints for e in exprs])), P(fr_arr([c for c, _ in monos])), P(u64([len(fs) for _, fs in monos])),
                                     P(fac), P(limbs(theta, 4)), P(out), P(sizes))
    assert rc == 0
    assert [int(x) for x in sizes[0::2]] == [len(w) for w in want] == [2 * n, 4 * n, n]
    assert [int(x) for x in sizes[1::2]] == [2 * (n - 1), 3 * (n - 1), n - 1]
    assert (out == np.concatenate(want)).all()
