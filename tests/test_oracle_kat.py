"""The oracle is pinned before it is trusted: pyoracle against the reference's own known-answer vectors
(tests/golden/ref_kat.json, literals extracted by tests/golden/make_ref_kat.py), then the C++ restatement
(oracle/zk_oracle.cpp) against pyoracle.  CPU only."""
import json
import os
import random

import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, fr_arr, fr_ints, limbs, pt_limbs, pts_arr

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "ref_kat.json")))
h = lambda s: int(s, 16)


def test_ref_polynomial_kat():
    """AGG:578-862 bls381_polynomial_test: 256 Fr coefficients + 1 evaluation."""
    p = KAT["polynomial_test"]
    r = po.BLS12_381.r
    tr = [h(x) for x in p["tr"]]
    assert po.ipp2_poly_coeffs(tr, h(p["r_shift"]), r) == [h(x) for x in p["coeffs"]]
    assert po.ipp2_poly_eval(tr, h(p["kzg_challenge"]), h(p["r_shift"]), r) == h(p["eval"])


def _kat_msm_inputs():
    q = KAT["prove_commitment_test"]
    C = po.BLS12_381
    tr = [h(x) for x in q["tr"]]
    qv, ha, hb = po.ipp2_prove_commitment_v(C, q["n"], h(q["alpha"]), h(q["beta"]), tr, h(q["kzg_challenge"]))
    qw, ga, gb = po.ipp2_prove_commitment_w(C, q["n"], h(q["alpha"]), h(q["beta"]), tr, h(q["r_shift"]), h(q["kzg_challenge"]))
    g2 = lambda e: ((h(e[0][0]), h(e[0][1])), (h(e[1][0]), h(e[1][1])))
    g1 = lambda e: (h(e[0]), h(e[1]))
    return [
        (2, ha, qv, g2(q["comm_v"][0])),
        (2, hb, qv, g2(q["comm_v"][1])),
        (1, ga, qw, g1(q["comm_w"][0])),
        (1, gb, qw, g1(q["comm_w"][1])),
    ]


def test_ref_msm_kat_python():
    """AGG:864-930 bls381_prove_commitment_test: 2 x G2 (n=8) and 2 x G1 (n=16) MSM through multiexp."""
    C = po.BLS12_381
    for group, bases, scalars, expected in _kat_msm_inputs():
        G = C.g1 if group == 1 else C.g2
        assert po.msm_naive(G, bases, scalars) == expected
        assert po.msm_pippenger(G, bases, scalars, c=3) == expected


def test_ref_msm_kat_cport():
    for group, bases, scalars, expected in _kat_msm_inputs():
        out, inf = cp.msm(0, group, pts_arr(0, group, bases), fr_arr(scalars), chunks=2)
        assert inf == 0 and (out == pt_limbs(0, group, expected)).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_ref_kzg_basic_identity(curve):
    """kzg.cpp:75-103: commit({-1,1,2,3}; alpha=10) = 3209 * G (curve independent)."""
    C = CURVES[curve]
    k = KAT["kzg_basic_test"]
    ck = po.structured_generators(C.g1, 4, k["alpha"])
    f = [x % C.r for x in k["f"]]
    exp = C.g1.mul(C.g1.gen, k["commit_scalar"])
    assert po.msm_naive(C.g1, ck, f) == exp
    out, inf = cp.msm(curve, 1, pts_arr(curve, 1, ck), fr_arr(f))
    assert (out == pt_limbs(curve, 1, exp)).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_cport_matches_python(curve):
    C = CURVES[curve]
    one = np.array([[1, 0, 0, 0]], dtype=np.uint64)
    assert (cp.batch_mul(curve, 1, one)[0][0] == pt_limbs(curve, 1, C.g1.gen)).all()
    assert (cp.batch_mul(curve, 2, one)[0][0] == pt_limbs(curve, 2, C.g2.gen)).all()
    sc = cp.random_fr(curve, 7, 24)
    rng = po.SplitMix64(7)
    ks = [rng.next_mod(C.r) for _ in range(24)]
    assert fr_ints(sc) == ks
    for group in (1, 2):
        G = C.g1 if group == 1 else C.g2
        pts, inf = cp.batch_mul(curve, group, sc[:12])
        exp = G.batch_mul_gen(ks[:12])
        assert (pts == pts_arr(curve, group, exp)).all()
        s2 = cp.random_fr(curve, 8, 12)
        s2[0] = 0
        s2[1] = [1, 0, 0, 0]
        s2[2] = limbs(C.r - 1, 4)
        e = po.msm_naive(G, exp, fr_ints(s2))
        for kw in (dict(chunks=1), dict(chunks=3), dict(naive=True)):
            out, oinf = cp.msm(curve, group, pts, s2, **kw)
            assert oinf == 0 and (out == pt_limbs(curve, group, e)).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_ntt_oracles(curve):
    C = CURVES[curve]
    random.seed(5)
    for lg, rad in ((3, [8]), (4, [4, 4]), (5, [2, 4, 4]), (6, [4, 2, 8]), (6, [8, 8])):
        m = 1 << lg
        w = C.root_of_unity(lg)
        a = [random.randrange(C.r) for _ in range(m)]
        d = po.dft_naive(a, w, C.r)
        assert po.ntt(a, w, C.r) == d
        assert po.stockham_model(a, w, C.r, rad) == d
        assert po.intt(d, w, C.r) == a
        arr = fr_arr(a).reshape(1, m, 4)
        assert fr_ints(cp.ntt(curve, arr, lg, limbs(w, 4))[0]) == d
        g = C.fr_generator
        e = po.ntt(po.multiply_by_coset(a, g, C.r), w, C.r)
        oc = cp.ntt(curve, arr, lg, limbs(w, 4), coset=limbs(g, 4))
        assert fr_ints(oc[0]) == e
        assert (cp.ntt(curve, oc, lg, limbs(w, 4), inverse=True, coset=limbs(g, 4)) == arr).all()


@pytest.mark.parametrize("curve,log_n", [(0, 4), (1, 5), (0, 7)])
def test_fast_gate_argument_oracle_equals_dense_oracle(curve, log_n):
    """cport.gate_argument_dfs (limb arrays, transform-based resize: the yard-stick of the fused gate kernel at 2^12 - 2^16 rows) against
    pyoracle.gate_argument_dfs (big integers; held to the dense definition in test_placeholder_quotient_chain_definitions): several gates
    sharing factors, rotations +-1 / +-2, a repeated column, a constant term, a mask that is not all ones."""
    C = CURVES[curve]
    r, n = C.r, 1 << log_n
    rng = random.Random(100 + log_n)
    cols = [[rng.randrange(r) for _ in range(n)] for _ in range(5)]
    sel = [[rng.randrange(r) if rng.random() < 0.5 else 0 for _ in range(n)] for _ in range(2)]
    mask = [1 if i < n - 3 else 0 for i in range(n)]
    products = [(rng.randrange(r), [(sel[0], 0), (cols[0], 0), (cols[1], 1)]), (r - 1, [(sel[0], 0), (cols[2], -1)]),
                (rng.randrange(r), [(sel[1], 0), (cols[0], 2), (cols[0], 0), (cols[3], -2)]), (7, [(sel[1], 0)]),
                (rng.randrange(r), [(cols[4], 1), (cols[4], 1)])]
    ext = 4 * n
    want = po.gate_argument_dfs(products, mask, ext, C.root_of_unity, r)
    got = cp.gate_argument_dfs(curve, [(c, [(fr_arr(e), rot) for e, rot in fs]) for c, fs in products], fr_arr(mask), ext)
    assert fr_ints(got) == want


def _toy_root_py(r):
    return lambda leaves, per_leaf: (per_leaf + sum((i + 1) * v for i, v in enumerate(leaves))) % r


@pytest.mark.parametrize("curve,log_domain,steps", [(0, 8, [1, 2]), (1, 7, [2, 1, 1]), (0, 7, [3])])
def test_fast_scheme_oracle_lpc_equals_dense_oracle(curve, log_domain, steps):
    """cport.lpc_proof_eval / fri_leaves / fold_polynomial_dfs / toy_root (limb arrays, transform-based: the yard-stick of the LPC scheme at
    2^12 - 2^21 points) against pyoracle.lpc_proof_eval (big integers, dense arithmetic; held to the definitions in
    test_lpc_oracle_definitions): batch roots over the leaf layout, evaluations, FRI round roots, final polynomial."""
    C = CURVES[curve]
    r = C.r
    logs = [log_domain - 3, log_domain - 3, log_domain - 2, log_domain - 3]
    evals = [cp.random_fr(curve, 1300 + i, 1 << l) for i, l in enumerate(logs)]
    rng = po.SplitMix64(55 + curve)
    p0, p1, p2 = (rng.next_mod(r) for _ in range(3))
    etha, theta = rng.next_mod(r), rng.next_mod(r)
    challenges = [etha, etha, theta] + [rng.next_mod(r) for _ in range(sum(steps))]
    points = {0: [[p0], [p0, p2]], 1: [[p0, p1], [p0]]}
    e_roots, e_z, e_fri, e_final = po.lpc_proof_eval(r, {0: [fr_ints(e) for e in evals[:2]], 1: [fr_ints(e) for e in evals[2:]]}, points, [0], log_domain, steps,
                                                     C.root_of_unity, challenges, _toy_root_py(r))
    g_roots, g_z, g_fri, g_final = cp.lpc_proof_eval(curve, {0: evals[:2], 1: evals[2:]}, points, [0], log_domain, steps, challenges, cp.toy_root(curve))
    assert g_roots == e_roots and g_z == e_z and g_fri == e_fri
    nfinal = 1 << (log_domain - sum(steps))
    assert fr_ints(g_final) == (e_final + [0] * nfinal)[:nfinal]
    # the leaf layout on its own, every fri_step the schemes use
    ext = [fr_ints(cp.dfs_resize(curve, e, 1 << log_domain)) for e in evals[:3]]
    for step in (1, 2, 3):
        assert fr_ints(cp.fri_leaves([fr_arr(e) for e in ext], step)) == po.fri_leaves(ext, step)


@pytest.mark.parametrize("curve", [0, 1])
def test_fast_scheme_oracle_kzg_equals_dense_oracle(curve):
    """cport.kzg_v2_proof_eval / kzg_v1_proof_eval against pyoracle's (kzg_v2.hpp:236-305, kzg.hpp:782-807): evaluations and the quotient
    polynomials committed as pi_1, pi_2 / kzg_proof, coefficient by coefficient; ragged sizes and point sets."""
    C = CURVES[curve]
    r = C.r
    rng = po.SplitMix64(77 + curve)
    x1, x2, x3 = (rng.next_mod(r) for _ in range(3))
    layout = [(0, 6, [x1, x2]), (0, 6, [x1, x2]), (0, 6, [x1, x2]), (2, 6, [x2]), (2, 7, [x1, x3])]
    polys_i, polys_a, points = {}, {}, {}
    for p, (k, log_n, pts) in enumerate(layout):
        c = cp.ntt(curve, cp.random_fr(curve, 500 + p, 1 << log_n).reshape(1, -1, 4), log_n, limbs(C.root_of_unity(log_n), 4), inverse=True)[0]
        polys_a.setdefault(k, []).append(c)
        polys_i.setdefault(k, []).append(fr_ints(c))
        points.setdefault(k, []).append(pts)
    theta, theta2 = rng.next_mod(r), rng.next_mod(r)
    z, f, L = po.kzg_v2_proof_eval(r, polys_i, points, theta, theta2)
    gz, gf, gL = cp.kzg_v2_proof_eval(curve, polys_a, points, theta, theta2)
    assert gz == z and fr_ints(gf) == po.poly_trim(f) and fr_ints(gL) == po.poly_trim(L)
    z1, acc = po.kzg_v1_proof_eval(r, polys_i, points, theta)
    gz1, gacc = cp.kzg_v1_proof_eval(curve, polys_a, points, theta)
    assert gz1 == z1 and fr_ints(gacc) == po.poly_trim(acc)


@pytest.mark.parametrize("curve", [0, 1])
def test_ntt_oracle_threads_inside_equals_one_thread_per_polynomial(curve):
    """cport.ntt transforms a batch smaller than half the thread count with the threads INSIDE each transform (bench.py's CPU NTT baseline
    on all cores); that path must be the serial one's output bit for bit (which test_ntt_oracles pins to the O(n^2) definition), and a
    random-point Horner check holds it to the definition directly at 2^14."""
    C = CURVES[curve]
    lg, batch = 14, 2
    w, g = limbs(C.root_of_unity(lg), 4), limbs(C.fr_generator, 4)
    a = cp.random_fr(curve, 31, batch << lg).reshape(batch, 1 << lg, 4)
    was = cp.num_threads()
    try:
        for inverse, coset in ((False, None), (True, None), (False, g), (True, g)):
            cp.set_threads(max(4, 2 * batch))
            inside = cp.ntt(curve, a, lg, w, inverse=inverse, coset=coset)
            cp.set_threads(1)
            serial = cp.ntt(curve, a, lg, w, inverse=inverse, coset=coset)
            assert (inside == serial).all(), (inverse, coset is not None)
        cp.set_threads(max(4, 2 * batch))
        ev = cp.ntt(curve, a, lg, w)
    finally:
        cp.set_threads(was)
    # out[i] = a(omega^i): three random rows by Horner over the coefficients
    coeffs = fr_ints(a[0])
    for i in (1, 777, (1 << lg) - 1):
        x = pow(C.root_of_unity(lg), i, C.r)
        assert po.from_limbs(ev[0][i]) == po.poly_eval(coeffs, x, C.r)


@pytest.mark.parametrize("curve", [0, 1])
def test_groth16_oracles_in_exponent(curve):
    """Whole proofs are unpinned in the reference (random r, s); pin both oracles to the trapdoor identity."""
    C = CURVES[curve]
    M, n = 16, 3
    cs, prim, aux = po.r1cs_example_field_input(C.r, M, n, seed=1)
    assert po.r1cs_is_satisfied(cs, prim, aux, C.r)
    g = cp.Groth16(curve, M, n, seed=1)
    assert g.is_satisfied() and fr_ints(g.assignment()) == prim + aux
    w = C.root_of_unity(g.log_m)
    rng = po.SplitMix64(99)
    trap = [rng.next_mod(C.r) for _ in range(5)]
    rr, ss = rng.next_mod(C.r), rng.next_mod(C.r)
    g.keygen(fr_arr(trap), limbs(w, 4))
    H = g.witness_map(limbs(w, 4), limbs(C.fr_generator, 4))
    assert fr_ints(H) == po.witness_map(po.swap_AB_if_beneficial(cs), prim, aux, w, C.fr_generator, C.r)
    proof = g.prove(limbs(rr, 4), limbs(ss, 4), limbs(w, 4), limbs(C.fr_generator, 4), chunks=2)
    eA, eB, eC = po.groth16_expected_in_exponent(C, cs, prim, aux, trap, rr, ss, w)
    assert (proof == np.concatenate([pt_limbs(curve, 1, eA), pt_limbs(curve, 2, eB), pt_limbs(curve, 1, eC)])).all()
    pk = po.groth16_keygen(C, cs, trap, w)
    assert po.groth16_prove(C, pk, prim, aux, rr, ss, w) == (eA, eB, eC)


def test_reference_serialisation_vectors():
    """AGG:932-1010: the byte encodings of an Fr element, a G1 and a G2 point (bincode::curve<bls12<381>>)"""
    k = KAT["serialisation_test"]
    assert list(int(k["fr"], 16).to_bytes(32, "little")) == k["fr_bytes"]
    g1 = (int(k["g1"][0], 16), int(k["g1"][1], 16))
    assert po.BLS12_381.g1.on_curve(g1)
    assert list(po.bls12_381_compress(1, g1)) == k["g1_bytes"]
    assert po.bls12_381_decompress(1, bytes(k["g1_bytes"])) == g1
    g2 = tuple((int(c[0], 16), int(c[1], 16)) for c in k["g2"])
    assert po.BLS12_381.g2.on_curve(g2)
    assert list(po.bls12_381_compress(2, g2)) == k["g2_bytes"]
    assert po.bls12_381_decompress(2, bytes(k["g2_bytes"])) == g2
    # round trips on other points, both signs, infinity
    for grp, G in ((1, po.BLS12_381.g1), (2, po.BLS12_381.g2)):
        for s in (1, 2, 3, 12345, po.BLS12_381.r - 1):
            P = G.mul(G.gen, s)
            assert po.bls12_381_decompress(grp, po.bls12_381_compress(grp, P)) == P
        assert po.bls12_381_decompress(grp, po.bls12_381_compress(grp, None)) is None


def test_lpc_oracle_definitions():
    """The oracle's LPC / FRI restatement is unpinned in the reference (its tests only check verify == true with hashes this
    tree does not hold): pin it to the definitions -- folding in coefficient form (f = f_e(X^2) + X f_o(X^2) -> f_e + alpha f_o),
    the shift as f(omega X), the leaf layout as a permutation into cosets, and the low degree of the final polynomial."""
    C = CURVES[0]
    r = C.r
    rng = po.SplitMix64(8)
    log_n = 6
    n = 1 << log_n
    w = C.root_of_unity(log_n)
    c = [rng.next_mod(r) for _ in range(n)]
    f = po.ntt(c, w, r)
    alpha = rng.next_mod(r)
    folded_c = [(c[2 * i] + alpha * c[2 * i + 1]) % r for i in range(n // 2)]
    assert po.fold_polynomial_dfs(f, alpha, w, r) == po.ntt(folded_c, w * w % r, r)
    assert po.polynomial_shift(f, 1) == po.ntt([x * pow(w, i, r) % r for i, x in enumerate(c)], w, r)
    assert po.polynomial_shift(f, -2, 16) == [f[(i - 2 * (n // 16)) % n] for i in range(n)]
    # leaves: every evaluation appears exactly once, and a leaf is closed under x -> -x (pairs s, s + D/2) and, for step 2, x -> i x
    ident = list(range(n))
    for step in (1, 2, 3):
        lv = po.fri_leaves([ident], step)
        assert sorted(lv) == ident
        per = 1 << step
        for x in range(n // per):
            leaf = set(lv[x * per:(x + 1) * per])
            assert all((s + n // 2) % n in leaf for s in leaf)
            if step >= 2:
                assert all((s + n // 4) % n in leaf for s in leaf)
    # whole proof_eval on a toy tree: final polynomial of degree < 2^(max log n) / 2^rounds
    evals = [[rng.next_mod(r) for _ in range(8)], [rng.next_mod(r) for _ in range(8)], [rng.next_mod(r) for _ in range(16)]]
    p0, p1, etha, theta, a0, a1 = (rng.next_mod(r) for _ in range(6))
    root = lambda leaves, per: (per + sum((i + 1) * v for i, v in enumerate(leaves))) % r
    roots, z, fri_roots, final = po.lpc_proof_eval(r, {0: evals[:2], 1: evals[2:]}, {0: [[p0], [p0, p1]], 1: [[p1]]}, [0], 6, [1, 1], C.root_of_unity,
                                                   [etha, etha, theta, a0, a1], root)
    assert len(fri_roots) == 2 and len(final) == 16 and not any(final[16 // 4:])
    assert z[1][0][0] == po.poly_eval(po.intt(evals[2], C.root_of_unity(4), r), p1, r)


# ---- evaluation domains: crypto3-math is absent from the reference tree, so the restatement of
# make_evaluation_domain's radix-2 family is pinned to the DEFINITIONS over each domain's point set --------------------
def _domain_definitions(C, dom, rng):
    r, m, xs = C.r, dom.m, dom.elements()
    assert len(set(xs)) == m
    a = [rng.next_mod(r) for _ in range(m)]
    ev = [po.poly_eval(a, x, r) for x in xs]
    assert dom.fft(a) == ev                       # fft = evaluation at get_domain_element(i)
    assert dom.inverse_fft(ev) == a               # inverse_fft = interpolation
    t = rng.next_mod(r)
    Z = 1
    for x in xs:
        Z = Z * (t - x) % r
    assert dom.compute_vanishing_polynomial(t) == Z
    lag = []
    for i, x in enumerate(xs):
        num = den = 1
        for j, y in enumerate(xs):
            if i != j:
                num, den = num * (t - y) % r, den * (x - y) % r
        lag.append(num * pow(den, -1, r) % r)
    assert dom.evaluate_all_lagrange_polynomials(t) == lag
    g = C.fr_generator
    P = [rng.next_mod(r) for _ in range(m)]
    assert dom.divide_by_z_on_coset(P, g) == [P[i] * pow(dom.compute_vanishing_polynomial(g * xs[i] % r), -1, r) % r for i in range(m)]
    # multiply_by_coset + fft = evaluation on g * domain (what divide_by_z_on_coset is applied to, r1cs_to_qap.hpp:266-308)
    assert dom.fft(po.multiply_by_coset(a, g, r)) == [po.poly_eval(a, g * x % r, r) for x in xs]
    H = [rng.next_mod(r) for _ in range(m + 1)]
    H2, c, zc = list(H), rng.next_mod(r), po.vanishing_poly(xs, r)
    dom.add_poly_z(c, H2)
    assert H2 == [(H[i] + c * zc[i]) % r for i in range(m + 1)]


@pytest.mark.parametrize("curve", [0, 1])
def test_evaluation_domain_definitions(curve):
    C = CURVES[curve]
    rng = po.SplitMix64(21 + curve)
    for n in (2, 3, 5, 6, 9, 10, 12, 17, 18, 19, 20, 24, 27, 33, 40, 48):
        _domain_definitions(C, po.make_evaluation_domain(C, n), rng)
    for s in (1, 2, 3, 4):    # the extended domain exists at 2^(s+1) points only: pretend a small two-adicity
        dom = po.make_evaluation_domain(C, 1 << (s + 1), two_adicity=s)
        assert dom.kind == po.EvaluationDomain.EXTENDED
        _domain_definitions(C, dom, rng)


def test_evaluation_domain_selection():
    """the order of get_evaluation_domain: basic, extended, step at min_size, then at big + rounded_small"""
    B, E, S = po.EvaluationDomain.BASIC, po.EvaluationDomain.EXTENDED, po.EvaluationDomain.STEP
    table = {2: (B, 2), 3: (S, 3), 7: (B, 8),          # 7 = 4 + 3: step(7) fails (3 is no power of two), 4 + 4 = 8 is basic
             11: (S, 12), 19: (S, 20), 27: (B, 32), 1 << 20: (B, 1 << 20),
             (1 << 20) + 11: (S, (1 << 20) + 16),      # BASELINE cfg 4: M = 2^20, n = 10
             (1 << 15) + 16: (S, (1 << 15) + 16), (1 << 10) + 11: (S, (1 << 10) + 16), (1 << 4) + 3: (S, 20),
             (1 << 20) + (1 << 19) + 1: (B, 1 << 21)}
    for n, exp in table.items():
        assert po.evaluation_domain_choice(n, 32) == exp, n
        assert cp.domain_choice(n, 32) == exp, n
    # beyond the two-adicity (BN254: s = 28): the extended domain at exactly 2^(s+1), also reached by rounding up
    for n in (1 << 29, (1 << 28) + (1 << 27) + 1):
        assert po.evaluation_domain_choice(n, 28) == (E, 1 << 29) and cp.domain_choice(n, 28) == (E, 1 << 29)
    with pytest.raises(ValueError):    # only a geometric / arithmetic sequence domain would do: out of scope
        po.evaluation_domain_choice((1 << 28) + 5, 28)


@pytest.mark.parametrize("curve", [0, 1])
def test_evaluation_domain_cport(curve):
    """the C++ restatement's domains against pyoracle's: transforms, and the whole Groth16 pipeline over the domain
    make_evaluation_domain(M + n + 1) picks (witness map, key, proof == the trapdoor identity)"""
    C = CURVES[curve]
    r = C.r
    rng = po.SplitMix64(5 + curve)
    for n, s in ((3, None), (5, None), (12, None), (20, None), (48, None), (1040, None), (32, None), (16, 3)):
        dom = po.make_evaluation_domain(C, n, two_adicity=s)
        a = [rng.next_mod(r) for _ in range(dom.m)]
        f = cp.domain_fft(curve, dom.kind, fr_arr(a), limbs(dom.omega, 4), limbs(dom.shift, 4))
        assert fr_ints(f) == dom.fft(a)
        assert fr_ints(cp.domain_fft(curve, dom.kind, f, limbs(dom.omega, 4), limbs(dom.shift, 4), inverse=True)) == a
    for M, n in ((8, 2), (13, 3), (16, 3), (1024, 11)):
        cs, prim, aux = po.r1cs_example_field_input(r, M, n, 7)
        dom = po.qap_domain(C, cs)
        assert dom.kind == po.EvaluationDomain.STEP
        g = cp.Groth16(curve, M, n, 7)
        g.set_domain(dom.kind, dom.m, limbs(dom.omega, 4), limbs(dom.shift, 4))
        assert g.m == dom.m
        trap = [rng.next_mod(r) for _ in range(5)]
        rr, ss = rng.next_mod(r), rng.next_mod(r)
        w, gen = limbs(dom.omega, 4), limbs(C.fr_generator, 4)
        small = M <= 16
        if small:
            assert fr_ints(g.witness_map(w, gen)) == po.witness_map(cs, prim, aux, dom, C.fr_generator, r)
        ex = [po.from_limbs(x) for x in g.expected_exponents(fr_arr(trap), w, limbs(rr, 4), limbs(ss, 4))]
        G1, G2 = C.g1, C.g2
        expect = (G1.mul(G1.gen, ex[0]), G2.mul(G2.gen, ex[1]), G1.mul(G1.gen, ex[2]))
        if small:
            assert po.groth16_expected_in_exponent(C, cs, prim, aux, trap, rr, ss, dom) == expect
            pk = po.groth16_keygen(C, cs, trap, dom)
            assert len(pk.H_query) == dom.m - 1
            assert po.groth16_prove(C, pk, prim, aux, rr, ss, dom) == expect
        g.keygen(fr_arr(trap), w)
        assert g.query(3)[0].shape[0] == dom.m - 1
        proof = g.prove(limbs(rr, 4), limbs(ss, 4), w, gen, chunks=2)
        assert (proof == np.concatenate([pt_limbs(curve, 1, expect[0]), pt_limbs(curve, 2, expect[1]), pt_limbs(curve, 1, expect[2])])).all()


def test_kzg_v1_oracle_identity():
    """kzg_commitment_scheme::proof_eval (kzg.hpp:782-807) is unpinned in the reference (its test only checks verify_eval, with
    pairings): pin the restatement to the verifier's equation (:826-866) evaluated at a clear alpha --
    sum_j gamma^j (f_j - U_j)(alpha) Z_{T \\ S_j}(alpha) == accum(alpha) V_T(alpha) -- and to exact divisibility."""
    C = CURVES[0]
    r = C.r
    rng = po.SplitMix64(31)
    x1, x2, x3 = (rng.next_mod(r) for _ in range(3))
    polys = {0: [[rng.next_mod(r) for _ in range(16)] for _ in range(3)], 3: [[rng.next_mod(r) for _ in range(32)], [rng.next_mod(r) for _ in range(8)]]}
    points = {0: [[x1, x2], [x1, x2], [x2]], 3: [[x1, x3, x2], []]}
    gamma, alpha = rng.next_mod(r), rng.next_mod(r)
    z, accum = po.kzg_v1_proof_eval(r, polys, points, gamma)
    merged = sorted({x for k in points for pl in points[k] for x in pl})
    lhs, fac = 0, 1
    for k in sorted(polys):
        for i, c in enumerate(polys[k]):
            assert z[k][i] == [po.poly_eval(c, x, r) for x in points[k][i]]
            U = po.lagrange_interpolation(list(zip(points[k][i], z[k][i])), r)
            diff = po.vanishing_poly([x for x in merged if x not in points[k][i]], r)
            lhs = (lhs + fac * (po.poly_eval(c, alpha, r) - po.poly_eval(U, alpha, r)) * po.poly_eval(diff, alpha, r)) % r
            fac = fac * gamma % r
    assert lhs == po.poly_eval(accum, alpha, r) * po.poly_eval(po.vanishing_poly(merged, r), alpha, r) % r
    assert len(accum) == 32 - 3


def test_placeholder_quotient_chain_definitions():
    """The oracle's restatement of placeholder's quotient chain (prover.hpp:55-69, 220-277; gates_argument.hpp:203-216).
    split_polynomial is PINNED to the reference's own test (test/systems/plonk/placeholder/placeholder.cpp:513-532: the literal
    polynomial, max_degree 3 -> 4 parts, f(y) = sum_i f_i(y) y^(4 i)); the DFS arithmetic to its definition in dense coefficient
    form: the product of the factors, the division by X^n - 1 and the split, all by big-integer polynomial arithmetic."""
    C = CURVES[0]
    r = C.r
    f = [1, 3, 4, 1, 5, 6, 7, 2, 8, 7, 5, 6, 1, 2, 1, 1]            # placeholder.cpp:514
    parts = po.split_polynomial(f, 3)
    assert len(parts) == 4                                          # :515, 521
    rng = random.Random(11)
    y = rng.randrange(r)
    assert po.poly_eval(f, y, r) == sum(po.poly_eval(p, y, r) * pow(y, 4 * i, r) for i, p in enumerate(parts)) % r   # :525-531
    # a satisfied toy circuit over n = 16 rows: gate q * (w0 * w1(shifted by one row) * w2) with q != 0 only where w0 == 0, and a
    # second part w1 * w2 - w3 with w3 := w1 o w2 on the domain; both vanish on the domain, so F / Z is exact
    n, log_n = 16, 4
    root = C.root_of_unity
    w = root(log_n)
    cols = [[rng.randrange(r) for _ in range(n)] for _ in range(3)]
    q = [rng.randrange(1, r) if i % 2 == 0 else 0 for i in range(n)]
    cols[0] = [0 if q[i] else cols[0][i] for i in range(n)]
    w3 = [cols[1][i] * cols[2][i] % r for i in range(n)]
    mask = [1] * n
    theta = rng.randrange(r)
    ext = 4 * n                                                    # degree of q w0 w1 w2 <= 4 (n - 1) < 4 n
    G = po.gate_argument_dfs([(theta, [(q, 0), (cols[0], 0), (cols[1], 1), (cols[2], 0)])], mask, ext, root, r)
    # definition: coefficients of every factor, dense product, evaluation on the extended domain
    def coeffs(e):
        return po.intt(list(e), root(len(e).bit_length() - 1), r)
    shifted = [cols[1][(i + 1) % n] for i in range(n)]
    dense = po.poly_scale(po.poly_mul(po.poly_mul(coeffs(q), coeffs(cols[0]), r), po.poly_mul(coeffs(shifted), coeffs(cols[2]), r), r), theta, r)
    dense = po.poly_trim(po.poly_mul(dense, coeffs(mask), r))
    assert G == po.ntt(dense + [0] * (ext - len(dense)), root(ext.bit_length() - 1), r)
    F1_dense = po.poly_trim(po.poly_sub(po.poly_mul(coeffs(cols[1]), coeffs(cols[2]), r), coeffs(w3), r))
    F1 = po.ntt(F1_dense + [0] * (2 * n - len(F1_dense)), root(5), r)
    alphas = [rng.randrange(r), rng.randrange(r)]
    T = po.quotient_polynomial([G, F1], alphas, n, root, r)
    assert len(T) == ext - n
    total = po.poly_add(po.poly_scale(dense, alphas[0], r), po.poly_scale(F1_dense, alphas[1], r), r)
    Z = [r - 1] + [0] * (n - 1) + [1]
    assert po.poly_trim(po.poly_mul(po.poly_trim(T), Z, r)) == po.poly_trim(total)
    parts = po.quotient_polynomial_split_dfs(T, n, 4, n, root, r)
    assert len(parts) == 4
    back = []
    for p in parts:
        back += po.intt(p, w, r)
    assert po.poly_trim(back) == po.poly_trim(T)
    # an unsatisfied row is caught: the division is no longer exact
    bad = list(G)
    bad[4] = (bad[4] + 1) % r    # index 4 of the 4x extended domain IS a row of the original domain
    with pytest.raises(AssertionError):
        po.quotient_polynomial([bad, F1], alphas, n, root, r)


@pytest.mark.parametrize("curve,log_n,k_in,k_val,big", [(0, 5, 1, 1, ()), (1, 5, 2, 1, (1,)), (0, 6, 3, 2, (0,))])
def test_placeholder_lookup_argument_definitions(curve, log_n, k_in, k_val, big):
    """The oracle's restatement of placeholder's lookup argument (lookup_argument.hpp:153-296, 375-409, 565-638) against what the
    argument is FOR: on a genuine instance (every input drawn from the tables) sort_polynomials' result makes the grand product close,
    V_L[usable_rows] = 1 -- the reference's own BOOST_CHECK (:217) --, and every constraint polynomial F_0 .. F_3 vanishes on the rows
    (divisible by X^n - 1), by big-integer polynomial arithmetic.  An input that is in no table breaks both."""
    from util import lookup_instance
    C = CURVES[curve]
    r = C.r
    n = 1 << log_n
    rng = po.SplitMix64(7700 + 10 * curve + log_n + k_in)
    inputs, values, usable = lookup_instance(C, rng, log_n, k_in, k_val, big)
    red_in = [po.reduce_dfs_polynomial_domain(f, n) for f in inputs]
    assert all(len(v) == n for v in red_in)
    sorted_ = po.lookup_sort_polynomials(red_in, values, n, usable)
    assert len(sorted_) == k_in + k_val and all(sorted_[i][usable] == sorted_[i + 1][0] for i in range(len(sorted_) - 1))
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    alphas = [rng.next_mod(r) for _ in range(k_in + k_val - 1)]
    V, F = po.lookup_argument(inputs, values, sorted_, q_last, q_blind, L0, beta, gamma, alphas, usable, C.root_of_unity, r)
    assert V[0] == 1 and V[usable] == 1 and all(v == 0 for v in V[usable + 1:])

    def on_rows(f):  # f mod (X^n - 1)
        out = [0] * n
        for i, c in enumerate(f):
            out[i % n] = (out[i % n] + c) % r
        return out
    for f in F:
        assert not any(on_rows(f))
    assert any(F[2])                                                # not the zero polynomial: the blinding rows carry junk
    # the multi-part form (lookup_parts(max_quotient_chunks != 0), :243-276): F_2 is another polynomial that vanishes on the rows as well,
    # the last intermediate polynomial times the last group's ratio is V_L one row on
    if k_in + k_val > 1:
        sizes = [1, k_in + k_val - 1] if k_in + k_val < 5 else [2, 1, 2]
        pa = [rng.next_mod(r) for _ in range(len(sizes) - 1)]
        V2, F2, cur = po.lookup_argument(inputs, values, sorted_, q_last, q_blind, L0, beta, gamma, alphas, usable, C.root_of_unity, r, part_sizes=sizes, part_alphas=pa)
        assert V2 == V and F2[0] == F[0] and F2[1] == F[1] and F2[3] == F[3] and F2[2] != F[2]
        assert not any(on_rows(F2[2])) and len(cur) == len(sizes) - 1 and all(c[usable:] == V[usable:] for c in cur)
    # a looked-up value that is in the table but one time too few in `sorted`: the product no longer closes
    bad = [list(v) for v in sorted_]
    j = next(j for j in range(1, usable - 1) if bad[0][j] != bad[0][j + 1] and bad[0][j] != 0)
    bad[0][j] = bad[0][j + 1]
    Vb, Fb = po.lookup_argument(inputs, values, bad, q_last, q_blind, L0, beta, gamma, alphas, usable, C.root_of_unity, r)
    assert Vb[usable] != 1 and any(on_rows(Fb[1]))


@pytest.mark.parametrize("curve,log_n,k,chunks", [(0, 5, 3, 0), (0, 5, 3, 2), (1, 5, 4, 3), (0, 6, 5, 3)])
def test_placeholder_permutation_argument_definitions(curve, log_n, k, chunks):
    """The oracle's restatement of placeholder's permutation argument (permutation_argument.hpp:70-224), one-part and multi-part
    (max_quotient_chunks != 0: groups of chunks - 1 factors, intermediate polynomials, :147-160, 188-207), against what the argument is FOR:
    on a genuine copy-constraint instance (columns constant along the cycles of a permutation of the usable rows' cells, blinding rows
    behind) the grand product closes at usable_rows and F_0, F_1, F_2 vanish on every row; a broken copy breaks F_2."""
    from util import permutation_instance
    C = CURVES[curve]
    r = C.r
    n = 1 << log_n
    usable = n - 3
    rng = po.SplitMix64(7900 + 10 * curve + log_n + k + chunks)
    cols, S_id, S_sigma = permutation_instance(C, rng, log_n, k, usable)
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    parts = 1 if chunks == 0 else -(-k // (chunks - 1))
    alphas = [rng.next_mod(r) for _ in range(parts - 1)]
    res = po.permutation_argument(cols, S_id, S_sigma, q_last, q_blind, L0, beta, gamma, C.root_of_unity, r, chunks, alphas, usable)
    V, F = res[0], res[1]
    assert V[0] == 1 and V[usable] == 1

    def on_rows(f):
        out = [0] * n
        for i, c in enumerate(f):
            out[i % n] = (out[i % n] + c) % r
        return out
    for f in F:
        assert not any(on_rows(f))
    assert any(F[1])
    if chunks:
        assert len(res[2]) == parts - 1 and all(c[usable:] == V[usable:] for c in res[2])
        V1, F1 = po.permutation_argument(cols, S_id, S_sigma, q_last, q_blind, L0, beta, gamma, C.root_of_unity, r)
        assert V1 == V and F1[0] == F[0] and F1[2] == F[2]
        # the multi-part F_1 has a lower degree than the one-part F_1: what the chunks are for
        assert parts == 1 or len(F[1]) < len(F1[1])
    bad = [list(c) for c in cols]
    bad[0][1] = (bad[0][1] + 1) % r
    Vb, Fb = po.permutation_argument(bad, S_id, S_sigma, q_last, q_blind, L0, beta, gamma, C.root_of_unity, r)[:2]
    assert Vb[usable] != 1 and any(on_rows(Fb[2]))


def test_lookup_prepare_value_and_argument_order():
    """pyoracle.lookup_prepare_value (lookup_argument.hpp:411-433) pinned to its definition: on the rows of the basic domain the polynomial
    takes mask tag ((t + 1) + sum_i theta^(i + 1) constant_i) -- a product of three degree-(n - 1) polynomials, so it lives on 4n points
    (2n without table columns) and its coefficients above 3 (n - 1) are zero.  The replayed call order of the two arguments is pinned to the
    statement counts of the reference's own code (one part: no part alphas, no intermediate appends)."""
    C = po.BLS12_381
    r, n = C.r, 16
    rng = po.SplitMix64(41)
    vec = lambda: [rng.next_mod(r) for _ in range(n)]
    tag, mask, consts, theta = [1 if 1 <= j <= 9 else 0 for j in range(n)], [1] * 13 + [0] * 3, [vec() for _ in range(3)], rng.next_mod(r)
    tables = [(0, 2, [[0, 1]]), (0, 1, [[2], [0]]), (0, 0, [[]])]
    vals = po.lookup_prepare_value(tables, [tag], consts, theta, mask, C.root_of_unity, r)
    assert [len(v) for v in vals] == [4 * n, 4 * n, 4 * n, 2 * n]
    want = [[(1 + theta * consts[0][j] + theta * theta * consts[1][j]) % r for j in range(n)], [(2 + theta * consts[2][j]) % r for j in range(n)],
            [(2 + theta * consts[0][j]) % r for j in range(n)], [3] * n]
    for v, w in zip(vals, want):
        assert po.reduce_dfs_polynomial_domain(v, n) == [mask[j] * tag[j] * w[j] % r for j in range(n)]
        c = po.intt(list(v), C.root_of_unity(len(v).bit_length() - 1), r)
        assert not any(c[3 * (n - 1) + 1:])
    E = po
    assert po.permutation_argument_events(1) == [E.EV_CHALLENGE, E.EV_CHALLENGE, E.EV_APPEND + 2]
    assert po.permutation_argument_events(3) == [1, 1, 102, 1, 1, 102, 102]
    assert po.lookup_argument_events(2, 1) == [1, 104, 104, 204, 3, 1, 1, 102, 1]
    assert po.lookup_argument_events(3, 2) == [1, 104, 104, 104, 204, 3, 1, 1, 1, 102, 102, 1, 1]


@pytest.mark.parametrize("curve,log_n,k,chunks", [(0, 5, 3, 0), (1, 6, 4, 3), (0, 8, 5, 3), (1, 7, 2, 8)])
def test_fast_argument_oracle_equals_dense_oracle_permutation(curve, log_n, k, chunks):
    """The C++-backed restatement of the permutation argument (oracle/cport.py permutation_argument: transform-based products, the row
    recurrence with one inversion per row in liboracle.so) -- what the -m gpu suite holds the device against at 2^12 and 2^16 rows -- PINNED to
    pyoracle.permutation_argument's dense big-integer arithmetic (itself pinned to what the argument is for): V_P, the intermediate
    polynomials of the multi-part form and F_0 .. F_2 coefficient by coefficient."""
    from util import CURVES, fr_arr, fr_ints, permutation_instance
    C = CURVES[curve]
    r, n = C.r, 1 << log_n
    usable = n - 3
    rng = po.SplitMix64(500 + log_n + k)
    cols, S_id, S_sigma = permutation_instance(C, rng, log_n, k, usable)
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    parts = 1 if chunks == 0 else -(-k // (chunks - 1))
    alphas = [rng.next_mod(r) for _ in range(parts - 1)]
    want = po.permutation_argument(cols, S_id, S_sigma, q_last, q_blind, L0, beta, gamma, C.root_of_unity, r, chunks, alphas, usable)
    A = fr_arr
    got = cp.permutation_argument(curve, [A(c) for c in cols], [A(c) for c in S_id], [A(c) for c in S_sigma], A(q_last), A(q_blind), A(L0), beta, gamma, chunks, alphas,
                                  usable)
    assert fr_ints(got[0]) == want[0]
    assert [fr_ints(f) for f in got[1]] == want[1]
    if chunks:
        assert [fr_ints(c) for c in got[2]] == want[2]


@pytest.mark.parametrize("curve,log_n,k_in,k_val,big,part_sizes", [(0, 6, 1, 1, (), None), (1, 5, 2, 1, (1,), [2, 1]), (0, 7, 3, 2, (0,), [2, 1, 2]), (1, 8, 2, 2, (), None)])
def test_fast_argument_oracle_equals_dense_oracle_lookup(curve, log_n, k_in, k_val, big, part_sizes):
    """likewise for the lookup argument (cport.lookup_sort_polynomials, lookup_argument; a `big` input lives on the 2n-point domain)"""
    from util import CURVES, fr_arr, fr_ints, lookup_instance
    C = CURVES[curve]
    r, n = C.r, 1 << log_n
    rng = po.SplitMix64(600 + log_n + k_in)
    inputs, values, usable = lookup_instance(C, rng, log_n, k_in, k_val, big)
    red_in = [po.reduce_dfs_polynomial_domain(f, n) for f in inputs]
    sorted_ = po.lookup_sort_polynomials(red_in, values, n, usable)
    q_last = [1 if j == usable else 0 for j in range(n)]
    q_blind = [1 if j > usable else 0 for j in range(n)]
    L0 = [1] + [0] * (n - 1)
    beta, gamma = rng.next_mod(r), rng.next_mod(r)
    alphas = [rng.next_mod(r) for _ in range(k_in + k_val - 1)]
    pa = [rng.next_mod(r) for _ in range(len(part_sizes) - 1)] if part_sizes else []
    want = po.lookup_argument(inputs, values, sorted_, q_last, q_blind, L0, beta, gamma, alphas, usable, C.root_of_unity, r, part_sizes, pa)
    A = fr_arr
    s2 = cp.lookup_sort_polynomials([A(x) for x in red_in], [A(x) for x in values], n, usable)
    assert [fr_ints(x) for x in s2] == sorted_
    got = cp.lookup_argument(curve, [A(x) for x in inputs], [A(x) for x in values], s2, A(q_last), A(q_blind), A(L0), beta, gamma, alphas, usable, part_sizes, pa)
    assert fr_ints(got[0]) == want[0]
    assert [fr_ints(f) for f in got[1]] == want[1]
    if part_sizes:
        assert [fr_ints(c) for c in got[2]] == want[2]
