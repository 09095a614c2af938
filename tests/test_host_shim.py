"""Host-side logic of the header-only C++ shim that needs no GPU: the small-polynomial helpers of the KZG opening
proof (get_U / get_V: Lagrange interpolation, vanishing polynomial), the group operators the Groth16 prover's last
lines use on the host (prover.hpp:142-155), and the query slices of the sharded prover -- against the big-integer
oracle.  (tests/cpp/libshimtest.so links libzkhip.so but these entry points never touch the device.)"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, fr_arr, fr_ints, limbs, pt_from_limbs, pt_limbs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(ROOT, "tests", "cpp", "libshimtest.so")
    src = os.path.join(ROOT, "tests", "cpp", "shim_test.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    return ctypes.CDLL(so)


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("k", [1, 2, 5])
def test_lagrange_and_vanishing(shim, curve, k):
    r = CURVES[curve].r
    rng = po.SplitMix64(31 * k + curve)
    xs = [rng.next_mod(r) for _ in range(k)]
    ys = [rng.next_mod(r) for _ in range(k)]
    at = rng.next_mod(r)
    U = po.lagrange_interpolation(list(zip(xs, ys)), r)
    V = po.vanishing_poly(xs, r)
    u_at, u_c, v_c = np.zeros(4, dtype=np.uint64), np.zeros((k, 4), dtype=np.uint64), np.zeros((k + 1, 4), dtype=np.uint64)
    assert shim.shim_host_small_poly(curve, P(fr_arr(xs)), P(fr_arr(ys)), ctypes.c_size_t(k), P(limbs(at, 4)), P(u_at), P(u_c), P(v_c)) == 0
    assert po.from_limbs(u_at) == po.poly_eval(U, at, r)
    assert [po.from_limbs(c) for c in u_c] == (U + [0] * k)[:k]
    assert [po.from_limbs(c) for c in v_c] == V
    for x, y in zip(xs, ys):  # the defining property, independent of the oracle's construction
        assert po.poly_eval([po.from_limbs(c) for c in u_c], x, r) == y


@pytest.mark.parametrize("curve", [0, 1])
def test_host_group_operators(shim, curve):
    C = CURVES[curve]
    G = C.g1
    Pt, Qt = G.mul(G.gen, 123456789), G.mul(G.gen, 987654321)
    k = po.SplitMix64(5).next_mod(C.r)
    L = len(pt_limbs(curve, 1, Pt))
    o_sum, o_mul = np.zeros(L, dtype=np.uint64), np.zeros(L, dtype=np.uint64)
    assert shim.shim_host_group(curve, P(pt_limbs(curve, 1, Pt)), P(pt_limbs(curve, 1, Qt)), P(limbs(k, 4)), P(o_sum), P(o_mul)) == 0
    assert pt_from_limbs(curve, 1, o_sum) == G.add(Pt, Qt)
    assert pt_from_limbs(curve, 1, o_mul) == G.mul(Pt, k)
    # the width-5 signed-window multiplication at its edges: 0, 1, digit boundaries (15, 16, 17, 31, 32), runs of ones that carry
    # through every limb, the largest canonical scalar and 2^255 - 1 (not canonical, still a 256-bit integer the operator accepts)
    edge = [0, 1, 15, 16, 17, 31, 32, 33, (1 << 64) - 1, (1 << 64), (1 << 128) - 1, (1 << 192) + 16, C.r - 1, C.r - 16, (1 << 254) + (1 << 253) - 1,
            (1 << 255) - 1, 0x5555555555555555555555555555555555555555555555555555555555555555 % C.r, 0x0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F]
    for e in edge:
        assert shim.shim_host_group(curve, P(pt_limbs(curve, 1, Pt)), P(pt_limbs(curve, 1, Qt)), P(limbs(e, 4)), P(o_sum), P(o_mul)) == 0
        want = G.mul(Pt, e % G.order) if hasattr(G, "order") else G.mul(Pt, e)
        if want is None:  # the point at infinity leaves the shim as zeroed limbs
            assert not o_mul.any(), hex(e)
        else:
            assert pt_from_limbs(curve, 1, o_mul) == want, hex(e)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_query_shards_partition(shim, world):
    a, b, h, l = 1048579, 1000003, 2097151, 1048568  # the query sizes of a 2^20-constraint proof (B sparse)
    out = np.zeros((world, 8), dtype=np.uint64)
    shim.shim_host_query_shards(ctypes.c_size_t(world), ctypes.c_size_t(a), ctypes.c_size_t(b), ctypes.c_size_t(h), ctypes.c_size_t(l), P(out))
    for col, total in ((0, a), (2, b), (4, h), (6, l)):
        lo, n = out[:, col].astype(int), out[:, col + 1].astype(int)
        assert lo[0] == 0 and (lo[1:] == lo[:-1] + n[:-1]).all() and lo[-1] + n[-1] == total  # contiguous, complete
        assert n.max() - n.min() <= 1                                                          # balanced


# domain: "ref" = what make_evaluation_domain(M + n + 1) picks (step radix-2 for 20, 5011; basic for 111), "basic" = the basic
# domain of the next power of two named explicitly, "extended" = an extended radix-2 domain named explicitly
@pytest.mark.parametrize("curve,M,n,domain", [(0, 16, 3, "ref"), (0, 16, 3, "basic"), (0, 16, 3, "extended"), (1, 100, 10, "ref"), (1, 1024, 10, "ref"),
                                              (0, 5000, 10, "ref"), (0, 5000, 10, "basic"), (1, 5000, 10, "extended")])
def test_generator_host_side_matches_oracles(shim, curve, M, n, domain):
    """The host half of the device key generator (hip/r1cs_gg_ppzksnark_generator.hpp: swap_AB_if_beneficial,
    instance_map_with_evaluation = r1cs_to_qap.hpp:138-187 over hip/evaluation_domain.hpp, and the trapdoor exponents of a
    proof) against the C++ oracle, and at the small sizes against the big-integer oracle (po.groth16_expected_in_exponent)."""
    C = CURVES[curve]
    g = cp.Groth16(curve, M, n, seed=3)
    pow2 = 1 << (M + n).bit_length()
    shift = None
    if domain == "ref":
        dom = po.make_evaluation_domain(C, M + n + 1)
        assert (dom.kind, dom.m) == cp.domain_choice(M + n + 1, C.two_adicity)
        shim.shim_set_domain(-1, ctypes.c_size_t(0), None)
    elif domain == "basic":
        dom = po.EvaluationDomain(po.EvaluationDomain.BASIC, pow2, C.root_of_unity(pow2.bit_length() - 1), C.r)
        shim.shim_set_domain(0, ctypes.c_size_t(pow2), None)
    else:
        dom = po.EvaluationDomain(po.EvaluationDomain.EXTENDED, pow2, C.root_of_unity(pow2.bit_length() - 2), C.r, pow(C.fr_generator, 2, C.r))
        shift = limbs(dom.shift, 4)
        shim.shim_set_domain(1, ctypes.c_size_t(pow2), P(shift))
    g.set_domain(dom.kind, dom.m, limbs(dom.omega, 4), shift)
    w = dom.omega
    rng = po.SplitMix64(77 + M)
    trap = [rng.next_mod(C.r) for _ in range(5)]
    rr, ss = rng.next_mod(C.r), rng.next_mod(C.r)
    args = []
    keep = []
    for k in range(3):  # the UNswapped system: the shim swaps itself
        rp, cl, cf = g.csr(k)
        keep += [rp, cl, cf]
        args += [P(rp), P(cl), P(cf)]
    assignment = g.assignment()
    out = np.zeros((3, 4), dtype=np.uint64)
    T, W, R, S = fr_arr(trap), limbs(w, 4), limbs(rr, 4), limbs(ss, 4)
    try:
        assert shim.shim_host_qap_exponents(curve, ctypes.c_size_t(g.M), ctypes.c_size_t(g.n), ctypes.c_size_t(g.N), *args, P(assignment), P(T), P(W), P(R),
                                            P(S), P(out)) == 0
    finally:
        shim.shim_set_domain(-1, ctypes.c_size_t(0), None)
    exp = g.expected_exponents(T, W, R, S)
    assert (out == exp).all()
    if M <= 100:
        cs, prim, aux = po.r1cs_example_field_input(C.r, M, n, seed=3)
        eA, eB, eC = po.groth16_expected_in_exponent(C, cs, prim, aux, trap, rr, ss, dom)
        a, b, c = fr_ints(out)
        assert (C.g1.mul(C.g1.gen, a), C.g2.mul(C.g2.gen, b), C.g1.mul(C.g1.gen, c)) == (eA, eB, eC)


@pytest.mark.parametrize("value,expect", [(None, []), ("0,1,2,3", [0, 1, 2, 3]), ("5", [5]), ("0,0", [0, 0]), ("", []), ("0,,1", []), ("0,1,", []), ("a,b", []), ("0, 1", []),
                                          ("7,6,5,4,3,2,1,0", [7, 6, 5, 4, 3, 2, 1, 0])])
def test_zkhip_devices_environment_variable(shim, value, expect):
    """ZKHIP_DEVICES names the GPUs of the default device group (hip/backend.hpp: device_group::devices_from_env): a comma-separated list of
    device ids; anything malformed reads as unset (one GPU) rather than as a partial list"""
    old = os.environ.pop("ZKHIP_DEVICES", None)
    try:
        if value is not None:
            os.environ["ZKHIP_DEVICES"] = value
        out = (ctypes.c_int * 16)()
        n = shim.shim_host_devices_from_env(out, 16)
        assert [out[i] for i in range(n)] == expect
    finally:
        os.environ.pop("ZKHIP_DEVICES", None)
        if old is not None:
            os.environ["ZKHIP_DEVICES"] = old
