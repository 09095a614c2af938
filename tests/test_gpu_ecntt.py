"""DFT over group elements (zkhip_ec_ntt_dev): evaluate_all_lagrange_polynomials on the powers [tau^i] G as the
powers-of-tau result computes it (commitments/detail/polynomial/powers_of_tau/result.hpp:81-94).  The expected values
come from the scalar side: the inverse transform of tau^i G is L_j(tau) G (closed form, pyoracle.lagrange_at), the
forward transform is (sum_i omega^(ij) tau^i) G."""
import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, FQ_LIMBS, fr_arr, jac_to_affine_py, limbs, pt_from_limbs

pytestmark = pytest.mark.gpu


def _jac_of(curve, group, scalars):
    """canonical Jacobian (x, y, 1) of scalar * generator; the zero scalar gives Z = 0"""
    pts, inf = cp.batch_mul(curve, group, fr_arr(scalars))
    L = FQ_LIMBS[curve] * group
    out = np.zeros((len(scalars), 3 * L), dtype=np.uint64)
    out[:, : 2 * L] = pts
    out[:, 2 * L] = 1
    out[inf != 0] = 0
    return out


@pytest.mark.parametrize("curve,group,log_m", [(0, 1, 0), (0, 1, 1), (0, 1, 4), (1, 1, 6), (0, 2, 3), (1, 2, 2), (0, 1, 8), (0, 1, 11), (1, 2, 7), (1, 1, 9), (0, 1, 13), (1, 1, 14), (0, 2, 9)])
def test_ec_ntt_lagrange_basis(ctx, curve, group, log_m):
    C = CURVES[curve]
    r, m = C.r, 1 << log_m
    w = C.root_of_unity(log_m)
    tau = po.SplitMix64(100 + log_m).next_mod(r)
    powers = [pow(tau, i, r) for i in range(m)]
    if m >= 4:
        powers[2] = 0  # a point at infinity among the inputs
    jac = _jac_of(curve, group, powers)
    d = ctx.malloc(jac.nbytes)
    out = np.zeros_like(jac)

    def run(inverse):
        ctx.h2d(d, jac)
        ctx.ec_ntt_dev(curve, group, d, log_m, limbs(w, 4), inverse=inverse)
        ctx.d2h(out, d)
        return [jac_to_affine_py(curve, group, out[j]) for j in range(m)]

    def expect(scalars):
        pts, inf = cp.batch_mul(curve, group, fr_arr(scalars))
        return [pt_from_limbs(curve, group, pts[j], inf[j]) for j in range(m)]

    # the expected exponents: the O(m^2) definition at the small sizes, the oracle's transform (itself pinned to that definition) beyond
    fwd = [sum(pow(w, i * j, r) * powers[i] for i in range(m)) % r for j in range(m)] if m <= 64 else po.ntt(powers, w, r)
    assert run(False) == expect(fwd)
    minv, winv = pow(m, -1, r), pow(w, -1, r)
    inv = [minv * sum(pow(winv, i * j, r) * powers[i] for i in range(m)) % r for j in range(m)] if m <= 64 else po.intt(powers, w, r)
    assert run(True) == expect(inv)
    if m >= 4:
        powers[2] = pow(tau, 2, r)
        jac = _jac_of(curve, group, powers)
    # with the genuine powers the inverse transform is the Lagrange basis at tau
    assert run(True) == expect(po.lagrange_at(m, w, tau, r))
    # round trip on the device
    ctx.ec_ntt_dev(curve, group, d, log_m, limbs(w, 4), inverse=False)
    ctx.d2h(out, d)
    assert [jac_to_affine_py(curve, group, out[j]) for j in range(m)] == expect(powers)
    ctx.free(d)


def test_ec_ntt_in_several_launches(ctx):
    """The multiplication pass of a large transform runs in several launches over one table region (option
    "ec_ntt_table_lanes"): the same result as in one."""
    curve, group, log_m = 0, 1, 8
    C = CURVES[curve]
    r, m = C.r, 1 << log_m
    w = C.root_of_unity(log_m)
    tau = po.SplitMix64(7).next_mod(r)
    jac = _jac_of(curve, group, [pow(tau, i, r) for i in range(m)])
    d = ctx.malloc(jac.nbytes)
    outs = []
    for lanes in (0, 64):
        ctx.set_option("ec_ntt_table_lanes", lanes)
        ctx.h2d(d, jac)
        ctx.ec_ntt_dev(curve, group, d, log_m, limbs(w, 4), inverse=True)
        out = np.zeros_like(jac)
        ctx.d2h(out, d)
        outs.append([jac_to_affine_py(curve, group, out[j]) for j in range(m)])
    ctx.set_option("ec_ntt_table_lanes", 0)
    ctx.free(d)
    assert outs[0] == outs[1]
    pts, inf = cp.batch_mul(curve, group, fr_arr(po.lagrange_at(m, w, tau, r)))
    assert outs[0] == [pt_from_limbs(curve, group, pts[j], inf[j]) for j in range(m)]


@pytest.mark.parametrize("curve,group,log_m", [(0, 1, 17), (0, 2, 12), (1, 1, 16)])
def test_ec_ntt_large_against_the_scalar_transform(ctx, zk, curve, group, log_m):
    """Sizes beyond the closed-form test above: P_i = k_i G with random k_i generated on the device; the DFT over the
    group of the P_i is the scalar DFT of the k_i in the exponent, out[j] = (sum_i omega^(ij) k_i) G -- 512 evenly spaced output points
    against the oracle's scalar transform (pinned to the O(n^2) definition) and its fixed-base multiples; then the inverse transform
    brings the inputs back (same sample + the point at infinity).  2^17 G1 points: every twiddle of 17 stages through the GLV split and the signed-window ladder."""
    C = CURVES[curve]
    r, m = C.r, 1 << log_m
    w = C.root_of_unity(log_m)
    ks = cp.random_fr(curve, 6000 + log_m, m)
    ks[5] = 0  # a point at infinity among the inputs
    bases = ctx.bases_from_scalars(curve, group, ks)
    pts, inf = bases.download()
    bases.free()
    L = FQ_LIMBS[curve] * group
    jac = np.zeros((m, 3 * L), dtype=np.uint64)
    jac[:, : 2 * L] = pts
    jac[:, 2 * L] = 1
    jac[inf != 0] = 0
    d = ctx.malloc(jac.nbytes)
    ctx.h2d(d, jac)
    ctx.ec_ntt_dev(curve, group, d, log_m, limbs(w, 4), inverse=False)
    out = np.zeros_like(jac)
    ctx.d2h(out, d)
    exps = cp.ntt(curve, ks.reshape(1, m, 4), log_m, limbs(w, 4))[0]
    exp_pts, exp_inf = cp.batch_mul(curve, group, exps)
    # compare in affine (python normaliser: one inversion per point, hence a sample)
    idx = list(range(0, m, m // 512))
    for j in idx:
        assert jac_to_affine_py(curve, group, out[j]) == pt_from_limbs(curve, group, exp_pts[j], exp_inf[j]), j
    ctx.ec_ntt_dev(curve, group, d, log_m, limbs(w, 4), inverse=True)
    ctx.d2h(out, d)
    for j in idx + [5]:
        assert jac_to_affine_py(curve, group, out[j]) == pt_from_limbs(curve, group, pts[j], inf[j]), j
    ctx.free(d)
