#!/usr/bin/env python3
"""Randomised differential run of the device path against the CPU oracle (checker only), for a time budget:
    python tests/fuzz_gpu.py [--seconds 300] [--seed 1]        (test infrastructure: it calls the oracle, so it lives under tests/)
Every iteration draws a shape the fixed tests do not enumerate -- curve, group, size (1 .. 6000, not only powers of two), offset,
scalar pattern (uniform, zeros / ones heavy, r - 1, few distinct values = large buckets, tiny values), MSM one by one / as a batch /
as a batch whose members share a sort, NTT size / batch / direction / coset, evaluation domain of any kind, Groth16 witness map over the
domain make_evaluation_domain picks, whole Groth16 proofs through the C++ shim, the grand products / pointwise kernels behind placeholder's
permutation and lookup arguments on vectors of any length, the gate argument's flat-program kernel on random programs, the device group's MSM / NTT over
1 .. 5 members and every transport -- and compares bit for bit.  Exit code 0 and a JSON line with the counts = no difference."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import cport as cp  # noqa: E402
import pyoracle as po  # noqa: E402
from util import CURVES, limbs, qap_domains  # noqa: E402


def scalars(rng, curve, n, pattern):
    r = CURVES[curve].r
    s = cp.random_fr(curve, int(rng.integers(1, 1 << 30)), n)
    if pattern == "zeros_ones":
        k = rng.integers(0, 3, size=n)
        s[k == 0] = 0
        s[k == 1] = np.array([1, 0, 0, 0], dtype=np.uint64)
    elif pattern == "minus_one":
        idx = rng.random(n) < 0.3
        s[idx] = np.array(po.to_limbs(r - 1, 4), dtype=np.uint64)
    elif pattern == "few_values":
        vals = cp.random_fr(curve, int(rng.integers(1, 1 << 30)), 3)
        s = vals[rng.integers(0, 3, size=n)]
    elif pattern == "tiny":
        s[:, 1:] = 0
        s[:, 0] &= np.uint64(0xFFFF)
    return np.ascontiguousarray(s)


def fuzz_msm(zk, ctx, rng, stats):
    curve, group = int(rng.integers(0, 2)), int(rng.integers(1, 3))
    n = int(rng.integers(1, (6000 if group == 1 else 1500) * SCALE))
    # the two-level bucket reduction (G1, window tables) starts at 2^16 buckets by default: draw its threshold, its run length and -- one call in
    # three -- a window size that gives it 2^9 ... 2^17 buckets to fold
    ctx.set_option("msm_window_bits", int(rng.integers(10, 19)) if rng.random() < 0.33 else 0)
    ctx.set_option("msm_tail_fold", int(rng.choice([0, 8, 8, 16])))
    ctx.set_option("msm_fold_run", int(rng.choice([0, 0, 1, 2, 4, 8])))
    b = ctx.bases_from_scalars(curve, group, cp.random_fr(curve, int(rng.integers(1, 1 << 30)), n))
    ctx.set_option("msm_window_bits", 0)
    pts, inf = b.download()
    off = int(rng.integers(0, n)) if rng.random() < 0.3 else 0
    cnt = n - off
    pattern = ["uniform", "zeros_ones", "minus_one", "few_values", "tiny"][int(rng.integers(0, 5))]
    sc = scalars(rng, curve, cnt, pattern)
    jac = 3 * zk.coord_limbs(curve, group) * 8
    d_s, d_o = ctx.malloc(max(1, cnt) * 32), ctx.malloc(3 * jac)
    ctx.h2d(d_s, sc)
    mode = int(rng.integers(0, 3))
    res = np.zeros((3, jac // 8), dtype=np.uint64)
    if mode == 0:
        ctx.msm_dev(b, d_s, d_o, offset=off, n=cnt)
        ctx.d2h(res, d_o)
        got = [res[0]]
        want = [(pts[off:], inf[off:], sc)]
    else:
        # a batch of three members over the same bases: mode 1 different scalars (rotations), mode 2 the SAME scalars (shared sort)
        d_s2, d_s3 = ctx.malloc(max(1, cnt) * 32), ctx.malloc(max(1, cnt) * 32)
        sc2, sc3 = (np.roll(sc, 1, axis=0), np.roll(sc, 2, axis=0)) if mode == 1 else (sc, sc)
        ctx.h2d(d_s2, sc2)
        ctx.h2d(d_s3, sc3)
        ptrs = [d_s, d_s2, d_s3] if mode == 1 else [d_s, d_s, d_s]
        ctx.set_option("msm_share_sort", int(rng.integers(0, 2)))
        ctx.msm_batch_dev([b, b, b], ptrs, [d_o, d_o + jac, d_o + 2 * jac], offsets=[off] * 3, ns=[cnt] * 3)
        ctx.set_option("msm_share_sort", 1)
        ctx.d2h(res, d_o)
        got = [res[0], res[1], res[2]]
        want = [(pts[off:], inf[off:], x) for x in (sc, sc2, sc3)]
        ctx.free(d_s2)
        ctx.free(d_s3)
    for g, (p, i, s) in zip(got, want):
        aff, ginf = ctx.jacobian_to_affine(curve, group, g)
        exp, einf = cp.msm(curve, group, p, s, inf=i, chunks=1 + int(rng.integers(0, 3)))
        if bool(ginf) != bool(einf) or (not einf and not (aff == exp).all()):
            raise SystemExit("MSM differs: curve %d group %d n %d off %d pattern %s mode %d" % (curve, group, n, off, pattern, mode))
    ctx.free(d_s)
    ctx.free(d_o)
    b.free()
    ctx.set_option("msm_tail_fold", 16)
    ctx.set_option("msm_fold_run", 0)
    stats["msm"] += len(got)


def fuzz_ntt(zk, ctx, rng, stats):
    curve = int(rng.integers(0, 2))
    C = CURVES[curve]
    log_m, batch = int(rng.integers(0, 15 + (3 if SCALE > 1 else 0))), int(rng.integers(1, 5))
    w = limbs(C.root_of_unity(log_m), 4)
    a = cp.random_fr(curve, int(rng.integers(1, 1 << 30)), batch << log_m).reshape(batch, 1 << log_m, 4)
    inverse, coset = bool(rng.integers(0, 2)), (limbs(C.fr_generator, 4) if rng.random() < 0.5 else None)
    got = ctx.ntt(curve, a, log_m, w, inverse=inverse, coset=coset)
    exp = cp.ntt(curve, a, log_m, w, inverse=inverse, coset=coset)
    if not (got == exp).all():
        raise SystemExit("NTT differs: curve %d log_m %d batch %d inverse %d coset %d" % (curve, log_m, batch, inverse, coset is not None))
    stats["ntt"] += 1


def fuzz_domain(zk, ctx, rng, stats):
    curve = int(rng.integers(0, 2))
    n = int(rng.integers(2, 20000 * SCALE))
    two_adicity = int(rng.integers(2, 12)) if rng.random() < 0.25 else None
    try:
        dom, zd = qap_domains(zk, curve, n, two_adicity=two_adicity)
    except Exception:
        stats["domain_skipped"] += 1  # no radix-2 domain of that size under the pretended two-adicity
        return
    m = dom.m
    batch = int(rng.integers(1, 4))
    a = cp.random_fr(curve, int(rng.integers(1, 1 << 30)), batch * m).reshape(batch, m, 4)
    sh = limbs(dom.shift, 4)
    exp = np.stack([cp.domain_fft(curve, dom.kind, a[b], limbs(dom.omega, 4), sh) for b in range(batch)])
    got = ctx.domain_fft(curve, zd, a)
    if not (got == exp).all() or not (ctx.domain_fft(curve, zd, got, inverse=True) == a).all():
        raise SystemExit("domain transform differs: curve %d %s" % (curve, dom.describe()))
    stats["domain"] += 1


def fuzz_witness(zk, ctx, rng, stats):
    curve = int(rng.integers(0, 2))
    C = CURVES[curve]
    M, nin = int(rng.integers(3, 5000 * SCALE)), int(rng.integers(1, 12))
    nin = min(nin, M + 2)  # the oracle's instances have M + 2 variables; more inputs than variables is no instance (the library says RANGE)
    g16 = cp.Groth16(curve, M, nin, seed=int(rng.integers(1, 1000)))
    kind, m = cp.domain_choice(M + nin + 1, C.two_adicity)
    wd = limbs(C.root_of_unity((M + nin).bit_length()), 4)
    gen = limbs(C.fr_generator, 4)
    g16.set_domain(kind, m, wd)
    r1cs = ctx.upload_r1cs(curve, g16.M, g16.n, g16.N, g16.csr(0), g16.csr(1), g16.csr(2))
    z = np.concatenate([np.array([[1, 0, 0, 0]], dtype=np.uint64), g16.assignment()])
    if (r1cs.kind, r1cs.m) != (kind, m) or not (ctx.groth16_witness_h(r1cs, z, wd, gen) == g16.witness_map(wd, gen)).all():
        raise SystemExit("witness map differs: curve %d M %d n %d" % (curve, M, nin))
    r1cs.free()
    stats["witness_map"] += 1


_SHIM = {}
SCALE = 1    # --scale: multiplies the size ranges (16: MSMs up to 96 000 points, NTTs up to 2^17, domains up to 320 000, witness maps up to 80 000 constraints)


def fuzz_proof(zk, ctx, rng, stats):
    """one whole Groth16 proof through the header-only shim (key from the oracle's generator, uploaded; prover on the device) against the
    oracle's prover -- random constraint count, input count, curve, evaluation domain (the reference's choice or the basic one)"""
    import ctypes
    import test_gpu_shim as tgs

    if "lib" not in _SHIM:
        _SHIM["lib"] = ctypes.CDLL(os.path.join(ROOT, "tests", "cpp", "libshimtest.so"))
    curve, M, nin = int(rng.integers(0, 2)), int(rng.integers(13, 3000)), int(rng.integers(1, 12))
    domain = "ref" if rng.random() < 0.7 else "basic"
    try:
        tgs._groth16_prover_shim(_SHIM["lib"], curve, M, nin, domain)
    except AssertionError:
        raise SystemExit("Groth16 proof differs: curve %d M %d n %d domain %s" % (curve, M, nin, domain))
    del tgs._KEEP[:]
    stats["groth16_proof"] += 1


def _up(ctx, vals):
    from util import fr_arr
    d = ctx.malloc(max(1, len(vals)) * 32)
    if len(vals):
        ctx.h2d(d, fr_arr(vals))
    return d


def _down(ctx, d, n):
    from util import fr_ints
    out = np.zeros((n, 4), dtype=np.uint64)
    if n:
        ctx.d2h(out, d)
    return fr_ints(out)


def fuzz_arguments(zk, ctx, rng, stats):
    """the kernels behind placeholder's permutation / lookup arguments through the C ABI on random vectors of any length (the recurrences are what
    is compared, no closing product needed): the two grand products against the oracle's row-by-row loops, a x + b y + c and a b / c against
    big integers"""
    curve = int(rng.integers(0, 2))
    r = CURVES[curve].r
    n = int(rng.integers(1, 3000 * SCALE))
    srng = po.SplitMix64(int(rng.integers(1, 1 << 30)))
    vec = lambda: [srng.next_mod(r) if rng.random() < 0.97 else int(rng.integers(0, 2)) * (r - 1) for _ in range(n)]
    beta, gamma = srng.next_mod(r), srng.next_mod(r)
    which = int(rng.integers(0, 4))
    if which == 3:
        # sort_polynomials (zkhip_lookup_sort_dev) over a SMALL alphabet, so that every branch of the reference's walk is hit: zero runs at
        # the head / in the middle / at the end, equal values in separate runs, looked-up values that are in no table; compared with the
        # reference as it runs without assertions (pyoracle strict = False); an emitted sequence that does not fit must raise bit 3 instead
        if n < 2:
            return
        import ctypes
        k_in, k_val = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        usable = int(rng.integers(0, n))
        sym = [0] + [srng.next_mod(r) for _ in range(int(rng.integers(1, 40)))]
        runs = float(rng.random())
        def table():
            out, cur = [], sym[int(rng.integers(0, len(sym)))]
            for _ in range(n):
                if rng.random() > runs:
                    cur = sym[int(rng.integers(0, len(sym)))]
                out.append(cur)
            return out
        foreign = rng.random() < 0.2
        values = [table() for _ in range(k_val)]
        inputs = [[sym[int(rng.integers(0, len(sym)))] if not (foreign and rng.random() < 0.01) else srng.next_mod(r) for _ in range(n)] for _ in range(k_in)]
        try:
            exp = po.lookup_sort_polynomials(inputs, values, n, usable, strict=False)
        except AssertionError:
            exp = None    # does not fit the vectors
        present = {v for col in values for v in col[:usable]}
        want_flags = (4 if any(v not in present for col in inputs for v in col[:usable]) else 0) | (8 if exp is None else 0)
        ptrs = [_up(ctx, v) for v in inputs + values]
        outs = [_up(ctx, [5] * n) for _ in range(k_in + k_val)]
        ctx.lookup_sort_dev(ptrs[:k_in], ptrs[k_in:], n, usable, outs)
        flags = ctypes.c_uint32()
        ctx.lib.zkhip_device_status(ctx.h, ctypes.byref(flags))
        ok = flags.value == want_flags and (exp is None or [_down(ctx, d, n) for d in outs] == exp)
        for p_ in ptrs + outs:
            ctx.free(p_)
        if not ok:
            raise SystemExit("sort_polynomials differs: n %d usable %d k_in %d k_val %d flags %d (expected %d)" % (n, usable, k_in, k_val, flags.value, want_flags))
        stats["lookup_sort"] = stats.get("lookup_sort", 0) + 1
        return
    if which == 0:
        k = int(rng.integers(1, 5))
        cols, sid, ssig = [vec() for _ in range(k)], [vec() for _ in range(k)], [vec() for _ in range(k)]
        g, h, V = po.permutation_grand_product(cols, sid, ssig, beta, gamma, r)
        ptrs = [_up(ctx, v) for v in cols + sid + ssig]
        d_g, d_h, d_v = ctx.malloc(k * n * 32), ctx.malloc(k * n * 32), ctx.malloc(n * 32)
        ctx.perm_grand_product_dev(curve, ptrs[:k], ptrs[k:2 * k], ptrs[2 * k:], n, limbs(beta, 4), limbs(gamma, 4), d_g, d_h, d_v)
        ok = _down(ctx, d_v, n) == V and _down(ctx, d_g, k * n) == [x for v in g for x in v] and _down(ctx, d_h, k * n) == [x for v in h for x in v]
        for p_ in ptrs + [d_g, d_h, d_v]:
            ctx.free(p_)
        if not ok:
            raise SystemExit("permutation grand product differs: curve %d n %d k %d" % (curve, n, k))
    elif which == 1:
        if n < 2:
            return
        k_in, k_val = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        usable = int(rng.integers(0, n))
        inputs, values, sorted_ = [vec() for _ in range(k_in)], [vec() for _ in range(k_val)], [vec() for _ in range(max(1, k_in + k_val))]
        V = po.lookup_grand_product(inputs, values, sorted_, beta, gamma, usable, r)
        ptrs = [_up(ctx, v) for v in inputs + values + sorted_]
        d_v = ctx.malloc(n * 32)
        ctx.lookup_grand_product_dev(curve, ptrs[:k_in], ptrs[k_in:k_in + k_val], ptrs[k_in + k_val:], n, usable, limbs(beta, 4), limbs(gamma, 4), d_v)
        ok = _down(ctx, d_v, n) == V
        for p_ in ptrs + [d_v]:
            ctx.free(p_)
        if not ok:
            raise SystemExit("lookup grand product differs: curve %d n %d k_in %d k_val %d usable %d" % (curve, n, k_in, k_val, usable))
    else:
        x, y, z = vec(), vec(), [v or 1 for v in vec()]
        a, b, c = srng.next_mod(r), srng.next_mod(r), srng.next_mod(r)
        count = int(rng.integers(0, n + 1))
        d_x, d_y, d_z, d_o = _up(ctx, x), _up(ctx, y), _up(ctx, z), _up(ctx, [5] * n)
        ctx.fr_vec_mul_div_dev(curve, d_x, d_y, d_z, d_o, count)
        ok = _down(ctx, d_o, n) == [u * v % r * pow(w, -1, r) % r for u, v, w in zip(x[:count], y[:count], z[:count])] + [5] * (n - count)
        ctx.fr_vec_affine_dev(curve, d_x, d_y, limbs(a, 4), limbs(b, 4), limbs(c, 4), d_o, n)
        ok = ok and _down(ctx, d_o, n) == [(a * u + b * v + c) % r for u, v in zip(x, y)]
        for p_ in (d_x, d_y, d_z, d_o):
            ctx.free(p_)
        if not ok:
            raise SystemExit("vector affine / mul-div differs: curve %d n %d count %d" % (curve, n, count))
    stats["argument_kernels"] += 1


def fuzz_gate(zk, ctx, rng, stats):
    """zkhip_gate_eval_dev on a random flat program (gates with / without selector, 0 .. 40 terms of 0 .. 5 factors, rotations of both signs up to the
    domain's size, with / without mask, in one piece or two accumulating ones) against the same sum from the oracle's pointwise arithmetic"""
    curve = int(rng.integers(0, 2))
    log_size = int(rng.integers(1, 13 if SCALE == 1 else 15))
    size = 1 << log_size
    r = CURVES[curve].r
    n_slots = int(rng.integers(1, 7))
    cols = [cp.random_fr(curve, int(rng.integers(1, 1 << 30)), size) for _ in range(n_slots)]
    if rng.random() < 0.5:
        cols[0][:: int(rng.integers(2, 5))] = 0
    rot = lambda: int(rng.integers(-size, size + 1)) if rng.random() < 0.5 else int(rng.integers(-2, 3))
    gates = []
    for _ in range(int(rng.integers(1, 6))):
        sel = (int(rng.integers(0, n_slots)), rot()) if rng.random() < 0.6 else None
        terms = []
        for _ in range(int(rng.integers(0, 41 if rng.random() < 0.2 else 6))):
            coeff = int(rng.integers(0, 5)) if rng.random() < 0.2 else po.from_limbs(cp.random_fr(curve, int(rng.integers(1, 1 << 30)), 1)[0])
            terms.append((coeff % r, [(int(rng.integers(0, n_slots)), rot()) for _ in range(int(rng.integers(0, 6)))]))
        gates.append((sel, terms))
    from util import fr_arr

    def expect(gs, mask, prev):
        acc = np.zeros((size, 4), dtype=np.uint64) if prev is None else prev
        for sel, terms in gs:
            g = np.zeros((size, 4), dtype=np.uint64)
            for c, fs in terms:
                t = np.repeat(fr_arr([c]), size, axis=0)
                for sl, rt in fs:
                    t = cp.fr_vec(curve, 2, t, np.roll(cols[sl], -rt, axis=0))
                g = cp.fr_vec(curve, 0, g, t)
            if sel is not None:
                g = cp.fr_vec(curve, 2, g, np.roll(cols[sel[0]], -sel[1], axis=0))
            acc = cp.fr_vec(curve, 0, acc, g)
        return cp.fr_vec(curve, 2, acc, mask) if mask is not None else acc

    d_slots = [ctx.malloc(size * 32) for _ in range(n_slots)]
    for p_, c in zip(d_slots, cols):
        ctx.h2d(p_, c)
    mask = cp.random_fr(curve, int(rng.integers(1, 1 << 30)), size) if rng.random() < 0.6 else None
    d_mask = ctx.malloc(size * 32) if mask is not None else 0
    if mask is not None:
        ctx.h2d(d_mask, mask)
    d_out = ctx.malloc(size * 32)
    out = np.zeros((size, 4), dtype=np.uint64)
    cut = int(rng.integers(1, len(gates))) if len(gates) > 1 and rng.random() < 0.4 else 0
    if cut:
        ctx.gate_eval_dev(curve, gates[:cut], d_slots, log_size, d_out)
        ctx.gate_eval_dev(curve, gates[cut:], d_slots, log_size, d_out, d_mask, accumulate=True)
    else:
        ctx.gate_eval_dev(curve, gates, d_slots, log_size, d_out, d_mask)
    ctx.d2h(out, d_out)
    assert (out == expect(gates, mask, None)).all(), ("gate_eval", curve, log_size, gates)
    for p_ in d_slots + [d_out] + ([d_mask] if d_mask else []):
        ctx.free(p_)
    stats["gate_eval"] = stats.get("gate_eval", 0) + 1


def fuzz_group(zk, ctx, rng, stats):
    """the device group (members on this box's one GPU or dealt over its GPUs): zkhip_group_msm and zkhip_group_ntt over 1 .. 5 members and a random
    transport against the oracle"""
    import torch
    have = max(1, torch.cuda.device_count())
    world = int(rng.integers(1, 6))
    g = zk.DeviceGroup([k % have for k in range(world)])
    distinct = world <= have
    g.set_transport(int(rng.choice([zk.zkhip.GROUP_AUTO, zk.zkhip.GROUP_PEER, zk.zkhip.GROUP_STAGED] + ([zk.zkhip.GROUP_RCCL] if distinct else []))))
    curve = int(rng.integers(0, 2))
    if rng.random() < 0.6:
        group = int(rng.integers(1, 3))
        n = int(rng.integers(1, (3000 if group == 1 else 600) * SCALE))
        pts, inf = cp.batch_mul(curve, group, cp.random_fr(curve, int(rng.integers(1, 1 << 30)), n))
        gb = g.upload_bases(curve, group, pts)
        off = int(rng.integers(0, n)) if rng.random() < 0.3 else 0
        cnt = int(rng.integers(0, n - off + 1)) if rng.random() < 0.3 else n - off
        sc = scalars(rng, curve, cnt, ["uniform", "zeros_ones", "few_values"][int(rng.integers(0, 3))])
        aff, is_inf = g.msm_affine(gb, sc, offset=off, n=cnt)
        exp, einf = cp.msm(curve, group, pts[off:off + cnt], sc, chunks=2) if cnt else (None, 1)
        assert is_inf == einf and (einf or (aff == exp).all()), ("group msm", curve, group, n, off, cnt, world)
        gb.free()
    else:
        log_m, batch = int(rng.integers(1, 13)), int(rng.integers(1, 7))
        w = limbs(CURVES[curve].root_of_unity(log_m), 4)
        a = cp.random_fr(curve, int(rng.integers(1, 1 << 30)), batch << log_m).reshape(batch, 1 << log_m, 4)
        inverse = bool(rng.integers(0, 2))
        coset = limbs(CURVES[curve].fr_generator, 4) if rng.random() < 0.4 else None
        assert (g.ntt(curve, a, log_m, w, inverse=inverse, coset=coset) == cp.ntt(curve, a, log_m, w, inverse=inverse, coset=coset)).all(), ("group ntt", curve, log_m, batch, world)
    g.close()
    stats["device_group"] = stats.get("device_group", 0) + 1


def fuzz_lpc_group(zk, ctx, rng, stats):
    """lpc_commitment_scheme_hip::commit through the shim, on one context or over a device group of 2 .. 9 members: a ragged batch (runs of one or two
    sizes), random domain, fri step, slice size -- the leaves the streaming tree builder is handed against the oracle's precommit leaves"""
    import ctypes
    import torch

    if "lib" not in _SHIM:
        _SHIM["lib"] = ctypes.CDLL(os.path.join(ROOT, "tests", "cpp", "libshimtest.so"))
    shim = _SHIM["lib"]
    curve = int(rng.integers(0, 2))
    C = CURVES[curve]
    log_domain = int(rng.integers(2, 13))
    fri_step = int(rng.integers(1, min(5, log_domain) + 1))
    npolys = int(rng.integers(1, 12))
    small = int(rng.integers(1, log_domain + 1))
    big = int(rng.integers(small, log_domain + 1))
    cut = int(rng.integers(0, npolys + 1))
    logs = [small] * cut + [big] * (npolys - cut)
    world = 1 if rng.random() < 0.2 else int(rng.integers(2, 10))
    slice_elems = int(rng.integers(1, 4 * npolys << log_domain))
    D = 1 << log_domain
    evals = [cp.random_fr(curve, int(rng.integers(1, 1 << 30)), 1 << l) for l in logs]
    ext = []
    for e, l in zip(evals, logs):
        c = cp.ntt(curve, e.reshape(1, -1, 4), l, limbs(C.root_of_unity(l), 4), inverse=True)[0]
        full = np.zeros((1, D, 4), dtype=np.uint64)
        full[0, : 1 << l] = c
        ext.append(cp.ntt(curve, full, log_domain, limbs(C.root_of_unity(log_domain), 4))[0])
    want = cp.fri_leaves(ext, fri_step)
    out = np.zeros((npolys * D, 4), dtype=np.uint64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        rc = shim.shim_lpc_commit_leaves(curve, P(np.concatenate(evals)), ctypes.c_size_t(npolys), P(np.array(logs, dtype=np.uint64)), ctypes.c_size_t(log_domain),
                                         ctypes.c_size_t(fri_step), ctypes.c_size_t(slice_elems), P(out))
    finally:
        shim.shim_set_world(1)
    assert rc == 0 and np.array_equal(out, want), ("lpc group", curve, logs, log_domain, fri_step, world, slice_elems, rc)
    stats["lpc_commit_leaves"] = stats.get("lpc_commit_leaves", 0) + 1


def fuzz_kzg_group(zk, ctx, rng, stats):
    """both batched KZG schemes through the shim (commit + proof_eval: evaluations, quotient commitments) on a random layout -- 1 .. 6 polynomials in
    up to three batches, ragged sizes and point sets -- against the oracle, and, for a group of 2 .. 5 members, the same over the device group
    (columns dealt, quotient commitments cut by point range) against the single-device scheme inside the harness"""
    import ctypes
    import torch
    import test_gpu_shim as tgs

    if "lib" not in _SHIM:
        _SHIM["lib"] = ctypes.CDLL(os.path.join(ROOT, "tests", "cpp", "libshimtest.so"))
    shim = _SHIM["lib"]
    curve = int(rng.integers(0, 2))
    r = CURVES[curve].r
    log_n = int(rng.integers(3, 10))
    pool = [int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % r for _ in range(4)]
    npolys = int(rng.integers(1, 7))
    layout, batch = [], 0
    for p in range(npolys):
        if p and rng.random() < 0.4:
            batch += int(rng.integers(1, 3))
        npts = int(rng.integers(1, 4))
        pts = [pool[i] for i in rng.choice(4, size=npts, replace=False)]
        layout.append((batch, log_n + int(rng.integers(0, 2)), pts))
    world = 1 if rng.random() < 0.3 else int(rng.integers(2, 6))
    shim.shim_set_world(world)
    shim.shim_set_gpus(max(1, torch.cuda.device_count()))
    try:
        tgs._kzg_proof_eval_at_multipass_sizes(shim, curve, log_n, layout, int(rng.integers(0, 1 << 20)))
    except AssertionError:
        raise SystemExit("KZG scheme differs: curve %d layout %r world %d" % (curve, [(k, l, len(p)) for k, l, p in layout], world))
    finally:
        shim.shim_set_world(1)
    stats["kzg_schemes"] = stats.get("kzg_schemes", 0) + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scale", type=int, default=1, help="multiply the size ranges (larger, fewer cases)")
    a = ap.parse_args()
    global SCALE
    SCALE = max(1, a.scale)
    zk = bench.load_pkg()
    ctx = zk.Context(0)
    rng = np.random.default_rng(a.seed)
    stats = {"msm": 0, "ntt": 0, "domain": 0, "domain_skipped": 0, "witness_map": 0, "groth16_proof": 0, "argument_kernels": 0}
    legs = [fuzz_msm, fuzz_msm, fuzz_ntt, fuzz_domain, fuzz_witness, fuzz_proof, fuzz_arguments, fuzz_gate, fuzz_group, fuzz_lpc_group, fuzz_kzg_group]
    t0 = time.time()
    while time.time() - t0 < a.seconds:
        legs[int(rng.integers(0, len(legs)))](zk, ctx, rng, stats)
    ctx.close()
    print(json.dumps({"fuzz": "device path against the CPU oracle, bit for bit", "seed": a.seed, "scale": SCALE, "seconds": round(time.time() - t0, 1), "compared": stats,
                      "differences": 0}))


if __name__ == "__main__":
    main()
