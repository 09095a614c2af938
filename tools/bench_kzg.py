#!/usr/bin/env python3
"""BASELINE config 5's commitment leg on one GPU: KZG commit of `cols` witness columns of 2^log_n rows
(placeholder's commit(VARIABLE_VALUES_BATCH): per column one inverse NTT + one G1 MSM against the resident SRS,
kzg.hpp:427-435 / kzg_v2.hpp:208-226).  Columns and SRS resident; one batched iNTT, then one MSM per column."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cols", type=int, default=50)
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    zk = bench.load_pkg()
    ctx = zk.Context(0)
    n = 1 << a.log_n
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    lim = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    omega = lim(pow(7, (r - 1) >> a.log_n, r))
    # SRS alpha^i G with alpha = 7 (placeholder.cpp:175): powers by repeated multiplication in Python ints
    alpha, x, pw = 7, 1, np.empty((n, 4), dtype=np.uint64)
    for i in range(n):
        pw[i] = lim(x)
        x = x * alpha % r
    t0 = time.time()
    srs = ctx.bases_from_scalars(zk.BLS12_381, zk.G1, pw)
    t_srs = time.time() - t0
    cols = bench.random_scalars(np, n * a.cols, 5).reshape(a.cols, n, 4)
    d = ctx.malloc(cols.nbytes)
    d_out = ctx.malloc(a.cols * 144)
    times = []
    for s in range(a.steps + 1):
        ctx.h2d(d, cols)
        if s == a.steps:
            ctx.profile_reset()
            ctx.profile(True)
        t0 = time.perf_counter()
        ctx.ntt_dev(zk.BLS12_381, d, a.log_n, a.cols, omega, inverse=True)
        ctx.msm_batch_dev([srs] * a.cols, [d + 32 * n * c for c in range(a.cols)], [d_out + 144 * c for c in range(a.cols)], ns=[n] * a.cols)
        ctx.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    ctx.profile(False)
    prof = ctx.profile_dump()
    best = min(times[1:])
    # proof_eval (kzg_v2.hpp:236-305) over the same columns, coefficient forms resident (left in d by the commit):
    # evaluate every column at two points, f = sum theta^i (f_i - U_i) / V, pi_1, L, pi_2 -- the device part of the
    # shim's kzg_commitment_scheme_v2_hip::proof_eval (the U_i corrections touch O(1) coefficients and are skipped here)
    pts = bench.random_scalars(np, 2, 9)
    th = bench.random_scalars(np, a.cols + 1, 10)
    d_f, d_l, d_pi = ctx.malloc(n * 32), ctx.malloc(n * 32), ctx.malloc(2 * 144)
    ptrs = [d + 32 * n * c for c in range(a.cols)]
    pe = []
    for s in range(a.steps + 1):
        if s == a.steps:
            ctx.profile_reset()
            ctx.profile(True)
        t0 = time.perf_counter()
        ctx.poly_eval_dev(zk.BLS12_381, d, n, a.cols, pts)
        ctx.poly_lincomb_dev(zk.BLS12_381, ptrs, [n] * a.cols, th[: a.cols], 1, d_f, n, False)
        ctx.poly_div_linear_dev(zk.BLS12_381, d_f, n, pts[0], d_f)
        ctx.poly_div_linear_dev(zk.BLS12_381, d_f + 32, n - 1, pts[1], d_f + 32)
        ctx.msm_dev(srs, d_f + 64, d_pi, 0, n - 2)
        ctx.poly_lincomb_dev(zk.BLS12_381, ptrs + [d_f + 64], [n] * a.cols + [n - 2], th, 1, d_l, n, False)
        ctx.poly_div_linear_dev(zk.BLS12_381, d_l, n, pts[0], d_l)
        ctx.msm_dev(srs, d_l + 32, d_pi + 144, 0, n - 1)
        ctx.sync()
        pe.append((time.perf_counter() - t0) * 1e3)
    ctx.profile(False)
    prof_pe = ctx.profile_dump()
    print(json.dumps({"workload": "KZG commit of %d columns x 2^%d rows (BLS12-381), 1 GPU, columns and SRS resident" % (a.cols, a.log_n),
                      "ms": [round(t, 2) for t in times], "columns_per_s": round(a.cols / best * 1e3, 2), "srs_setup_s": round(t_srs, 2),
                      "kernel_ms": {k: round(v[0], 2) for k, v in sorted(prof.items())},
                      "proof_eval": {"workload": "opening proof of the same %d columns at 2 points (device part of proof_eval)" % a.cols,
                                     "ms": [round(t, 2) for t in pe], "kernel_ms": {k: round(v[0], 2) for k, v in sorted(prof_pe.items())}}}))


if __name__ == "__main__":
    main()
