#!/usr/bin/env python3
"""NTT timing on the GPU box (BASELINE config 3: BLS12-381 Fr, 2^22 x 8), device-resident, with the per-kernel
HIP-event profile.  Not the headline bench (bench.py); used to fill DESIGN.md's NTT roofline row."""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pkg():
    pkg_dir = os.path.join(ROOT, "crypto3-zk_amd")
    spec = importlib.util.spec_from_file_location("crypto3_zk_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["crypto3_zk_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-m", type=int, default=22)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--radix", type=int, default=0)
    ap.add_argument("--tile", type=int, default=-1)
    args = ap.parse_args()
    zk = load_pkg()
    ctx = zk.Context(0)
    if args.radix:
        ctx.set_option("ntt_radix_log", args.radix)
    if args.tile >= 0:
        ctx.set_option("ntt_tile_log", args.tile)
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    w = pow(7, (r - 1) >> args.log_m, r)
    omega = np.array([(w >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    g = np.array([7, 0, 0, 0], dtype=np.uint64)
    m = 1 << args.log_m
    rng = np.random.default_rng(1)
    data = rng.integers(0, 1 << 62, size=(args.batch, m, 4), dtype=np.uint64)
    d = ctx.malloc(data.nbytes)
    ctx.h2d(d, data)
    out = {}
    for name, kw in (("forward", {}), ("inverse", {"inverse": True}), ("coset_forward", {"coset": g})):
        ctx.ntt_dev(zk.BLS12_381, d, args.log_m, args.batch, omega, **kw)  # warm-up (tables)
        ctx.sync()
        ctx.profile_reset()
        ctx.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.ntt_dev(zk.BLS12_381, d, args.log_m, args.batch, omega, **kw)
        ctx.sync()
        dt = (time.perf_counter() - t0) / args.steps
        ctx.profile(False)
        prof = ctx.profile_dump()
        alg = args.batch * m * 64
        out[name] = {"ms": round(dt * 1e3, 4), "Melem_per_s": round(args.batch * m / dt / 1e6, 2), "algorithmic_GBps": round(alg / dt / 1e9, 2),
                     "kernel_ms": {k: round(v[0] / args.steps, 4) for k, v in prof.items()}, "launches": {k: v[1] // args.steps for k, v in prof.items()}}
    print(json.dumps({"workload": "NTT BLS12-381 Fr 2^%d x %d" % (args.log_m, args.batch), "results": out}))
    ctx.free(d)
    ctx.close()


if __name__ == "__main__":
    main()
