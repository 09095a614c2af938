#!/usr/bin/env python3
"""One MSM spread over `world` GPUs, emulated rank by rank on ONE GPU: per-rank time of both partitions.
  strong: a fixed 2^log_n-point MSM cut `world` ways (BASELINE config 4's MSMs);  weak: world x 2^log_n points (bench.py --gpus).
  points  : rank g runs the whole pipeline over its n / world points (own bases object, window size chosen for that size)
  windows : rank g holds all points and the tables of windows {w : w mod world == g}
The slowest rank bounds the job (the exchange is one 144-byte all-gather either way).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def timed(ctx, bases, d_s, d_o, n, off=0, reps=4):
    ctx.msm_dev(bases, d_s, d_o, off, n)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        ctx.msm_dev(bases, d_s, d_o, off, n)
    ctx.sync()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    zk = bench.load_pkg()
    ctx = zk.Context(0)
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    out = {"workload": "BLS12-381 G1 MSM, per-rank ms of the slowest rank, emulated on one MI355X", "log_n": log_n, "strong": {}, "weak": {}}
    d_o = ctx.malloc(144)
    for mode in ("strong", "weak"):
        for world in (1, 2, 4, 8):
            total = (1 << log_n) * (world if mode == "weak" else 1)
            ks = bench.random_scalars(np, total, 1)
            sc = bench.random_scalars(np, total, 2)
            d_s = ctx.malloc(sc.nbytes)
            ctx.h2d(d_s, sc)
            per = total // world
            # points: every rank's slice has the same size and distribution: time rank 0's
            b = ctx.bases_from_scalars(zk.BLS12_381, zk.G1, ks[:per])
            t_points = timed(ctx, b, d_s, d_o, per)
            b.free()
            # windows: ranks own ceil(W / world) or floor(W / world) windows: time rank 0 (the most loaded)
            ctx.set_option("msm_shard_world", world)
            ctx.set_option("msm_shard_rank", 0)
            b = ctx.bases_from_scalars(zk.BLS12_381, zk.G1, ks)
            ctx.set_option("msm_shard_world", 1)
            t_windows = timed(ctx, b, d_s, d_o, total)
            b.free()
            ctx.free(d_s)
            out[mode][world] = {"points_ms": round(t_points, 3), "windows_ms": round(t_windows, 3)}
            print(mode, world, out[mode][world], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
