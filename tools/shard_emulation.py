#!/usr/bin/env python3
"""One MSM spread over `world` GPUs, emulated rank by rank on ONE GPU: per-rank time of both partitions.
  strong: a fixed 2^log_n-point MSM cut `world` ways (BASELINE config 4's MSMs);  weak: world x 2^log_n points (bench.py --gpus).
  points  : rank g runs the whole pipeline over its n / world points (own bases object, window size chosen for that size)
  windows : rank g holds all points and the tables of windows {w : w mod world == g}
The slowest rank bounds the job (the exchange is one 144-byte all-gather either way).  Prints one JSON object.
  `--proof`: the same for ONE Groth16 proof of 2^log_n constraints sharded `world` ways (BASELINE config 4): rank 0 and rank world - 1
  each generate their slices of the key and time process_partial (replicated witness map + five partial MSMs + download of the
  864 bytes); the exchange and the assembly (~0.5 ms of host arithmetic) are not in the figure."""
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


def proof_emulation(log_m):
    import ctypes

    lib = bench._bench_lib()
    r, g = bench.R_BLS, 7
    M, inputs, steps = 1 << log_m, 10, 4
    m = 1
    while m < M + inputs + 1:
        m <<= 1
    omega, coset = bench.lim(np, pow(g, (r - 1) // m, r)), bench.lim(np, g)
    out = {"workload": "ONE Groth16 proof, BLS12-381, 2^%d constraints, sharded `world` ways: per-rank ms of process_partial (witness map replicated + 5 partial "
                       "MSMs + download of the partial sums), emulated rank by rank on one MI355X" % log_m, "per_rank_ms": {}}
    lib.zkhip_bench_set_partial_only(1)
    for world in (1, 2, 4, 8):
        res = {}
        for rank in sorted({0, world - 1}):
            times = np.zeros(steps, dtype=np.float64)
            setup = ctypes.c_double()
            rc = lib.zkhip_bench_groth16_sharded(0, ctypes.c_size_t(rank), ctypes.c_size_t(world), None, 0, ctypes.c_size_t(M), ctypes.c_size_t(inputs),
                                                 ctypes.c_uint64(1), steps, omega.ctypes.data_as(ctypes.c_void_p), coset.ctypes.data_as(ctypes.c_void_p),
                                                 times.ctypes.data_as(ctypes.c_void_p), ctypes.byref(setup), None)
            assert rc == 0, rc
            res["rank%d" % rank] = round(float(times[1:].mean()), 3)
        out["per_rank_ms"][world] = dict(res, slowest=max(res.values()))
        print("proof", world, out["per_rank_ms"][world], flush=True)
    lib.zkhip_bench_set_partial_only(0)
    one = out["per_rank_ms"][1]["slowest"]
    out["speedup_vs_world_1"] = {w: round(one / v["slowest"], 2) for w, v in out["per_rank_ms"].items()}
    print(json.dumps(out))


def main():
    if "--proof" in sys.argv:
        sys.argv.remove("--proof")
        return proof_emulation(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
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
