"""Aggregate throughput of the 2^20-point G1 MSM with one and with two MSMs in flight (two contexts = two streams, the same resident
bases object uploaded once per context): how much of the single-stream time is latency that a second stream can fill."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

zk = bench.load_pkg()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
ctxs = [zk.Context(0), zk.Context(0)]
ks, sc = bench.random_scalars(np, n, 1), bench.random_scalars(np, n, 2)
state = []
for c in ctxs:
    b = c.bases_from_scalars(zk.BLS12_381, zk.G1, ks)
    d_s, d_o = c.malloc(sc.nbytes), c.malloc(144)
    c.h2d(d_s, sc)
    state.append((c, b, d_s, d_o))
res = {}
for tiles in (14, 12):
    for c in ctxs:
        c.set_option("msm_sort_tile_log", tiles)
    for inflight in (1, 2):
        use = state[:inflight]
        for c, b, d_s, d_o in use:
            c.msm_dev(b, d_s, d_o)
            c.sync()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            for c, b, d_s, d_o in use:
                c.msm_dev(b, d_s, d_o)
        for c, _, _, _ in use:
            c.sync()
        dt = time.perf_counter() - t0
        res["tiles 2^%d, %d in flight" % (tiles, inflight)] = {"ms_per_msm": round(dt / (reps * inflight) * 1e3, 3),
                                                              "Mpoints_per_s": round(reps * inflight * n / dt / 1e6, 1)}
outs = []
for c, b, d_s, d_o in state:  # the same point from both contexts (compared in affine: the order of additions inside a bucket, hence the
    o = np.zeros((3, 6), dtype=np.uint64)  # projective representative, depends on the sort's atomics)
    c.d2h(o, d_o)
    outs.append(c.jacobian_to_affine(zk.BLS12_381, zk.G1, o))
same = outs[0][1] == outs[1][1] and bool((outs[0][0] == outs[1][0]).all())
print(json.dumps({"workload": "BLS12-381 G1 MSM 2^%d, 1 MI355X, one vs two MSMs in flight (two contexts)" % log_n, "same_result": same, **res}))
