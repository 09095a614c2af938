"""MSM time against the window size c and the number of bucket sets S at several sizes (tables rebuilt for every c):
the input of zk_msm_auto_window / zk_msm_target_lanes.  Usage: python tools/msm_window_sweep.py [G1|G2] [log_n ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

zk = bench.load_pkg()
ctx = zk.Context(0)
group = zk.G2 if len(sys.argv) > 1 and sys.argv[1] == "G2" else zk.G1
logs = [int(x) for x in sys.argv[2:]] or [10, 12, 14, 16, 18, 20, 21]
jac = 3 * zk.coord_limbs(zk.BLS12_381, group) * 8
out = {}
for log_n in logs:
    n = 1 << log_n
    ks = bench.random_scalars(np, n, 1)
    sc = bench.random_scalars(np, n, 2)
    d_s = ctx.malloc(sc.nbytes)
    ctx.h2d(d_s, sc)
    d_o = ctx.malloc(jac)
    row = {}
    for c in range(max(4, log_n - 3), min(21, log_n + 2) + 1):
        ctx.set_option("msm_window_bits", c)
        b = ctx.bases_from_scalars(zk.BLS12_381, group, ks)
        W = (255 + c - 1) // c
        for S in sorted({1, 2, 4, 8, W}):
            if S > W or (S << (c - 1)) > (1 << 20):
                continue
            ctx.set_option("msm_sets", S)
            ctx.msm_dev(b, d_s, d_o)
            ctx.sync()
            reps = 5 if log_n <= 20 else 3
            t = time.perf_counter()
            for _ in range(reps):
                ctx.msm_dev(b, d_s, d_o)
            ctx.sync()
            row["c%d/S%d" % (c, S)] = round((time.perf_counter() - t) / reps * 1e3, 3)
        b.free()
    ctx.set_option("msm_sets", 0)
    ctx.set_option("msm_window_bits", 0)
    ctx.free(d_s)
    ctx.free(d_o)
    best = min(row, key=row.get)
    out[log_n] = {"best": best, "ms": row[best], "all": row}
    print(log_n, best, row[best], row, flush=True)
print(json.dumps({"workload": "BLS12-381 %s MSM, window size c x bucket sets S sweep, 1 MI355X" % ("G2" if group == zk.G2 else "G1"), "rows": out}))
