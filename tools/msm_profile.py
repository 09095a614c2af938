"""Per-kernel HIP-event times of one MSM configuration: python tools/msm_profile.py LOG_N C S [G1|G2] [SEGLOG]
(ZK_SORT_TILE_LOG=12|14 in the environment picks the sort tile shape: 12 is what contexts that share the GPU use)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

zk = bench.load_pkg()
ctx = zk.Context(0)
log_n, c, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
group = zk.G2 if len(sys.argv) > 4 and sys.argv[4] == "G2" else zk.G1
if len(sys.argv) > 5:
    ctx.set_option("msm_segment_log", int(sys.argv[5]))
n = 1 << log_n
ctx.set_option("msm_window_bits", c)
if "ZK_SORT_TILE_LOG" in os.environ:
    ctx.set_option("msm_sort_tile_log", int(os.environ["ZK_SORT_TILE_LOG"]))
ctx.set_option("msm_sets", S)
b = ctx.bases_from_scalars(zk.BLS12_381, group, bench.random_scalars(np, n, 1))
sc = bench.random_scalars(np, n, 2)
d_s = ctx.malloc(sc.nbytes)
ctx.h2d(d_s, sc)
d_o = ctx.malloc(3 * zk.coord_limbs(zk.BLS12_381, group) * 8)
for _ in range(2):
    ctx.msm_dev(b, d_s, d_o)
ctx.sync()
ctx.profile_reset()
ctx.profile(True)
reps = 5
for _ in range(reps):
    ctx.msm_dev(b, d_s, d_o)
ctx.sync()
ctx.profile(False)
p = ctx.profile_dump()
tot = sum(v[0] for v in p.values()) / reps
ctx.sync()
t0 = time.perf_counter()
for _ in range(20):
    ctx.msm_dev(b, d_s, d_o)
ctx.sync()
wall = (time.perf_counter() - t0) / 20 * 1e3
print("log_n %d c %d S %d: %.3f ms wall, %.3f ms kernels |" % (log_n, c, S, wall, tot), " ".join("%s %.3f" % (k.replace("msm_", ""), v[0] / reps) for k, v in sorted(p.items())))
