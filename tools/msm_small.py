"""Per-kernel times of G1 MSMs at small sizes (where the fixed costs dominate)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
for log_n in (10, 12, 14, 16, 18):
    n = 1 << log_n
    b = ctx.bases_from_scalars(0, 1, bench.random_scalars(np, n, 1))
    sc = bench.random_scalars(np, n, 2)
    d_s = ctx.malloc(sc.nbytes); ctx.h2d(d_s, sc); d_o = ctx.malloc(144)
    ctx.msm_dev(b, d_s, d_o); ctx.sync()
    ctx.profile_reset(); ctx.profile(True)
    t = time.perf_counter()
    for _ in range(5): ctx.msm_dev(b, d_s, d_o)
    ctx.sync(); dt = (time.perf_counter() - t) / 5
    ctx.profile(False)
    for _ in range(3): ctx.msm_dev(b, d_s, d_o)   # identical calls: the third is captured into a graph
    ctx.sync()
    t = time.perf_counter()
    for _ in range(10): ctx.msm_dev(b, d_s, d_o)
    ctx.sync(); dg = (time.perf_counter() - t) / 10
    ctx.set_option("msm_graphs", 0)
    t = time.perf_counter()
    for _ in range(10): ctx.msm_dev(b, d_s, d_o)
    ctx.sync(); dn = (time.perf_counter() - t) / 10
    ctx.set_option("msm_graphs", 1)
    print(log_n, "graph %.3f ms, direct %.3f ms;" % (dg * 1e3, dn * 1e3), "profiled:", "%.3f ms" % (dt * 1e3), {k: (round(v[0] / 5, 3), v[1] // 5) for k, v in ctx.profile_dump().items()}, flush=True)
    b.free()
