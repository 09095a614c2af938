"""MSM time against the window size c at several sizes (tables built for each c): input for zk_msm_auto_window."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
for log_n in (10, 12, 14, 16, 18, 19):
    n = 1 << log_n
    ks = bench.random_scalars(np, n, 1)
    sc = bench.random_scalars(np, n, 2)
    d_s = ctx.malloc(sc.nbytes); ctx.h2d(d_s, sc); d_o = ctx.malloc(144)
    row = {}
    for c in range(max(4, log_n - 6), min(16, log_n + 1) + 1):
        ctx.set_option("msm_window_bits", c)
        b = ctx.bases_from_scalars(0, 1, ks)
        ctx.msm_dev(b, d_s, d_o); ctx.sync()
        t = time.perf_counter()
        for _ in range(5): ctx.msm_dev(b, d_s, d_o)
        ctx.sync()
        row[c] = round((time.perf_counter() - t) / 5 * 1e3, 3)
        b.free()
    print(log_n, row, flush=True)
