import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
n = 1 << 20
ks = bench.random_scalars(np, n, 1)
b = ctx.bases_from_scalars(0, 1, ks)
sc = bench.random_scalars(np, n, 2)
d_s = ctx.malloc(sc.nbytes); ctx.h2d(d_s, sc)
d_o = ctx.malloc(144)
for seg in (0, 1, 2, 3):
    ctx.set_option("msm_segment_log", seg)
    ctx.msm_dev(b, d_s, d_o); ctx.sync()
    ctx.profile_reset(); ctx.profile(True)
    t = time.time()
    for _ in range(5): ctx.msm_dev(b, d_s, d_o)
    ctx.sync(); dt = (time.time() - t) / 5
    ctx.profile(False)
    pr = ctx.profile_dump()
    print("segment_log", seg, "%.3f ms" % (dt * 1e3), {k: round(v[0] / 5, 3) for k, v in pr.items() if k in ("msm_bucket_red", "msm_window_sum")})
