"""Per-kernel times of single MSMs at the sizes a 2^20-constraint Groth16 proof uses (diagnostic)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
for group, n in ((1, (1 << 20) - 8), (1, (1 << 21) - 1), (1, (1 << 20) + 3), (2, (1 << 20) + 3)):
    ks = bench.random_scalars(np, n, 1)
    b = ctx.bases_from_scalars(0, group, ks)
    sc = bench.random_scalars(np, n, 2)
    ctx.msm(b, sc)
    ctx.profile_reset(); ctx.profile(True)
    t = time.time(); ctx.msm(b, sc); dt = time.time() - t
    ctx.profile(False)
    print("group", group, "n", n, "msm %.1f ms" % (dt * 1e3), {k: round(v[0], 2) for k, v in ctx.profile_dump().items()})
    b.free()
