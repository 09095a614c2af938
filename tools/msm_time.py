import sys, os, time, importlib.util
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT)
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
for group in (1, 2):
    for n in (65539, 1<<16):
        ks = bench.random_scalars(np, n, 1)
        t=time.time(); b = ctx.bases_from_scalars(0, group, ks); t_up=time.time()-t
        sc = bench.random_scalars(np, n, 2)
        ctx.msm(b, sc)
        ctx.profile_reset(); ctx.profile(True)
        t=time.time(); ctx.msm(b, sc); dt=time.time()-t
        ctx.profile(False)
        print("group",group,"n",n,"upload %.2fs"%t_up,"msm %.1f ms"%(dt*1e3), {k:round(v[0],2) for k,v in ctx.profile_dump().items()})
        # sub-range like the L query
        t=time.time(); ctx.msm(b, sc[:n-11], offset=11, n=n-11); print("   sub-range %.1f ms"%((time.time()-t)*1e3))
        b.free()
