"""Workload of bench.py's live PMC passes: the headline MSM (2^log_n BLS12-381 G1 points, 3 launches after a warm-up)
and one 2^22 x 8 NTT, nothing else -- run as `rocprofv3 --pmc <counter> -- python3 tools/pmc_child.py LOG_N`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (libzkhip.so shares torch's HIP runtime)

import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
zk = bench.load_pkg()
ctx = zk.Context(0)
n = 1 << log_n
bases = ctx.bases_from_scalars(zk.BLS12_381, zk.G1, bench.random_scalars(np, n, 1000))
sc = bench.random_scalars(np, n, 2000)
d_s, d_o = ctx.malloc(sc.nbytes), ctx.malloc(144)
ctx.h2d(d_s, sc)
for _ in range(4):
    ctx.msm_dev(bases, d_s, d_o, 0, n)
ctx.sync()
log_m, batch = 22, 8
data = bench.random_scalars(np, batch << log_m, 3)
d = ctx.malloc(data.nbytes)
ctx.h2d(d, data)
r = bench.R_BLS
ctx.ntt_dev(zk.BLS12_381, d, log_m, batch, bench.lim(np, pow(7, (r - 1) >> log_m, r)))
ctx.sync()
ctx.close()
