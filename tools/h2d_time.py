"""H2D time of a 50 x 2^20 x 32 B column batch (what a host-resident KZG batch adds to the resident-columns figure)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

zk = bench.load_pkg()
ctx = zk.Context(0)
data = bench.random_scalars(np, 50 << 20, 5)
d = ctx.malloc(data.nbytes)
for k in range(3):
    t0 = time.perf_counter()
    ctx.h2d(d, data)
    ctx.sync()
    dt = time.perf_counter() - t0
    print("pageable numpy -> device: %.1f ms for %.2f GB = %.1f GB/s" % (dt * 1e3, data.nbytes / 1e9, data.nbytes / dt / 1e9))
