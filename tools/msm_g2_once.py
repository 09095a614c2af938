"""One 2^20 BLS12-381 G2 MSM (after a warm-up) for counter collection."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
n = 1 << 20
b = ctx.bases_from_scalars(0, 2, bench.random_scalars(np, n, 1))
sc = bench.random_scalars(np, n, 2)
ctx.msm(b, sc)
ctx.msm(b, sc)
