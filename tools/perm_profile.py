"""The grand-product kernels on their own (for rocprofv3 --kernel-trace --stats): python3 tools/perm_profile.py [log_n] [k]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
zk = bench.load_pkg()
ctx = zk.Context(0)
n = 1 << log_n
cols = []
for i in range(3 * k + 3):
    d = ctx.malloc(n * 32)
    ctx.h2d(d, bench.random_scalars(np, n, 100 + i))
    cols.append(d)
d_g, d_h, d_v = ctx.malloc(k * n * 32), ctx.malloc(k * n * 32), ctx.malloc(n * 32)
one = bench.lim(np, 12345)
for _ in range(10):
    ctx.perm_grand_product_dev(0, cols[:k], cols[k:2 * k], cols[2 * k:3 * k], n, one, one, d_g, d_h, d_v)
for _ in range(10):
    ctx.lookup_grand_product_dev(0, cols[:2], cols[2:3], cols[3:6], n, n - 4, one, one, d_v)
for _ in range(10):
    ctx.fr_vec_mul_div_dev(0, cols[0], cols[1], cols[2], d_v, n)
ctx.sync()
ctx.close()
