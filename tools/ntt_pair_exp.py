"""NTT 2^22 x 8 with one / two polynomials per workgroup and tile widths 4 / 8 (experiment behind option "ntt_pair")."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
zk = bench.load_pkg(); ctx = zk.Context(0)
log_m, batch = 22, 8
r = bench.R_BLS
w = bench.lim(np, pow(7, (r - 1) >> log_m, r))
data = bench.random_scalars(np, batch << log_m, 3)
d = ctx.malloc(data.nbytes); ctx.h2d(d, data)
for pair, tile in ((0, 3), (1, 3), (1, 4), (2, 3), (1, 3), (1, 2), (1, 4), (0, 3), (2, 4)):
    ctx.set_option("ntt_pair", pair); ctx.set_option("ntt_tile_log", tile)
    ctx.ntt_dev(0, d, log_m, batch, w); ctx.sync()
    t = time.perf_counter()
    for _ in range(20): ctx.ntt_dev(0, d, log_m, batch, w)
    ctx.sync()
    print("pair", pair, "tile_log", tile, round((time.perf_counter() - t) / 20 * 1e3, 3), "ms", flush=True)
