"""EC-NTT (Lagrange-basis transform of powers of tau) timing on one GPU."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
lim = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
for log_m in (12, 16):
    m = 1 << log_m
    b = ctx.bases_from_scalars(0, 1, bench.random_scalars(np, m, 1))
    pts, _ = b.download()
    jac = np.zeros((m, 18), dtype=np.uint64); jac[:, :12] = pts; jac[:, 12] = 1
    d = ctx.malloc(jac.nbytes); ctx.h2d(d, jac)
    w = lim(pow(7, (r - 1) >> log_m, r))
    ctx.ec_ntt_dev(0, 1, d, log_m, w, inverse=True); ctx.sync()
    t = time.perf_counter(); ctx.ec_ntt_dev(0, 1, d, log_m, w, inverse=True); ctx.sync()
    print("G1 EC-NTT 2^%d: %.1f ms" % (log_m, (time.perf_counter() - t) * 1e3), flush=True)
    b.free()
