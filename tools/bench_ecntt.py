"""EC-NTT (Lagrange-basis transform of powers of tau) timing on one GPU:  python tools/bench_ecntt.py [log_m ...] [--g2] [--curve 1]"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
group = 2 if "--g2" in sys.argv else 1
curve = 1 if "--curve1" in sys.argv else 0
R = [0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001][curve]
GEN = [7, 5][curve]
CW = [6, 4][curve] * group
lim = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
for log_m in [int(a) for a in args] or (12, 16):
    m = 1 << log_m
    b = ctx.bases_from_scalars(curve, group, bench.random_scalars(np, m, 1) if curve == 0 else bench.random_scalars(np, m, 1) >> np.uint64(2))
    pts, _ = b.download()
    jac = np.zeros((m, 3 * CW), dtype=np.uint64); jac[:, :2 * CW] = pts; jac[:, 2 * CW] = 1
    d = ctx.malloc(jac.nbytes); ctx.h2d(d, jac)
    w = lim(pow(GEN, (R - 1) >> log_m, R))
    for inverse in (False, True):
        ctx.ec_ntt_dev(curve, group, d, log_m, w, inverse=inverse); ctx.sync()
        t = time.perf_counter(); ctx.ec_ntt_dev(curve, group, d, log_m, w, inverse=inverse); ctx.sync()
        print("curve %d G%d EC-NTT 2^%d %s: %.1f ms" % (curve, group, log_m, "inverse" if inverse else "forward", (time.perf_counter() - t) * 1e3), flush=True)
    b.free(); ctx.free(d)
