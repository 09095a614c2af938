"""the tail of small MSMs with the group law over lane QUADS (fu_quad.hpp) against lane pairs: wall ms per MSM and the
tail kernels' HIP-event times, G1 2^12 .. 2^19 points, the same inputs, results compared."""
import sys, time, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
out = {}
for log_n in (12, 14, 15, 16, 17, 18, 19):
    n = 1 << log_n
    b = ctx.bases_from_scalars(0, 1, bench.random_scalars(np, n, 1))
    sc = bench.random_scalars(np, n, 2)
    d_s = ctx.malloc(sc.nbytes); ctx.h2d(d_s, sc); d_o = ctx.malloc(144)
    row, res = {}, []
    for quads in (0, 1, 0, 1):
        ctx.set_option("msm_tail_quads", quads)
        ctx.msm_dev(b, d_s, d_o); ctx.sync()
        t = time.perf_counter()
        for _ in range(20): ctx.msm_dev(b, d_s, d_o)
        ctx.sync(); dt = (time.perf_counter() - t) / 20 * 1e3
        ctx.profile_reset(); ctx.profile(True)
        for _ in range(5): ctx.msm_dev(b, d_s, d_o)
        ctx.sync(); ctx.profile(False)
        p = ctx.profile_dump()
        tail = {k: round(p[k][0] / 5, 4) for k in ("msm_bucket_red", "msm_window_sum", "msm_final") if k in p}
        jac = np.zeros((3, 6), dtype=np.uint64); ctx.d2h(jac, d_o)
        res.append(ctx.jacobian_to_affine(0, 1, jac)[0])
        row.setdefault("quads" if quads else "pairs", []).append({"ms": round(dt, 4), **tail})
    assert all((r == res[0]).all() for r in res), "quads and pairs disagree"
    out[log_n] = row
    print(log_n, row, flush=True)
    b.free(); ctx.free(d_s); ctx.free(d_o)
ctx.set_option("msm_tail_quads", 1)
print(json.dumps(out))
