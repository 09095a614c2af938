#!/usr/bin/env python3
"""MSM throughput over sizes 2^14 .. 2^24 (BLS12-381 G1, and G2 up to 2^22), scalars and bases resident, with a
size-independent parity property at every size: MSM(s1) + MSM(s2) == MSM(s1 + s2 mod r) (all on the device)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    zk = bench.load_pkg()
    ctx = zk.Context(0)
    rows = []
    for group, logs in ((zk.G1, (14, 16, 18, 20, 22, 24)), (zk.G2, (14, 16, 18, 20, 22))):
        jac = 3 * zk.coord_limbs(zk.BLS12_381, group) * 8
        for log_n in logs:
            n = 1 << log_n
            t0 = time.time()
            b = ctx.bases_from_scalars(zk.BLS12_381, group, bench.random_scalars(np, n, 1))
            t_setup = time.time() - t0
            s1, s2 = bench.random_scalars(np, n, 2), bench.random_scalars(np, n, 3)
            d1, d2, d3 = ctx.malloc(n * 32), ctx.malloc(n * 32), ctx.malloc(n * 32)
            ctx.h2d(d1, s1)
            ctx.h2d(d2, s2)
            ctx.fr_vec_op_dev(zk.BLS12_381, 0, d1, d2, d3, n)
            d_out = ctx.malloc(3 * jac)
            for i, d in enumerate((d1, d2, d3)):
                ctx.msm_dev(b, d, d_out + i * jac)
            ctx.jacobian_sum_dev(zk.BLS12_381, group, d_out, 2, d_out)  # slot 0 <- MSM(s1) + MSM(s2)
            res = np.zeros((3, jac // 8), dtype=np.uint64)
            ctx.d2h(res, d_out)
            a0, i0 = ctx.jacobian_to_affine(zk.BLS12_381, group, res[0])
            a2, i2 = ctx.jacobian_to_affine(zk.BLS12_381, group, res[2])
            ok = bool(i0 == i2 and (a0 == a2).all())
            ctx.sync()
            reps = 5 if log_n <= 22 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.msm_dev(b, d1, d_out)
            ctx.sync()
            ms = (time.perf_counter() - t0) / reps * 1e3
            rows.append({"group": "G1" if group == zk.G1 else "G2", "log_n": log_n, "ms": round(ms, 3), "Mpoints_per_s": round(n / ms / 1e3, 2),
                         "linearity_ok": ok, "bases_setup_s": round(t_setup, 2)})
            print(rows[-1], flush=True)
            for d in (d1, d2, d3, d_out):
                ctx.free(d)
            b.free()
    print(json.dumps({"workload": "BLS12-381 MSM size sweep, 1 MI355X, bases (with window tables) and scalars resident", "rows": rows}))


if __name__ == "__main__":
    main()
