"""Host-side cost of one call of the small polynomial-layer entry points on 64-element vectors (launch + argument staging)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes
import bench
zk = bench.load_pkg()
ctx = zk.Context(0)
n = 64
d = [ctx.malloc(n * 32) for _ in range(4)]
for p in d:
    ctx.h2d(p, bench.random_scalars(np, n, 3))
one = bench.lim(np, 1)
def t(name, f, reps=200):
    for _ in range(10): f()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync()
    print("%-40s %.1f us per call" % (name, (time.perf_counter() - t0) / reps * 1e6))
t("fr_vec_op (no host copies)", lambda: ctx.fr_vec_op_dev(0, 0, d[0], d[1], d[2], n))
t("fr_vec_affine (scalars by value)", lambda: ctx.fr_vec_affine_dev(0, d[0], d[1], one, one, one, d[2], n))
t("fr_vec_prod (1 pageable H2D)", lambda: ctx.fr_vec_prod_dev(0, [d[0], d[1]], d[2], n))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
t("fri_fold (2 pageable H2D)", lambda: ctx.lib.zkhip_fri_fold_dev(ctx.h, 0, ctypes.c_void_p(d[0]), ctypes.c_size_t(6), P(one), P(one), ctypes.c_void_p(d[2])))
t("poly_lincomb (3 pageable H2D)", lambda: ctx.poly_lincomb_dev(0, [d[0], d[1]], [n, n], np.stack([one, one]), 1, d[2], n, False))
r = bench.R_BLS
w6, w7 = bench.lim(np, pow(7, (r - 1) >> 6, r)), bench.lim(np, pow(7, (r - 1) >> 7, r))
big = ctx.malloc(256 * 32)
t("ntt 2^6", lambda: ctx.ntt_dev(0, d[0], 6, 1, w6))
t("poly_resize 2^6 -> 2^7", lambda: ctx.poly_resize_dev(0, d[0], 6, 1, w6, big, 7, w7))
t("poly_shift", lambda: ctx.poly_shift_dev(d[0], 6, 1, d[2]))
t("poly_eval (1 point; returns to the host)", lambda: ctx.poly_eval_dev(0, d[0], n, 1, one.reshape(1, 4)))
t("poly_div_linear (returns the remainder)", lambda: ctx.poly_div_linear_dev(0, d[0], n, one, d[2]))
t("poly_div_vanishing", lambda: ctx.poly_div_vanishing_dev(0, big, 128, 64, d[2]))
t("fr_vec_mul_div", lambda: ctx.fr_vec_mul_div_dev(0, d[0], d[1], d[3], d[2], n))
t("perm_grand_product k=2", lambda: ctx.perm_grand_product_dev(0, [d[0], d[1]], [d[1], d[0]], [d[3], d[0]], n, one, one, 0, 0, d[2]))
t("lookup_grand_product", lambda: ctx.lookup_grand_product_dev(0, [d[0]], [d[1]], [d[3], d[0]], n, n - 3, one, one, d[2]))
t("fri_leaves", lambda: ctx.lib.zkhip_fri_leaves_dev(ctx.h, ctypes.c_void_p(d[0]), ctypes.c_size_t(6), ctypes.c_size_t(1), ctypes.c_size_t(1), ctypes.c_void_p(d[2])))
# small MSMs: the fixed cost of the launch chain
for log_n in (0, 6, 10, 12, 14, 16):
    nn = 1 << log_n
    ks = bench.random_scalars(np, nn, 11)
    b = ctx.bases_from_scalars(0, 1, ks)
    d_s = ctx.malloc(nn * 32)
    ctx.h2d(d_s, bench.random_scalars(np, nn, 12))
    d_o = ctx.malloc(3 * 6 * 8)
    t("G1 MSM of 2^%d points (resident)" % log_n, lambda: ctx.msm_dev(b, d_s, d_o, 0, nn), reps=50)
    b.free()
