"""Key loading from the wire form: time of zkhip_bases_upload_compressed (H2D of the blob + device decoding + window tables)
against zkhip_bases_upload of the same points as affine limbs."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import bench
import pyoracle as po
zk = bench.load_pkg()
ctx = zk.Context(0)
p = po.BLS12_381.p
for group, log_n in ((1, 18), (2, 16)):
    n = 1 << log_n
    b = ctx.bases_from_scalars(0, group, bench.random_scalars(np, n, 1))
    pts, inf = b.download()
    b.free()
    L = 6
    blob = bytearray()
    for i in range(n):
        c = [po.from_limbs(pts[i, k * L:(k + 1) * L]) for k in range(2 * group)]
        P = (c[0], c[1]) if group == 1 else ((c[0], c[1]), (c[2], c[3]))
        blob += po.bls12_381_compress(group, P)
    blob = bytes(blob)
    ctx.upload_bases_compressed(0, group, blob, n).free()
    t = time.perf_counter(); bb = ctx.upload_bases_compressed(0, group, blob, n); t_c = time.perf_counter() - t
    got, _ = bb.download(); assert (got == pts).all(); bb.free()
    t = time.perf_counter(); bu = ctx.upload_bases(0, group, pts, inf); t_u = time.perf_counter() - t; bu.free()
    print("G%d 2^%d points: compressed %.1f ms (%.1f Mpoints/s), affine limbs %.1f ms" % (group, log_n, t_c * 1e3, n / t_c / 1e6, t_u * 1e3), flush=True)
