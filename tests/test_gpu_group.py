"""Device group (include/zkhip.h, zkhip_group_*): N contexts behind one caller, the exchange inside the library.  On a one-GPU box every
member sits on device 0 (the group logic, the stream ordering and the transports PEER / STAGED are the same code as over N GPUs; RCCL runs
at group size 1, its constraint being pairwise distinct devices).  Bit-exact against the oracle and against the single-context entry
points."""
import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import CURVES, limbs, pt_from_limbs

pytestmark = pytest.mark.gpu


def _devices(zk, want):
    """`want` members over the GPUs this box has (round-robin: distinct devices when there are enough, device 0 repeated otherwise)"""
    import torch
    have = max(1, torch.cuda.device_count())
    return [k % have for k in range(want)]


@pytest.mark.parametrize("world,transport", [(1, "peer"), (2, "peer"), (4, "peer"), (3, "staged"), (4, "staged"), (1, "rccl"), (2, "auto"), (8, "auto")])
def test_group_all_gather_and_copy(zk, world, transport):
    g = zk.DeviceGroup(_devices(zk, world))
    kinds = {"auto": zk.zkhip.GROUP_AUTO, "rccl": zk.zkhip.GROUP_RCCL, "peer": zk.zkhip.GROUP_PEER, "staged": zk.zkhip.GROUP_STAGED}
    g.set_transport(kinds[transport])
    nbytes = 864
    rng = np.random.default_rng(world)
    mine = [rng.integers(0, 256, nbytes, dtype=np.uint8) for _ in range(world)]
    d_send = [c.malloc(nbytes) for c in g.members]
    d_recv = [c.malloc(nbytes * world) for c in g.members]
    for c, p, a in zip(g.members, d_send, mine):
        c.h2d(p, a)
    expect = np.concatenate(mine)
    # everyone receives
    g.all_gather(d_send, d_recv, nbytes)
    g.sync()
    assert g.transport() != zk.zkhip.GROUP_AUTO
    if transport == "auto" and len(set(g.devices)) < world:
        assert g.transport() == zk.zkhip.GROUP_PEER  # members share a device: RCCL is not an option
    for c, p in zip(g.members, d_recv):
        got = np.zeros(nbytes * world, dtype=np.uint8)
        c.d2h(got, p)
        assert (got == expect).all()
    # only member 0 receives; the others' buffers must stay untouched
    for c, p in zip(g.members, d_recv):
        c.h2d(p, np.full(nbytes * world, 0xA5, dtype=np.uint8))
    for c, p, a in zip(g.members, d_send, mine):
        c.h2d(p, a[::-1].copy())
    g.all_gather(d_send, [d_recv[0]] + [None] * (world - 1), nbytes)
    g.sync()
    got = np.zeros(nbytes * world, dtype=np.uint8)
    g.members[0].d2h(got, d_recv[0])
    assert (got == np.concatenate([a[::-1] for a in mine])).all()
    for c, p in list(zip(g.members, d_recv))[1:]:
        c.d2h(got, p)
        assert (got == 0xA5).all()
    # member-to-member copy, ordered after the source's stream
    if world > 1:
        g.copy(world - 1, d_recv[world - 1], 0, d_send[0], nbytes)
        g.sync()
        got = np.zeros(nbytes, dtype=np.uint8)
        g.members[world - 1].d2h(got, d_recv[world - 1])
        assert (got == mine[0][::-1]).all()
    g.close()


def test_strided_copy(zk, ctx):
    """zkhip_memcpy_2d_d2d_async: what the LPC scheme over a group packs a leaf owner's segments with -- rows of `width` bytes from one pitch
    to another, stream-ordered; a width beyond a pitch is refused, empty copies are no-ops"""
    rng = np.random.default_rng(5)
    for rows, width, sp, dp in ((7, 96, 160, 96), (1, 32, 32, 64), (33, 4096, 8192, 4096), (256, 32, 1 << 14, 32)):
        src = rng.integers(0, 256, rows * sp, dtype=np.uint8)
        d_src, d_dst = ctx.malloc(rows * sp), ctx.malloc(rows * dp)
        ctx.h2d(d_src, src)
        ctx.h2d(d_dst, np.full(rows * dp, 0x5A, dtype=np.uint8))
        ctx.copy_2d(d_dst, dp, d_src, sp, width, rows)
        got = np.zeros(rows * dp, dtype=np.uint8)
        ctx.d2h(got, d_dst)
        want = np.full((rows, dp), 0x5A, dtype=np.uint8)
        want[:, :width] = src.reshape(rows, sp)[:, :width]
        assert (got.reshape(rows, dp) == want).all()
        with pytest.raises(zk.ZkhipError):
            ctx.copy_2d(d_dst, dp, d_src, sp, min(sp, dp) + 1, rows)
        ctx.copy_2d(d_dst, dp, d_src, sp, 0, rows)
        ctx.copy_2d(d_dst, dp, d_src, sp, width, 0)
        ctx.free(d_src)
        ctx.free(d_dst)


def test_group_rccl_refuses_shared_devices(zk):
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a box where two members must share a GPU")
    g = zk.DeviceGroup([0, 0])
    with pytest.raises(zk.ZkhipError):
        g.set_transport(zk.zkhip.GROUP_RCCL)
    g.close()


@pytest.mark.parametrize("curve,group,n", [(0, 1, 5000), (1, 1, 3000), (0, 2, 700), (1, 2, 500)])
def test_group_msm_matches_oracle_and_single_context(zk, ctx, curve, group, n):
    """zkhip_group_msm over 1, 2, 3 and 4 members: the oracle's sum, and the affine point zkhip_msm gives on one context"""
    ks = cp.random_fr(curve, 900 + curve * 10 + group, n)
    pts, _ = cp.batch_mul(curve, group, ks)
    sc = cp.random_fr(curve, 901 + n, n)
    sc[3] = 0
    sc[4] = np.array([1, 0, 0, 0], dtype=np.uint64)
    exp, einf = cp.msm(curve, group, pts, sc, chunks=4)
    single = ctx.upload_bases(curve, group, pts)
    aff1, inf1 = ctx.msm_affine(single, sc)
    assert inf1 == einf and (aff1 == exp).all()
    for world, transport in ((1, zk.zkhip.GROUP_AUTO), (2, zk.zkhip.GROUP_PEER), (3, zk.zkhip.GROUP_STAGED), (4, zk.zkhip.GROUP_AUTO)):
        g = zk.DeviceGroup(_devices(zk, world))
        g.set_transport(transport)
        gb = g.upload_bases(curve, group, pts)
        assert [gb.member_first(k) for k in range(world)] == [k * (n // world) + min(k, n % world) for k in range(world)]
        aff, inf = g.msm_affine(gb, sc)
        assert inf == einf and (aff == exp).all(), (world, transport)
        # a sub-range that leaves some members without points, and the empty multiexp
        lo, cnt = n // 3, n // 5
        e2, i2 = cp.msm(curve, group, pts[lo:lo + cnt], sc[:cnt], chunks=2)
        a2, f2 = g.msm_affine(gb, sc[:cnt], offset=lo, n=cnt)
        assert f2 == i2 and (a2 == e2).all()
        a3, f3 = g.msm_affine(gb, sc[:0], offset=0, n=0)
        assert f3 == 1
        # bases generated on the devices, slice by slice
        gs = g.bases_from_scalars(curve, group, ks)
        a4, f4 = g.msm_affine(gs, sc)
        assert f4 == einf and (a4 == exp).all()
        gs.free()
        gb.free()
        g.close()
    single.free()


def test_group_msm_fewer_points_than_members(zk):
    g = zk.DeviceGroup(_devices(zk, 4))
    ks = cp.random_fr(0, 77, 3)
    pts, _ = cp.batch_mul(0, 1, ks)
    sc = cp.random_fr(0, 78, 3)
    gb = g.upload_bases(0, 1, pts)
    aff, inf = g.msm_affine(gb, sc)
    exp, einf = cp.msm(0, 1, pts, sc, chunks=1)
    assert inf == einf and (aff == exp).all()
    gb.free()
    g.close()


@pytest.mark.parametrize("curve,log_m,batch", [(0, 12, 8), (1, 10, 5), (0, 14, 3), (0, 9, 1)])
def test_group_ntt_bit_identical(zk, ctx, curve, log_m, batch):
    C = CURVES[curve]
    w = limbs(C.root_of_unity(log_m), 4)
    gen = limbs(C.fr_generator, 4)
    a = cp.random_fr(curve, 40 + log_m, batch << log_m).reshape(batch, 1 << log_m, 4)
    for inverse, coset in ((False, None), (True, None), (False, gen), (True, gen)):
        one = ctx.ntt(curve, a, log_m, w, inverse=inverse, coset=coset)
        for world in (1, 2, 4):
            g = zk.DeviceGroup(_devices(zk, world))
            got = g.ntt(curve, a, log_m, w, inverse=inverse, coset=coset)
            assert (got == one).all(), (world, inverse, coset is not None)
            g.close()
    assert (ctx.ntt(curve, a, log_m, w) == cp.ntt(curve, a, log_m, w)).all()


@pytest.mark.parametrize("torch_first", [False, True])
def test_group_rccl_matches_the_hip_runtime_in_use(torch_first):
    """A process may hold two ROCm installations: PyTorch wheels ship their own libamdhip64 / librccl next to /opt/rocm's, and whichever
    libamdhip64 is loaded first serves everybody.  The group must load the RCCL that belongs to the HIP runtime libzkhip.so is bound to (found by
    tests/fuzz_gpu.py: with libzkhip.so loaded BEFORE torch, torch's librccl -- same soname -- failed inside ncclCommInitAll): both load orders,
    each in a process of its own, RCCL transport at group size 1."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "%s"
        "zk = bench.load_pkg(); zk.load_library()\n"
        "%s"
        "g = zk.DeviceGroup([0]); g.set_transport(zk.zkhip.GROUP_RCCL)\n"
        "c = g.members[0]; a = c.malloc(864); b = c.malloc(864)\n"
        "x = np.arange(864, dtype=np.uint8); c.h2d(a, x); g.all_gather([a], [b], 864); g.sync()\n"
        "y = np.zeros(864, dtype=np.uint8); c.d2h(y, b); assert (x == y).all() and g.transport() == zk.zkhip.GROUP_RCCL\n"
        "g.close(); print('ok')\n"
    ) % (root, "import torch\n" if torch_first else "", "" if torch_first else "import torch\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("value,expect", [("0,0,0", [3, 3]), ("0", [0, 0]), ("0,4096", [-1, -1])])
def test_default_group_from_the_environment(value, expect):
    """ZKHIP_DEVICES=... makes the calling thread's DEFAULT device group (what the reference's static process(proving_key, x, w) and the
    multiexp policy spread over): three members on device 0; one device = no group; a device that does not exist fails at EVERY use (a
    misconfigured list must not quietly become one GPU)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, torch\n"
            "s = ctypes.CDLL(%r)\n"
            "print('sizes', s.shim_default_group_size(), s.shim_default_group_size())\n") % os.path.join(root, "tests", "cpp", "libshimtest.so")
    env = dict(os.environ, ZKHIP_DEVICES=value)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("sizes")][0]
    assert [int(x) for x in line.split()[1:]] == expect
