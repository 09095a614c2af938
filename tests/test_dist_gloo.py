"""The N > 1 path on CPU: world_size 2 over the gloo backend.  The sharding and the all-gather + fold logic of
crypto3-zk_amd/dist.py run for real; the per-rank partial MSM (a GPU kernel in production) is supplied by the
oracle here, so this checks the distributed plumbing, not the kernels (those are covered by -m gpu)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, out_q):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    import cport as cp

    zk = conftest.load_pkg()
    from crypto3_zk_amd import dist as zd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ks = cp.random_fr(0, 11, n)
    pts, _ = cp.batch_mul(0, 1, ks)
    sc = cp.random_fr(0, 12, n)
    lo, hi = zd.shard_range(n, rank, world)
    part, pinf = cp.msm(0, 1, pts[lo:hi], sc[lo:hi])  # stand-in for the rank's GPU MSM
    # Jacobian (X, Y, Z) with Z = 1 (or 0 for infinity), as the C ABI returns it
    jac = np.concatenate([part, np.array([0 if pinf else 1, 0, 0, 0, 0, 0], dtype=np.uint64)])
    t = torch.from_numpy(jac.view(np.int64).copy())

    def fold(gathered, w):
        acc, ainf = None, 1
        g = gathered.numpy().view(np.uint64).reshape(w, 18)
        for j in range(w):
            p, pi = cp.jac_to_affine(0, 1, g[j].reshape(3, 6))
            acc, ainf = (p, pi) if acc is None else cp.point_add(0, 1, acc, ainf, p, pi)
        return acc, ainf

    total, tinf = zd.allgather_fold(t, world, lambda o, i: dist.all_gather_into_tensor(o, i), fold)
    exp, einf = cp.msm(0, 1, pts, sc, chunks=2)
    ok = bool(tinf == einf and (total == exp).all())
    # the same exchange pipelined (the collective of step i under the work of step i + 1): asynchronous gloo collectives, three steps whose
    # partial sums differ (step s contributes its point only on the ranks >= s), results one step late and in order
    pipe = zd.AllgatherFoldPipeline(world, lambda o, i: dist.all_gather_into_tensor(o, i, async_op=True), fold, lambda count: torch.zeros(count, dtype=torch.int64))
    inf_t = torch.zeros(18, dtype=torch.int64)
    seen = []
    for s in range(3):
        seen.append(pipe.push(t if rank >= s else inf_t))
    seen.append(pipe.flush())
    ok = ok and seen[0] is None and pipe.flush() is None
    for s in range(3):
        members = [r for r in range(world) if r >= s]
        if members:
            lo_s, hi_s = zd.shard_range(n, members[0], world)[0], n
            e_s, i_s = cp.msm(0, 1, pts[lo_s:hi_s], sc[lo_s:hi_s], chunks=1)
        else:
            e_s, i_s = None, 1
        got, ginf = seen[s + 1]
        ok = ok and ginf == i_s and (i_s == 1 or bool((got == e_s).all()))
    # NTT batches shard by polynomial without any collective
    polys = zd.shard_polys(8, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, polys)
    ok = ok and sorted(sum(gathered, [])) == list(range(8))
    dist.barrier()
    dist.destroy_process_group()
    out_q.put((rank, ok))


def test_pipeline_defers_wait_and_fold_by_one_step():
    """AllgatherFoldPipeline on test doubles: the collective of step i is started at push(i), its wait + fold happen at push(i + 1) -- after the
    next collective was started --, receive buffers alternate, flush drains the last step; a synchronous all_gather (handle None) works too"""
    conf = __import__("conftest")
    conf.load_pkg()
    from crypto3_zk_amd import dist as zd

    log = []

    class Handle:
        def __init__(self, step):
            self.step = step

        def wait(self):
            log.append(("wait", self.step))

    step_of = {}

    def all_gather(out, inp):
        out[: inp.numel()] = inp
        out[inp.numel():] = inp * 10
        step_of[id(out)] = int(inp[0])
        log.append(("start", int(inp[0]), id(out)))
        return Handle(int(inp[0]))

    def fold(gathered, w):
        log.append(("fold", step_of[id(gathered)]))
        return int(gathered.view(w, -1).sum())

    pipe = zd.AllgatherFoldPipeline(2, all_gather, fold, lambda count: torch.zeros(count, dtype=torch.int64))
    outs = [pipe.push(torch.full((3,), s, dtype=torch.int64)) for s in (1, 2, 3)]
    outs.append(pipe.flush())
    assert outs == [None, 33, 66, 99] and pipe.flush() is None
    kinds = [(e[0], e[1]) for e in log]
    assert kinds == [("start", 1), ("start", 2), ("wait", 1), ("fold", 1), ("start", 3), ("wait", 2), ("fold", 2), ("wait", 3), ("fold", 3)]
    bufs = [e[2] for e in log if e[0] == "start"]
    assert bufs[0] != bufs[1] and bufs[0] == bufs[2]    # two receive buffers, alternating
    # synchronous collective
    pipe = zd.AllgatherFoldPipeline(2, lambda o, i: (o.__setitem__(slice(0, 3), i), o.__setitem__(slice(3, 6), i), None)[2], lambda g, w: int(g.sum()),
                                    lambda count: torch.zeros(count, dtype=torch.int64))
    assert [pipe.push(torch.full((3,), 5, dtype=torch.int64)), pipe.flush()] == [None, 30]


def test_shard_range():
    conf = __import__("conftest")
    conf.load_pkg()
    from crypto3_zk_amd import dist as zd

    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 3, 4, 8):
            ranges = [zd.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        zd.shard_range(10, 2, 2)
    for W in (16, 17, 3):
        for world in (1, 2, 4, 8, 19):
            owned = [zd.shard_windows(W, r, world) for r in range(world)]
            assert sorted(sum(owned, [])) == list(range(W))
            assert max(map(len, owned)) - min(map(len, owned)) <= 1


def test_point_range_sharding_world2_gloo():
    world, n = 2, 301
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8, 64])
def test_column_deal_and_gathered_layout(world):
    """bench.py's kzg_sharded leg: column c of a batch lives on rank c % world in slot c // world of that rank's padded block of
    the all-gathered commitments; every column is owned exactly once and no rank holds more than ceil(cols / world)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest

    conftest.load_pkg()
    from crypto3_zk_amd import dist as zd

    cols = 50
    slots = (cols + world - 1) // world
    owned = [zd.shard_polys(cols, g, world) for g in range(world)]
    assert sorted(sum(owned, [])) == list(range(cols))
    assert max(len(o) for o in owned) <= slots
    for c in range(cols):
        assert owned[c % world][c // world] == c
