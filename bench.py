#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X: MSM Mpoints/s, BLS12-381 G1, 2^20 points.

One step = one pass of the hot path (zkhip_msm_dev: scalars and bases resident in HBM, Jacobian result
left in HBM) over one batch of synthetic input.  N = 1: BASELINE configs[1].  N > 1 (one process per GPU,
launched by torch.distributed.run): point-range sharding -- every rank owns 2^20 bases and scalars of a
job of N * 2^20 points, computes its partial sum, then one all-gather of the 144-byte partial results
over RCCL and an on-device fold (weak scaling; no other collective on the data path).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
LOG_N = 20
ALG_BYTES_PER_POINT = 128  # BLS12-381 G1: 96 B affine base + 32 B scalar, each read once
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def load_pkg():
    pkg_dir = os.path.join(ROOT, "crypto3-zk_amd")
    spec = importlib.util.spec_from_file_location("crypto3_zk_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["crypto3_zk_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def random_scalars(np, n, seed):
    """uniform in [0, r) by rejection from 255-bit draws; (n, 4) u64 canonical little-endian"""
    r_limbs = [0xffffffff00000001, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48]
    rng = np.random.default_rng(seed)
    out = np.empty((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    while todo.size:
        v = rng.integers(0, 1 << 64, size=(todo.size, 4), dtype=np.uint64)
        v[:, 3] &= np.uint64((1 << 63) - 1)
        lt = np.zeros(todo.size, dtype=bool)
        eq = np.ones(todo.size, dtype=bool)
        for k in (3, 2, 1, 0):
            lt |= eq & (v[:, k] < np.uint64(r_limbs[k]))
            eq &= v[:, k] == np.uint64(r_limbs[k])
        out[todo[lt]] = v[lt]
        todo = todo[~lt]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="points per GPU = 2^log_n (default: the BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-groth16", action="store_true", help="skip the Groth16 constraints/s leg (N = 1 only)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL all-gather + fold at N = 1 too (checks the N > 1 path on one GPU)")
    ap.add_argument("--no-kzg", action="store_true", help="skip the KZG commit / opening-proof leg (BASELINE config 5's commitment layer, N = 1 only)")
    ap.add_argument("--no-ntt", action="store_true", help="skip the NTT leg (BASELINE config 3, N = 1 only)")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: libzkhip.so must share torch's HIP runtime

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: zkhip has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,  # nccl == RCCL on ROCm
                                device_id=torch.device("cuda", local_rank))

    zk = load_pkg()
    ctx = zk.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    n = 1 << args.log_n
    # synthetic input: bases P_i = k_i * G (device fixed-base kernel), scalars uniform in [0, r); per-rank seeds
    ks = random_scalars(np, n, 1000 + rank)
    bases = ctx.bases_from_scalars(zk.BLS12_381, zk.G1, ks)
    scalars = random_scalars(np, n, 2000 + rank)
    d_scalars = torch.from_numpy(scalars.view(np.int64)).to(f"cuda:{local_rank}")
    d_out = torch.zeros(3 * 6, dtype=torch.int64, device=f"cuda:{local_rank}")
    d_total = torch.zeros(3 * 6, dtype=torch.int64, device=f"cuda:{local_rank}")

    from crypto3_zk_amd import dist as zd

    def fold(gathered, w):
        ctx.jacobian_sum_dev(zk.BLS12_381, zk.G1, gathered.data_ptr(), w, d_total.data_ptr())
        return d_total

    def step():
        ctx.msm_dev(bases, d_scalars.data_ptr(), d_out.data_ptr(), 0, n)
        # N > 1: one all-gather of the 144-byte partial sums over RCCL, then the on-device fold
        zd.allgather_fold(d_out, world, lambda o, i: dist.all_gather_into_tensor(o, i), fold, always=use_dist)

    def fence():
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile_reset()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = ctx.profile_dump()
    g16_sharded = None
    if use_dist and not args.no_groth16:
        # BASELINE config 4: ONE 2^20-constraint proof sharded over all ranks (every rank takes part in the exchange)
        g16_sharded = groth16_sharded_leg(np, torch, dist, rank, world, local_rank)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed / 1e6
        dom_ms, dom_cnt = prof.get("msm_bucket_acc", (0.0, 0))
        dom_avg_ms = dom_ms / max(1, dom_cnt)
        achieved = (ALG_BYTES_PER_POINT * n) / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        line = {
            "metric": "MSM Mpoints/sec, BLS12-381 G1 Pippenger, 2^%d points per GPU" % args.log_n,
            "value": round(value, 4),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (381-bit Montgomery Fq, 255-bit Fr)",
            "data": "synthetic",
            "config": {"workload": "BLS12-381 G1 Pippenger MSM, 2^%d random points/scalars per GPU, bases resident" % args.log_n,
                       "points_per_gpu": n, "parallelism": "point-range shard x%d + all-gather of partial sums" % world},
            "roofline": {"bound": "hbm", "kernel": "msm_bucket_acc", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "avg_launch_ms": round(dom_avg_ms, 4), "algorithmic_bytes_per_launch": ALG_BYTES_PER_POINT * n},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())},
        }
        line["roofline"]["traffic"] = pmc_traffic()
        if world == 1 and not args.no_ntt:
            line["ntt"] = ntt_leg(np, zk, ctx)
        if world == 1 and not args.no_groth16:
            line["groth16"] = groth16_leg(np)
        if g16_sharded is not None:
            line["groth16_sharded"] = g16_sharded
        if world == 1 and not args.no_kzg:
            line["kzg"] = kzg_leg(np, zk, ctx)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(np, bases)
        print(json.dumps(line), flush=True)
    fence()
    bases.free()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def pmc_traffic(prefix="msm_bucket_acc"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_msm_bench.json, collected by tools/pmc_collect.sh): 2 x FETCH_SIZE (gfx950 reports half the
    bytes of 16-B-per-lane reads; the factor reproduces the expected 16 x 2^20 x 128 B of point gathers + index reads
    to within 15 %) + WRITE_SIZE."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_msm_bench.json")
    try:
        k = json.load(open(path))["kernels"]
        name = next(n for n in k if n.startswith(prefix))
        return int((2 * k[name]["FETCH_SIZE"]["mean_per_launch"] + k[name]["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    except Exception:
        return None


def ntt_leg(np, zk, ctx, log_m=22, batch=8, steps=5):
    """BASELINE config 3: radix-2 NTT over BLS12-381 Fr, domain 2^22, batch of 8, device resident, in place."""
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    lim = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    omega = lim(pow(7, (r - 1) >> log_m, r))
    m = 1 << log_m
    data = random_scalars(np, batch * m, 3)
    d = ctx.malloc(data.nbytes)
    ctx.h2d(d, data)
    ctx.ntt_dev(zk.BLS12_381, d, log_m, batch, omega)  # warm-up: builds the twiddle tables
    ctx.sync()
    ctx.profile_reset()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.ntt_dev(zk.BLS12_381, d, log_m, batch, omega)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    ctx.profile(False)
    k_ms, k_cnt = ctx.profile_get("ntt_pass")
    ctx.free(d)
    alg = batch * m * 64  # one read + one write of every element per transform
    achieved = alg / (k_ms / steps * 1e-3) / 1e9
    return {"metric": "NTT elements/sec, BLS12-381 Fr, 2^%d x %d" % (log_m, batch), "value": round(batch * m / dt / 1e6, 2), "unit": "Melements/s",
            "ms_per_transform_batch": round(dt * 1e3, 4),
            "roofline": {"bound": "hbm", "kernel": "ntt_pass (x%d per transform)" % (k_cnt // steps), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": (lambda t: None if t is None else t * (k_cnt // steps))(pmc_traffic("ntt_pass")),  # per transform batch, like `achieved`
                         "algorithmic_bytes_per_transform_batch": alg}}


def groth16_leg(np, log_constraints=20, inputs=10, steps=3):
    """The other half of BASELINE.json's metric: Groth16 prove constraints/s on one GPU (config 4's single-GPU leg:
    M = 2^20, n = 10, domain 2^21) through the header-only shim, assignment H2D and result D2H included."""
    import ctypes
    import subprocess

    so = os.path.join(ROOT, "crypto3-zk_amd", "libzkhip_bench.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "crypto3-zk_amd"), "libzkhip_bench.so"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    r, g = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 7
    M = 1 << log_constraints
    m = 1
    while m < M + inputs + 1:
        m <<= 1
    lim = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    omega, coset = lim(pow(g, (r - 1) // m, r)), lim(g)
    times = np.zeros(steps, dtype=np.float64)
    setup = ctypes.c_double()
    prof = ctypes.create_string_buffer(16384)
    rc = lib.zkhip_bench_groth16(0, ctypes.c_size_t(M), ctypes.c_size_t(inputs), ctypes.c_uint64(1), steps, omega.ctypes.data_as(ctypes.c_void_p),
                                 coset.ctypes.data_as(ctypes.c_void_p), times.ctypes.data_as(ctypes.c_void_p), ctypes.byref(setup), prof,
                                 ctypes.c_size_t(16384))
    if rc != 0:
        return {"error": rc}
    best = float(times[1:].min())
    return {"metric": "Groth16 prove constraints/sec, BLS12-381, 2^%d constraints, 1 GPU" % log_constraints, "value": round(M / best * 1e3, 1),
            "unit": "constraints/s", "ms_per_proof": [round(float(x), 2) for x in times], "domain": m,
            "key": "synthetic (random multiples of the generators, resident)", "key_setup_ms": round(setup.value, 1)}


def groth16_sharded_leg(np, torch, dist, rank, world, local_rank, log_constraints=20, inputs=10, steps=3):
    """BASELINE config 4: one Groth16 proof (M = 2^20, n = 10) sharded over `world` GPUs, one process each: every rank holds
    a point-range slice of each query (r1cs_gg_ppzksnark_proving_key_hip(ctx, pk, dom, rank, world)), runs the witness
    map in full and its five partial MSMs; the only exchange is one RCCL all-gather of 864 bytes per rank per proof,
    after which every rank assembles the proof.  Strong scaling: the work of one proof is fixed."""
    import ctypes
    import subprocess

    so = os.path.join(ROOT, "crypto3-zk_amd", "libzkhip_bench.so")
    if rank == 0 and not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "crypto3-zk_amd"), "libzkhip_bench.so"], stdout=subprocess.DEVNULL)
    dist.barrier(device_ids=[local_rank])
    lib = ctypes.CDLL(so)
    r, g = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 7
    M = 1 << log_constraints
    m = 1
    while m < M + inputs + 1:
        m <<= 1
    lim = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    omega, coset = lim(pow(g, (r - 1) // m, r)), lim(g)
    dev = torch.device("cuda", local_rank)

    @ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
    def all_gather(mine, words, out):
        src = np.ctypeslib.as_array(ctypes.cast(mine, ctypes.POINTER(ctypes.c_uint64)), (words,))
        t = torch.from_numpy(src.view(np.int64).copy()).to(dev)
        gathered = torch.empty(world * words, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered, t)  # RCCL
        dst = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint64)), (world * words,))
        dst[:] = gathered.cpu().numpy().view(np.uint64)

    times = np.zeros(steps, dtype=np.float64)
    setup = ctypes.c_double()
    rc = lib.zkhip_bench_groth16_sharded(local_rank, ctypes.c_size_t(rank), ctypes.c_size_t(world), all_gather, 0, ctypes.c_size_t(M),
                                         ctypes.c_size_t(inputs), ctypes.c_uint64(1), steps, omega.ctypes.data_as(ctypes.c_void_p),
                                         coset.ctypes.data_as(ctypes.c_void_p), times.ctypes.data_as(ctypes.c_void_p), ctypes.byref(setup))
    t = torch.tensor(list(times) + [float(rc != 0)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # a proof is done when the slowest rank is
    t = t.cpu().numpy()
    if t[-1] != 0:
        return {"error": "a rank failed"}
    best = float(t[1:-1].min()) if steps > 1 else float(t[0])
    return {"metric": "Groth16 prove constraints/sec, BLS12-381, 2^%d constraints, ONE proof sharded over %d GPU(s)" % (log_constraints, world),
            "value": round(M / best * 1e3, 1), "unit": "constraints/s", "scaling": "strong", "ms_per_proof": [round(float(x), 2) for x in t[:-1]],
            "domain": m, "exchange": "one RCCL all-gather of 864 B per rank per proof",
            "key": "synthetic (random multiples of the generators), each rank holds 1/%d of every query" % world}


def kzg_leg(np, zk, ctx, log_n=20, cols=50, steps=2):
    """BASELINE config 5's commitment layer on one GPU: KZG commit of 50 witness columns of 2^20 rows (per column one
    inverse NTT + one G1 MSM against the resident SRS alpha^i G, alpha = 7 as placeholder.cpp:175; kzg_v2.hpp:208-226)
    and the device part of the batched opening proof of the same columns at two points (kzg_v2.hpp:236-305), the
    coefficient forms staying resident in between."""
    n = 1 << log_n
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    lim = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    omega = lim(pow(7, (r - 1) >> log_n, r))
    x, pw = 1, np.empty((n, 4), dtype=np.uint64)
    for i in range(n):  # alpha^i as canonical limbs
        pw[i, 0], pw[i, 1], pw[i, 2], pw[i, 3] = x & 0xFFFFFFFFFFFFFFFF, (x >> 64) & 0xFFFFFFFFFFFFFFFF, (x >> 128) & 0xFFFFFFFFFFFFFFFF, x >> 192
        x = x * 7 % r
    srs = ctx.bases_from_scalars(zk.BLS12_381, zk.G1, pw)
    data = random_scalars(np, n * cols, 5).reshape(cols, n, 4)
    d = ctx.malloc(data.nbytes)
    d_out = ctx.malloc(cols * 144)
    ptrs = [d + 32 * n * c for c in range(cols)]
    commit = []
    for _ in range(steps + 1):
        ctx.h2d(d, data)
        t0 = time.perf_counter()
        ctx.ntt_dev(zk.BLS12_381, d, log_n, cols, omega, inverse=True)
        ctx.msm_batch_dev([srs] * cols, ptrs, [d_out + 144 * c for c in range(cols)], ns=[n] * cols)
        ctx.sync()
        commit.append((time.perf_counter() - t0) * 1e3)
    pts = random_scalars(np, 2, 9)
    th = random_scalars(np, cols + 1, 10)
    d_f, d_l, d_pi = ctx.malloc(n * 32), ctx.malloc(n * 32), ctx.malloc(2 * 144)
    opening = []
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        ctx.poly_eval_dev(zk.BLS12_381, d, n, cols, pts)
        ctx.poly_lincomb_dev(zk.BLS12_381, ptrs, [n] * cols, th[:cols], 1, d_f, n, False)
        ctx.poly_div_linear_dev(zk.BLS12_381, d_f, n, pts[0], d_f)
        ctx.poly_div_linear_dev(zk.BLS12_381, d_f + 32, n - 1, pts[1], d_f + 32)
        ctx.msm_dev(srs, d_f + 64, d_pi, 0, n - 2)
        ctx.poly_lincomb_dev(zk.BLS12_381, ptrs + [d_f + 64], [n] * cols + [n - 2], th, 1, d_l, n, False)
        ctx.poly_div_linear_dev(zk.BLS12_381, d_l, n, pts[0], d_l)
        ctx.msm_dev(srs, d_l + 32, d_pi + 144, 0, n - 1)
        ctx.sync()
        opening.append((time.perf_counter() - t0) * 1e3)
    for p in (d, d_out, d_f, d_l, d_pi):
        ctx.free(p)
    srs.free()
    best = min(commit[1:])
    return {"metric": "KZG commit columns/sec, BLS12-381, %d columns x 2^%d rows, 1 GPU (columns and SRS resident)" % (cols, log_n),
            "value": round(cols / best * 1e3, 2), "unit": "columns/s", "ms_per_commit": [round(t, 2) for t in commit],
            "opening_proof_ms": [round(t, 2) for t in opening],
            "opening_proof": "device part of kzg_v2 proof_eval for the same %d columns at 2 points, coefficient forms resident" % cols}


def cpu_baseline(np, bases):
    """The oracle's BDLO12 Pippenger (the CPU restatement of algebra::multiexp with chunks = #threads, as
    prover.hpp:94-99) timed on this host on a bounded sample of the same workload.  Reported, not a target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cport as cp

    cores = cp.num_threads()
    sample = bases.n
    pts, inf = bases.download(0, sample)
    hb = cp.Bases(0, 1, pts, inf)
    sc = random_scalars(np, sample, 77)
    hb.msm(sc[: 1 << 16], chunks=cores)  # spin the thread pool up
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or time.perf_counter() - t0 < 10.0:  # ~10 s of wall time on all host cores
        hb.msm(sc, chunks=cores)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(reps * sample / dt / 1e6, 5), "unit": "Mpoints/s", "cores": cores, "kind": "port",
            "sample": "%d MSMs over all 2^%d points, chunks = %d OpenMP threads, %.1f s wall" % (reps, sample.bit_length() - 1, cores, dt)}


if __name__ == "__main__":
    main()
