"""Two REAL rank processes over the real kernels on the one GPU a test box has (VERDICT r2 "Missing" #2): torch.distributed.run
starts two ranks of bench.py, both mapped to device 0 (--same-device); the tiny exchanges -- the 144-byte partial sums of every
MSM step, the 864-byte partial sums of the sharded proof, the dealt KZG commitments -- go over gloo (RCCL cannot place two ranks
on one device; with more GPUs the same code path runs over RCCL, which tests/test_bench_launch.py covers at world = 1).  Every
leg's result is checked inside bench.py (`verified`): the MSM against (sum s_i k_i) G, the sharded proof against the trapdoor
identity on every rank, every gathered commitment against f(alpha) G."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(port, extra, tmp_path, ranks=2, log_n=18, ntt_log_m=18):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--dist-backend", "gloo", "--same-device", "--steps", "3", "--warmup", "1", "--log-n", str(log_n),
           "--no-cpu-baseline", "--no-pmc", "--ntt-log-m", str(ntt_log_m), "--detail", str(tmp_path / "detail.json")] + extra
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    assert len(lines[0]) <= 6144
    return json.loads(lines[0]), json.load(open(tmp_path / "detail.json"))


def test_two_ranks_point_split_and_sharded_legs(tmp_path):
    """point-range partition of the MSM, ONE Groth16 proof sharded over the two ranks (each generates and holds half of every query
    of the same valid key, over the step radix-2 domain the reference picks for 2^16 + 11), the 50 KZG columns dealt over the ranks"""
    line, detail = _run(29641, ["--split", "points", "--log-constraints", "16", "--kzg-log-rows", "16"], tmp_path)
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert "point-range partition x2" in line["config"]["parallelism"]
    assert all(l.get("verified") is True for l in line["legs"].values()), line["legs"]
    d = line["dist"]    # two ranks, ONE device (--same-device): the line says so
    assert d["backend"] == "gloo" and d["world_size"] == 2 and len(d["devices"]) == 2 and d["distinct_devices"] == 1 and d["same_device_flag"]
    g = detail["groth16_sharded"]
    assert g["verified"] is True and len(g["ms_per_proof"]) >= 2
    k = detail["kzg_sharded"]
    assert k["verified"] is True and k["columns_per_rank"] == 25
    kg = detail["kzg_device_group"]    # and rank 0's device group over the same two "GPUs" (one process, the drop-in scheme class): 50 columns dealt 25 / 25
    assert kg["verified"] is True and kg["members"] == 2 and line["legs"]["kzg_device_group"]["verified"] is True
    lg = detail["lpc_device_group"]    # ... and the LPC scheme over the same group: 16 polynomials dealt 8 / 8, two leaf owners
    assert lg["verified"] is True and lg["members"] == 2 and lg["leaf_owners"] == 2 and line["legs"]["lpc_device_group"]["verified"] is True
    t = detail["ntt_sharded"]    # BASELINE config 3's split: 8 polynomials dealt 4 / 4, no collective in the data path
    assert t["verified"] is True and t["polynomials_per_rank"] == 4 and t["scaling"] == "strong"


def test_two_ranks_window_split(tmp_path):
    """north_star's bucket-window shard: every rank holds all points and the window tables {w : w mod 2 == rank}"""
    line, _ = _run(29643, ["--split", "windows", "--no-groth16", "--no-kzg"], tmp_path)
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert "window partition x2" in line["config"]["parallelism"]


@pytest.mark.parametrize("split", ["points", "windows"])
def test_eight_ranks_the_shape_of_the_scale_run(tmp_path, split):
    """the launch the driver's SCALE run makes at N = 8 -- eight rank processes of bench.py with every sharded leg -- at
    small sizes on the one GPU of the test box (--same-device, gloo): rank 0 prints exactly ONE line, dist.world_size == 8, every leg
    verified.  13 windows over 8 ranks (the window split): ranks own two windows or one.  The point split deals 2^14 points per rank,
    50 KZG columns as 7 / 7 / 6 ..., 8 polynomials one per rank, and the proof's queries in eighths; and rank 0 drives a device group
    of eight members (the drop-in class's own multi-GPU path) while the other ranks wait on the host."""
    extra = ["--split", split, "--log-constraints", "12", "--kzg-log-rows", "12"]
    if split == "windows":
        extra += ["--no-kzg"]
    line, detail = _run(29651 if split == "points" else 29653, extra, tmp_path, ranks=8, log_n=14, ntt_log_m=14)
    assert line["n_gpus"] == 8 and line["verified"] is True
    d = line["dist"]
    assert d["world_size"] == 8 and len(d["devices"]) == 8 and d["distinct_devices"] == 1 and d["same_device_flag"]
    assert all(l.get("verified") is True for l in line["legs"].values()), line["legs"]
    assert detail["groth16_sharded"]["verified"] is True and detail["ntt_sharded"]["polynomials_per_rank"] == 1
    g = detail["groth16_device_group"]
    assert g["verified"] is True and g["members"] == 8 and g["distinct_gpus"] == 1
    if split == "points":
        assert detail["kzg_sharded"]["verified"] is True and detail["kzg_sharded"]["columns_per_rank"] in (6, 7)
        assert "point-range partition x8" in line["config"]["parallelism"]
    else:
        assert "window partition x8" in line["config"]["parallelism"]
