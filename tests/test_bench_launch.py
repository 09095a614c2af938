"""`python bench.py --gpus N` must really start N ranks (one process per GPU) when no launcher did: the parent spawns
`python -m torch.distributed.run` BEFORE touching a GPU.  Checked here without a GPU through --dry-run (gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus,split", [(2, "points"), (4, "windows")])
def test_bench_gpus_flag_launches_ranks(gpus, split):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--split", split, "--dry-run"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == gpus and line["exchange_ok"] and line["split"] == split
    assert sorted(sum(line["windows_per_rank"], [])) == list(range(16))


def test_bench_under_external_launcher():
    """the driver's way: torch.distributed.run starts the ranks, bench.py reads WORLD_SIZE / RANK from the environment"""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["exchange_ok"]


@pytest.mark.gpu
def test_bench_rccl_path_on_one_gpu():
    """The N > 1 code path with the real backend, as far as one GPU can take it: torch.distributed.run starts one rank, --force-dist
    makes it initialise RCCL, all-gather + fold the 144-byte partial sums every step, run the sharded Groth16 proof (864-byte
    all-gather) and the KZG commit with its columns dealt over the ranks (all-gather of the commitments) at world = 1.  stdout must carry exactly one JSON line, every leg verified."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29619", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-dist",
                          "--ntt-log-m", "20", "--no-pmc", "--no-cpu-baseline"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["verified"] is True and line["scaling"] == "weak"
    assert line["groth16_sharded"]["verified"] is True and line["groth16"]["verified"] is True
    assert line["kzg_sharded"]["verified"] is True and line["kzg"]["verified"] is True
    assert line["ntt_sharded"]["verified"] is True and line["ntt"]["verified"] is True
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["achieved"] > 0
