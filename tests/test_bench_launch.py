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
    lines = lines_of(out.stdout)
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == gpus and line["exchange_ok"] and line["split"] == split
    assert sorted(sum(line["windows_per_rank"], [])) == list(range(16))
    d = line["dist"]    # one device id per rank, gathered over the process group: what shows that a SCALE run saw N distinct GPUs
    assert d["world_size"] == gpus and len(d["devices"]) == gpus and d["distinct_devices"] == gpus


def test_bench_under_external_launcher():
    """the driver's way: torch.distributed.run starts the ranks, bench.py reads WORLD_SIZE / RANK from the environment"""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["exchange_ok"]
    assert len(lines_of(out.stdout)[0]) <= 6144 and "roofline" in line and "legs" in line  # the real line's frame, within the driver's parser


def lines_of(stdout):
    return [l for l in stdout.splitlines() if l.startswith("{")]


def test_compact_line_fits_the_drivers_parser():
    """the driver keeps an 8 KiB tail of the output and round 4's 23.6 KB line was not parsed.  bench.py now prints
    compact_line(full) and writes `full` to a sidecar; held here against the largest full line on record (round 4's, every leg present,
    with every optional part filled in) -- contract fields, roofline and cpu_baseline must survive, within LINE_LIMIT."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("zk_bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_final.json")))
    full["verified_vs"] = ["(sum s_i k_i) G by the fixed-base kernel", "oracle (cport) BDLO12 MSM of the same points and scalars, affine, bit-exact"]
    full["timing"] = {"regime": "x"}
    full["host_scalars"] = {"value": 300.123, "unit": "Mpoints/s", "ms_per_msm": 3.4944, "verified": True, "what": "y" * 300}
    full["dist"] = {"backend": "rccl", "world_size": 8, "devices": ["GPU-%032x" % i for i in range(8)], "distinct_devices": 8, "same_device_flag": False}
    for k in ("groth16_sharded", "kzg_sharded", "ntt_sharded"):
        full[k] = {"value": 1.0, "unit": "u", "verified": True, "scaling": "strong", "ms_per_proof_mean": 1.0}
    line = b.compact_line(full, "bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= b.LINE_LIMIT <= 6144, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["groth16_constraints_per_s"] == full["groth16"]["value"] and line["legs"]["placeholder_round"]["verified"] is True


@pytest.mark.gpu
def test_bench_rccl_path_on_one_gpu(tmp_path):
    """The N > 1 code path with the real backend, as far as one GPU can take it: torch.distributed.run starts one rank, --force-dist
    makes it initialise RCCL, all-gather + fold the 144-byte partial sums every step, run the sharded Groth16 proof (864-byte
    all-gather) and the KZG commit with its columns dealt over the ranks (all-gather of the commitments) at world = 1.  stdout must carry exactly one JSON line, every leg verified."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29619", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-dist",
                          "--ntt-log-m", "20", "--no-pmc", "--no-cpu-baseline", "--detail", str(tmp_path / "detail.json")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) <= 6144, len(lines[0])  # what the driver's parser holds
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["verified"] is True and line["scaling"] == "weak"
    legs = line["legs"]
    assert legs["groth16_sharded"]["verified"] is True and legs["groth16"]["verified"] is True
    assert legs["kzg_sharded"]["verified"] is True and legs["kzg"]["verified"] is True
    assert legs["ntt_sharded"]["verified"] is True and legs["ntt"]["verified"] is True
    assert all(l.get("verified") is True for l in legs.values()), legs
    assert line["groth16_constraints_per_s"] == legs["groth16"]["value"] > 0 and line["host_scalars"]["verified"] is True
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["achieved"] > 0
    assert line["dist"]["backend"] == "rccl" and line["dist"]["world_size"] == 1 and len(line["dist"]["devices"]) == 1
    detail = json.load(open(tmp_path / "detail.json"))   # the sidecar keeps every leg's full object
    assert detail["value"] == line["value"] and "kernel_ms_per_step" in detail and "ms_by_phase" in detail["placeholder_round"]


def test_group_child_failure_costs_the_leg_not_the_line(tmp_path):
    """bench.py runs the device-group legs in a CHILD process (a group over several distinct GPUs is RCCL single-process over xGMI, which no
    round could run on hardware): whatever happens there -- here: no GPU at all, so the group cannot even be made; then a command that dies;
    then one that hangs past its timeout -- must come back as an `error` entry of the leg, never as an exception in the parent."""
    import importlib.util
    import types

    spec = importlib.util.spec_from_file_location("zk_bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    args = types.SimpleNamespace(log_constraints=10, kzg_log_rows=10, no_verify=True)
    import torch
    if not torch.cuda.is_available():
        out = b.run_group_child([0, 0], ["groth16"], args, timeout=300)
        assert "error" in out["groth16_device_group"], out          # the child ran, the leg failed inside it (zkhip has no CPU fallback)
    real = b.sys.executable
    try:
        b.sys.executable = "/bin/false"                              # a child that dies at once
        out = b.run_group_child([0, 1], ["groth16", "kzg"], args, timeout=30)
        assert set(out) == {"groth16_device_group", "kzg_device_group"} and all("error" in v for v in out.values())
        hang = tmp_path / "hang.sh"                                  # ... and one that never answers: the timeout ends it
        hang.write_text("#!/bin/sh\nexec sleep 120\n")
        hang.chmod(0o755)
        b.sys.executable = str(hang)
        out = b.run_group_child([0, 1], ["groth16"], args, timeout=2)
        assert "error" in out["groth16_device_group"] and "timed out" in out["groth16_device_group"]["error"]
    finally:
        b.sys.executable = real
