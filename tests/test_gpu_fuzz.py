"""A bounded slice of the randomised differential run (tests/fuzz_gpu.py): shapes the fixed tests do not enumerate, device against
oracle, bit for bit.  The long runs (minutes, several seeds) are kept under profiles/."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11])
def test_randomised_shapes_against_the_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "--seconds", "45", "--seed", str(seed)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])  # librccl announces itself on stdout at exit, after our line
    assert out["differences"] == 0 and out["compared"]["msm"] > 0 and out["compared"]["ntt"] > 0 and out["compared"]["witness_map"] > 0
