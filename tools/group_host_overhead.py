#!/usr/bin/env python3
"""Host-side cost of ONE proof over a device group as the member count grows, measured where only one GPU exists: 1, 2, 4, 8 members on device 0
(their kernels queue on the one device, so ms_per_proof is NOT a scaling figure; host_phase_ms.launches_all_members -- the assignment's one upload,
its broadcast and the members' launches, from threads of their own beyond two members -- is what a real N-GPU group pays before its last member starts).
python3 tools/group_host_overhead.py > profiles/rNN_group_host_overhead.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import bench

out = {}
for members in (1, 2, 4, 8):
    leg = bench.groth16_group_leg(np, [0] * members, steps=4)
    out[str(members)] = {k: leg.get(k) for k in ("ms_per_proof_mean", "host_phase_ms", "transport", "verified", "key_setup_ms", "error")}
print(json.dumps(out, indent=1))
