#!/usr/bin/env python3
"""The realistic gate-argument leg of bench.py on its own (GPU box): python3 tools/bench_gate_argument.py > profiles/rNN_gate_argument.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (first: libzkhip.so must share torch's HIP runtime)

import bench

print(json.dumps(bench.gate_argument_leg(np), indent=1))
