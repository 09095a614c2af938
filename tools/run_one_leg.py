"""One placeholder leg of bench.py on its own (for rocprofv3): python3 tools/run_one_leg.py permutation|lookup|quotient|round|lpc"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "permutation"
sys.argv = ["bench.py"]
import numpy as np

import bench

leg = {"permutation": bench.permutation_leg, "lookup": bench.lookup_leg, "quotient": bench.quotient_leg, "round": bench.placeholder_round_leg, "lpc": bench.lpc_leg}[which]
print(json.dumps(leg(np)))
