"""The placeholder legs of bench.py on their own (round, lookup argument, permutation argument, quotient chain): one JSON line each."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv=['bench.py']
import numpy as np
import bench
print(json.dumps(bench.placeholder_round_leg(np)))
print(json.dumps(bench.lookup_leg(np)))
print(json.dumps(bench.permutation_leg(np)))
print(json.dumps(bench.quotient_leg(np)))
