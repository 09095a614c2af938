#!/bin/bash
# A/B of the 512-lane NTT pass (option "ntt_wide") on one box, interleaved: tools/experiments/ab_ntt_wide.sh [rounds]
# (needs tools/experiments/r06_ntt_wide.patch applied and the library rebuilt)
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${1:-3}); do
  for v in 0 1 0; do
    echo -n "ntt_wide=$v  "
    ZKHIP_OPTIONS="ntt_wide=$v" timeout 120 python3 tools/bench_ntt.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['results']; print({k: v['ms'] for k, v in d.items()})"
  done
done
