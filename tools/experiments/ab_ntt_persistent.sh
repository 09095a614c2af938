#!/bin/bash
# A/B of the persistent, cross-tile pipelined NTT pass (option "ntt_persistent") on one box: interleaved runs of tools/bench_ntt.py
# (needs tools/experiments/r06_ntt_pass_pf.patch applied and the library rebuilt: the option does not exist in the shipped library)
# (2^22 x 8: forward / inverse / coset forward, ms) -- tools/ab_ntt_persistent.sh [rounds]
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${1:-3}); do
  for v in 0 1 2 0; do
    echo -n "ntt_persistent=$v  "
    ZKHIP_OPTIONS="ntt_persistent=$v" timeout 120 python3 tools/bench_ntt.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['results']; print({k: v['ms'] for k, v in d.items()})"
  done
done
