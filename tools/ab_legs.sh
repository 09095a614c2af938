#!/bin/bash
# A/B of the placeholder legs under ZKHIP_OPTIONS settings, same box, interleaved: tools/ab_legs.sh "ntt_pair=0" "poly_coset_extend=0" ...
# (the empty setting = the defaults is always run too); prints value / round_ms per leg
for rep in 1 2; do
  for opt in "" "$@"; do
    echo "== rep $rep ZKHIP_OPTIONS='$opt'"
    ZKHIP_OPTIONS="$opt" timeout 300 python3 tools/run_placeholder_legs.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['metric'][:42], {k: v for k, v in d.items() if k in ('value', 'round_ms', 'verified')})"
  done
done
