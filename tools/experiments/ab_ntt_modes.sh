#!/bin/bash
# (needs tools/experiments/r06_ntt_pass_pf.patch applied and the library rebuilt: the option does not exist in the shipped library)
cd "$(dirname "$0")/../.."
run() { echo -n "$1  "; ZKHIP_OPTIONS="$1" timeout 120 python3 tools/bench_ntt.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['results']; print({k: v['ms'] for k, v in d.items()})"; }
for r in 1 2; do
run "ntt_persistent=0"
run "ntt_persistent=1,ntt_pf_mode=2"
run "ntt_persistent=1,ntt_pf_mode=1"
run "ntt_persistent=0,ntt_pair=0"
run "ntt_persistent=1,ntt_pf_mode=0,ntt_pair=0"
run "ntt_persistent=1,ntt_pf_mode=1,ntt_pair=0"
run "ntt_persistent=1,ntt_pf_mode=2,ntt_pair=0"
run "ntt_persistent=2,ntt_pf_mode=0,ntt_pair=0"
done
