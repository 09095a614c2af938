#!/bin/bash
# A/B of libzkhip.so builds for the NTT: swaps the library in place (base, the named builds under crypto3-zk_amd/ab/, base again) and prints
# tools/bench_ntt.py's line for each: tools/ab_ntt.sh nttu2 nttu4
cd "$(dirname "$0")/.."
cp crypto3-zk_amd/libzkhip.so /tmp/libzkhip_base.so
for v in base "$@" base; do
  if [ "$v" != base ]; then cp crypto3-zk_amd/ab/libzkhip_$v.so crypto3-zk_amd/libzkhip.so; else cp /tmp/libzkhip_base.so crypto3-zk_amd/libzkhip.so; fi
  echo "== $v"
  for rep in 1 2; do timeout 120 python3 tools/bench_ntt.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['results']; print({k: v['ms'] for k, v in d.items()})"; done
done
cp /tmp/libzkhip_base.so crypto3-zk_amd/libzkhip.so
