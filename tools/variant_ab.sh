#!/bin/bash
# A/B of libzkhip.so builds on one GPU box: swaps the library in place (base, the named variants under crypto3-zk_amd/variants/, base again) and
# prints single-MSM kernel times, a Groth16 proof and the 50-column KZG commit for each.
cd "$(dirname "$0")/.."
cp crypto3-zk_amd/libzkhip.so /tmp/libzkhip_base.so
for v in base "$@" base; do
  if [ "$v" != base ]; then cp crypto3-zk_amd/variants/libzkhip_$v.so crypto3-zk_amd/libzkhip.so; else cp /tmp/libzkhip_base.so crypto3-zk_amd/libzkhip.so; fi
  echo "== $v"
  timeout 300 python tools/msm_time.py 2>&1 | grep "^group" | cut -c1-330
  timeout 300 python tools/bench_groth16.py --steps 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('groth16', d.get('ms_per_proof'), {k:v for k,v in (d.get('kernel_ms_serial_proof') or d.get('kernel_ms_last_proof') or {}).items() if 'red' in k or 'acc' in k})"
  timeout 300 python tools/bench_kzg.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kzg', d['ms'], d['kernel_ms'].get('msm_bucket_red'), 'proof_eval', d['proof_eval']['ms'])"
done
cp /tmp/libzkhip_base.so crypto3-zk_amd/libzkhip.so
