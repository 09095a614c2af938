#!/bin/bash
# A/B of libzkhip.so builds on one GPU box: swaps the library in place, runs the MSM bench (G1) and tools/msm_time.py
# (last line: G2), restores it.  Variants are crypto3-zk_amd/variants/libzkhip_<name>.so.
cd "$(dirname "$0")/.."
cp crypto3-zk_amd/libzkhip.so /tmp/libzkhip_base.so
for v in base "$@"; do
  if [ "$v" != base ]; then cp crypto3-zk_amd/variants/libzkhip_$v.so crypto3-zk_amd/libzkhip.so; fi
  echo "== $v"; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-groth16 --no-ntt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step']['msm_bucket_acc'])"
  timeout 300 python tools/msm_time.py 2>&1 | tail -1 | cut -c1-80
done
cp /tmp/libzkhip_base.so crypto3-zk_amd/libzkhip.so
