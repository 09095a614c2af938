#!/bin/bash
# A/B of libzkhip.so builds for the G2 multiexp on one GPU box: swaps the library in place, prints tools/msm_time.py's G2 line.
cd "$(dirname "$0")/.."
cp crypto3-zk_amd/libzkhip.so /tmp/libzkhip_base.so
for v in base "$@" base; do
  if [ "$v" != base ]; then cp crypto3-zk_amd/variants/libzkhip_$v.so crypto3-zk_amd/libzkhip.so; else cp /tmp/libzkhip_base.so crypto3-zk_amd/libzkhip.so; fi
  echo "== $v"; timeout 300 python tools/msm_time.py 2>&1 | tail -1 | cut -c1-400
done
cp /tmp/libzkhip_base.so crypto3-zk_amd/libzkhip.so
