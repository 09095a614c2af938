#!/bin/bash
# per-kernel totals of ONE placeholder leg under rocprofv3 (kernel trace only): tools/leg_kernel_breakdown.sh permutation|lookup|quotient|round <out.csv>
# run on the GPU box from the repository root
leg=${1:-permutation}; out=${2:-gpurun_out/leg_${leg}_kernel_stats.csv}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_$leg
rocprofv3 --kernel-trace --stats -d /tmp/rp_$leg -o leg --output-format csv -- python3 "$root/tools/run_one_leg.py" $leg > /tmp/rp_$leg.log 2>&1
f=$(find /tmp/rp_$leg -name "*kernel_stats.csv" | head -1)
cp "$f" "$root/$out"
grep "^{" /tmp/rp_$leg.log | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('value', 'unit', 'ms_per_run', 'ms_by_phase', 'round_ms') if k in d})"
