#!/bin/bash
# The dominant kernel's average duration by rocprofv3's kernel trace and by bench.py's HIP events, over THE SAME launches of ONE process
# (on the GPU box): tools/rocprof_vs_events.sh > gpurun_out/rNN_rocprofv3_vs_events.json
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
rm -rf /tmp/rpe && ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rpe -o msm --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-groth16 --no-kzg --no-ntt --no-other-msm --no-pmc --no-two-in-flight --steps 20 --warmup 5 > /tmp/rpe.log 2>/tmp/rpe.err )
python3 - <<'P'
import csv, glob, json
line = [l for l in open('/tmp/rpe.log') if l.startswith('{')][-1]
d = json.loads(line)
stats = glob.glob('/tmp/rpe/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(stats)) if 'msm_bucket_acc' in r['Name']]
r = max(rows, key=lambda r: float(r['TotalDurationNs']))
print(json.dumps({"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-groth16 --no-kzg --no-ntt --no-other-msm --no-pmc --no-two-in-flight --steps 20 --warmup 5",
                  "kernel": r['Name'][:80], "rocprofv3": {"calls": int(r['Calls']), "average_ms": round(float(r['AverageNs']) / 1e6, 4), "min_ms": round(float(r['MinNs']) / 1e6, 4), "max_ms": round(float(r['MaxNs']) / 1e6, 4)},
                  "bench_hip_events": {"avg_launch_ms": d['roofline']['avg_launch_ms'], "launches_timed": 20, "what": "HIP events on the context's stream around the kernel, timed region only"},
                  "bench_value_under_profiler": d['value'], "ms_per_step_under_profiler": d['ms_per_step']}, indent=1))
P
