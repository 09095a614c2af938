#!/bin/bash
# Kernel timeline of the LAST of 3 proofs (both streams), from rocprofv3's kernel trace: gpurun_out/proof_timeline.csv
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/gpurun_out"
export TMPDIR=/tmp
rm -rf /tmp/rpt && cd /tmp && rocprofv3 --kernel-trace -d /tmp/rpt -o tl --output-format csv -- python3 "$root/tools/bench_groth16.py" --steps 3 > /tmp/rpt.log 2>&1
f=$(find /tmp/rpt -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$root/gpurun_out/proof_timeline.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last proof starts at its first r1cs_eval_rows kernel
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("r1cs_eval_rows") or "r1cs_eval_rows" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    o.write("queue,kernel,start_us,end_us\n")
    for r in rows[i0:]:
        import re
        mm = re.search(r"([A-Za-z_][A-Za-z_0-9]*)\s*(<|\(|$)", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("zkhip::", ""))
        name = mm.group(1) if mm else r["Kernel_Name"][:40]
        o.write("%s,%s,%.1f,%.1f\n" % (r["Queue_Id"], name, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
print(len(rows) - i0, "kernels")
PY
