"""Fold rocprofv3 counter-collection CSVs (one directory per pass) into {kernel: {counter: mean per launch}}."""
import csv, glob, json, re, sys
from collections import defaultdict

src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
regs = {}
for path in glob.glob(src + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("zkhip::", "")
        a = acc[name][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1].add(row["Dispatch_Id"])
        regs[name] = {"vgpr": int(row["VGPR_Count"]), "lds": int(row["LDS_Block_Size"]), "scratch": int(row["Scratch_Size"]),
                      "workgroup": int(row["Workgroup_Size"])}
res = {"command": "rocprofv3 --pmc <one group per pass> -- python3 " + cmd,
       "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: x2 on FETCH_SIZE for 16-B-per-lane reads, see MI355X_MICROARCH.md)",
       "kernels": {}}
for k in sorted(acc):
    if not ("msm_" in k or "ntt_" in k or "r1cs" in k or "h_" in k or "bases_" in k or "jac" in k or "fri" in k or "poly" in k or "gate_" in k or "fr_vec" in k or "perm_" in k or "lookup" in k or "ls_" in k):
        continue
    res["kernels"][k] = {c: {"mean_per_launch": v[0] / max(1, len(v[1])), "launches": len(v[1])} for c, v in sorted(acc[k].items())}
    res["kernels"][k]["resources"] = regs[k]
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res["kernels"]), "kernels")
