#!/bin/bash
# Diagnostic PMC pass: dynamic instruction counts per kernel (scalar / vector / branch ...) over one bench step; run on the GPU box via gpurun.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcinst
GSA_SIDE_LEVELS=0 timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmcinst -o insts -- python bench.py --steps 1 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-secondary > gpurun_out/pmcinst/insts.json 2> gpurun_out/pmcinst/insts.err; echo "insts rc=$?"
python3 - <<'PY'
import collections, csv
rows = collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/pmcinst/insts_counter_collection.csv")):
    if "gsa::" not in r["Kernel_Name"]:
        continue
    k = int(r["Dispatch_Id"])
    e = rows.setdefault(k, {"name": r["Kernel_Name"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = list(rows.values()); rows = rows[-(len(rows) // 3):]
agg = collections.OrderedDict()
for e in rows:
    a = agg.setdefault(e["name"], collections.Counter())
    for k, v in e.items():
        if k != "name": a[k] += v
    a["launches"] += 1
print("%-66s %9s %8s %8s %7s %7s %7s %7s" % ("kernel (per WAVE, summed over the launches of a step)", "waves", "salu", "valu", "smem", "lds", "branch", "vmem"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:18]:
    w = a["SQ_WAVES"] or 1
    print("%-66s %9d %8.0f %8.0f %7.0f %7.0f %7.0f %7.0f" % (n.replace("void gsa::", "").replace("(gsa::ConvParams)", "")[:66], w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_VALU"] / w,
          a["SQ_INSTS_SMEM"] / w, a["SQ_INSTS_LDS"] / w, a["SQ_INSTS_BRANCH"] / w, (a["SQ_INSTS_VMEM_RD"] + a["SQ_INSTS_VMEM_WR"]) / w))
PY
