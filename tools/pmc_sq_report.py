#!/usr/bin/env python3
"""Per-kernel sums of the tools/pmc_sq.sh passes (last bench step)."""
import collections, csv, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmcsq"
agg = collections.OrderedDict()
for name in ("vmem", "lds", "act"):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open("%s/%s_counter_collection.csv" % (d, name))):
        if "gsa::" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = rows.setdefault(k, {"name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = list(rows.values())
    rows = rows[-(len(rows) // 3):]        # warm-up, timed step, roofline-pass step -> keep the last
    for e in rows:
        a = agg.setdefault(e["name"], collections.Counter())
        for k, v in e.items():
            if k != "name":
                a[name + ":" + k if k == "ns" else k] += v
        a["launches:" + name] += 1
def short(n):
    return n.replace("void gsa::", "").replace("(gsa::ConvParams)", "")[:44]
print("%-44s %7s | %7s %7s %7s %7s %7s | %7s %7s %7s %7s | %6s %6s" % (
    "kernel", "us", "rd/w", "wr/w", "vmlat", "addrFF%", "cmdFF%", "lds/w", "ldslat", "confl%", "ldsFF%", "valu%", "mfma%"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["vmem:ns"])[:16]:
    waves_cyc = a["SQ_WAVE_CYCLES"]; busy = a["SQ_BUSY_CYCLES"] or 1
    nw = waves_cyc / busy if busy else 0
    rd, wr = a["SQ_INSTS_VMEM_RD"], a["SQ_INSTS_VMEM_WR"]
    print("%-44s %7.1f | %7.0f %7.0f %7.0f %7.1f %7.1f | %7.0f %7.0f %7.1f %7.1f | %6.1f %6.1f" % (
        short(n), a["vmem:ns"] / a["launches:vmem"] / 1e3, rd, wr,
        a["SQ_INST_LEVEL_VMEM"] / max(rd + wr, 1), 100 * a["SQ_VMEM_TA_ADDR_FIFO_FULL"] / busy, 100 * a["SQ_VMEM_TA_CMD_FIFO_FULL"] / busy,
        a["SQ_INSTS_LDS"], a["SQ_INST_LEVEL_LDS"] / max(a["SQ_INSTS_LDS"], 1), 100 * a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1),
        100 * (a["SQ_LDS_DATA_FIFO_FULL"] + a["SQ_LDS_CMD_FIFO_FULL"]) / busy,
        100 * a["SQ_ACTIVE_INST_VALU"] / max(a["SQ_BUSY_CYCLES"], 1), 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_BUSY_CYCLES"], 1)))
