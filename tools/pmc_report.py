#!/usr/bin/env python3
"""Summarises the PMC passes of tools/pmc.sh (gpurun_out/pmc/*_counter_collection.csv) per kernel
symbol over the LAST timed bench step and writes profiles/<tag>_pmc_summary.json.

FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads -- MI355X_MICROARCH.md
section HBM); WRITE_SIZE is taken as is.  Both are KiB in the CSV."""
import collections
import csv
import json
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"


def load(name):
    out = collections.OrderedDict()
    for r in csv.DictReader(open("%s/%s_counter_collection.csv" % (d, name))):
        k = int(r["Dispatch_Id"])
        e = out.setdefault(k, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "vgpr": int(r["VGPR_Count"]),
                               "agpr": int(r["Accum_VGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                               "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [v for v in out.values() if "gsa::" in v["name"]]


sq, fe, wr = load("sq1"), load("fetch"), load("write")
n = len(sq) // 5            # bench ran warm-up 1 + 2 timed steps + 2 roofline-pass steps (all serialized here)
sq, fe, wr = sq[-n:], fe[-n:], wr[-n:]
agg = collections.OrderedDict()
for a, f, w in zip(sq, fe, wr):
    assert a["name"] == f["name"] == w["name"]
    e = agg.setdefault(a["name"], collections.Counter())
    e["launches"] += 1
    e["ns_pmc_pass"] += a["ns"]
    e["fetch_bytes"] += 2 * 1024 * f.get("FETCH_SIZE", 0)
    e["write_bytes"] += 1024 * w.get("WRITE_SIZE", 0)
    e["mfma_busy"] += a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    e["gui_active"] += a.get("GRBM_GUI_ACTIVE", 0)
    e["wave_cycles"] += a.get("SQ_WAVE_CYCLES", 0)
    e["wait_any"] += a.get("SQ_WAIT_ANY", 0)
    e["lds_conflict"] += a.get("SQ_LDS_BANK_CONFLICT", 0)
    e["active_inst"] += a.get("SQ_ACTIVE_INST_ANY", 0)
rows = []
for name, e in agg.items():
    cyc = e["gui_active"] / 8.0
    rows.append({
        "kernel": name, "launches_per_step": e["launches"],
        "hbm_bytes_per_launch": (e["fetch_bytes"] + e["write_bytes"]) / e["launches"],
        "fetch_bytes_per_launch": e["fetch_bytes"] / e["launches"], "write_bytes_per_launch": e["write_bytes"] / e["launches"],
        "avg_us_in_pmc_pass": e["ns_pmc_pass"] / e["launches"] / 1e3,
        "mfma_busy_frac": e["mfma_busy"] / (cyc * 1024) if cyc else 0.0,     # 1024 SIMDs
        "wait_frac": e["wait_any"] / e["wave_cycles"] if e["wave_cycles"] else 0.0,
        "lds_conflict_per_active_inst": e["lds_conflict"] / e["active_inst"] if e["active_inst"] else 0.0,
    })
rows.sort(key=lambda r: -r["avg_us_in_pmc_pass"] * r["launches_per_step"])
json.dump({"source": "rocprofv3 --pmc passes (tools/pmc.sh), last timed step of bench.py --steps 2", "kernels": rows},
          open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
print("%-62s %5s %9s %8s %8s %6s" % ("kernel", "n", "us/launch", "fetchMB", "writeMB", "mfma%"))
for r in rows[:16]:
    print("%-62s %5d %9.1f %8.1f %8.1f %6.1f" % (r["kernel"].replace("void gsa::", "").replace("(gsa::ConvParams)", "")[:62],
          r["launches_per_step"], r["avg_us_in_pmc_pass"], r["fetch_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6,
          100 * r["mfma_busy_frac"]))
