#!/usr/bin/env python3
"""Summarises gpurun_out/pmc/*_counter_collection.csv per (kernel, grid) for the LAST bench step."""
import collections, csv, re, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
def load(name):
    out = collections.OrderedDict()
    for r in csv.DictReader(open("%s/%s_counter_collection.csv" % (d, name))):
        k = int(r["Dispatch_Id"])
        e = out.setdefault(k, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]),
                               "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "lds": r["LDS_Block_Size"],
                               "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out
sq, fe, wr = load("sq1"), load("fetch"), load("write")
def short(n):
    n = re.sub(r"gsa::|void |\(gsa::ConvParams\)|\(.*\)$", "", n)
    return n[:44]
# keep the last third of dispatches (the last timed step), match by order
ids = [k for k in sq if "gsa::" in sq[k]["name"]]
ids = ids[-len(ids) // 3:]
fids = [k for k in fe if "gsa::" in fe[k]["name"]][-len(ids):]
wids = [k for k in wr if "gsa::" in wr[k]["name"]][-len(ids):]
print("%-44s %9s %8s %6s %6s %6s %6s %7s %8s %8s" % ("kernel", "grid", "us", "clkGHz", "mfma%", "wait%", "ldscf%", "v/a/lds", "fetchMB", "writeMB"))
for k, fk, wk in zip(ids, fids, wids):
    e = sq[k]
    us = e["t"] / 1e3
    if us < 40: continue
    clk = e.get("GRBM_GUI_ACTIVE", 0) / 8 / (e["t"] * 1e-9) / 1e9
    wc = e.get("SQ_WAVE_CYCLES", 1)
    busy = e.get("SQ_BUSY_CYCLES", 1)
    mf = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    # MFMA busy is counted per SIMD-cycle; GRBM_GUI_ACTIVE/8 = cycles per XCD; 1024 SIMDs on the chip
    cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
    mfma_pct = 100.0 * mf / (cyc * 1024) if cyc else 0
    print("%-44s %9d %8.1f %6.2f %6.1f %6.1f %6.2f %7s %8.1f %8.1f" % (
        short(e["name"]), e["grid"], us, clk, mfma_pct, 100.0 * e.get("SQ_WAIT_ANY", 0) / wc,
        100.0 * e.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, e.get("SQ_ACTIVE_INST_ANY", 1)),
        "%s/%s/%sK" % (e["vgpr"], e["agpr"], int(e["lds"]) // 1024),
        2 * fe[fk].get("FETCH_SIZE", 0) / 1024, wr[wk].get("WRITE_SIZE", 0) / 1024))
