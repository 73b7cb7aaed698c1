"""Summarise a `bench.py --layers` run: headline + per-layer ms for layers matching a filter.
usage: python tools/blayers.py gpurun_out/x.log gpurun_out/x.err [filter]"""
import json
import sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print("pairs/s %.1f  ms/step %.3f  kernel-ms %.3f  oracle %s" % (d["value"], d["ms_per_step"], d["whole_path"]["kernel_ms_per_step"], d["output"]["matches_oracle"]))
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = []
for l in open(sys.argv[2]):
    if "ms/step" not in l or flt not in l:
        continue
    name, rest = l.split("|", 1) if "|" in l else (l[:70], l[70:])
    parts = rest.split()
    rows.append((parts[0], float(parts[1]), name.replace("void gsa::", "").replace("(gsa::ConvParams)", "").strip()))
tot = 0.0
for lay, ms, name in sorted(rows):
    tot += ms
    print("%-18s %7.3f  %s" % (lay, ms, name[:60]))
print("sum %.3f" % tot)
