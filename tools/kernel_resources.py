"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c gsa_kernels.hip -o /tmp/k.o 2> res.txt; python tools/kernel_resources.py res.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names, rows = [], []
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip().split(" ")[0]      # the mangled name (the remark flag follows it)

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    names.append(name)
    rows.append((g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs"), g(r"LDS Size \[bytes/block\]")))
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for d, r in zip(dem, rows):
    if flt in d:
        print("%-110s vgpr %3d agpr %3d scratch %4d occ %d sgpr %3d" % (d[:110], *r[:5]))
