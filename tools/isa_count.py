#!/usr/bin/env python3
"""Static instruction mix of the main loop of device kernels, from the ISA hipcc emits (round 5: the evidence that the Winograd /
stride-2 kernels were bound by their own instruction count -- DESIGN.md section 4.1).

    hipcc --offload-arch=gfx950 <Makefile FLAGS> --save-temps -c gsa_kernels.hip   (writes *-hip-amdgcn-amd-amdhsa-gfx950.s)
    python tools/isa_count.py <file.s> [substring of the demangled kernel name ...]

For every matching kernel: the instructions between the head of its LARGEST loop and that loop's back edge, by class.  The count is
static -- both the border and the interior staging path are in it, and an epilogue that runs once per tile is counted once per item --
so it overstates a streamed-weight kernel's per-item work; it compares like with like between two kernels of the same structure."""
import collections
import re
import subprocess
import sys


def cat(op):
    for prefix, name in (("v_mfma", "mfma"), ("v_pk_", "valu_pk"), ("v_accvgpr", "accvgpr"), ("v_", "valu"), ("ds_read", "lds_read"),
                         ("ds_write", "lds_write"), ("ds_", "lds_other"), ("global_load_lds", "lds_dma"), ("global_load", "vmem_load"),
                         ("buffer_load", "vmem_load"), ("global_store", "vmem_store"), ("global_atomic", "atomic"), ("scratch_", "scratch"),
                         ("s_waitcnt", "s_waitcnt"), ("s_barrier", "s_barrier"), ("s_nop", "s_nop"), ("s_load", "smem"), ("s_buffer", "smem"), ("s_", "salu")):
        if op.startswith(prefix):
            return name
    return "other"


def main():
    src = open(sys.argv[1]).read()
    filters = sys.argv[2:]
    names = re.findall(r"^(_Z\w+):\s*; @", src, re.M)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    for name, d in zip(names, dem):
        if filters and not any(f in d for f in filters):
            continue
        m = re.search(r"^" + re.escape(name) + r":.*?^\s*s_endpgm", src, re.S | re.M)
        if not m:
            continue
        body = m.group(0).split("\n")
        labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        loops = []
        for i, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if mm:
                t = mm.group(1) or mm.group(2)
                if t in labels and labels[t] < i:
                    loops.append((labels[t], i))
        if not loops:
            continue
        a, b = max(loops, key=lambda x: x[1] - x[0])
        c = collections.Counter()
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            c[cat(l.split()[0])] += 1
        short = d.replace("void gsa::", "").replace("(gsa::ConvParams)", "").replace("(gsa::PostParams)", "")
        order = ["mfma", "valu", "valu_pk", "salu", "lds_read", "lds_write", "lds_dma", "vmem_load", "vmem_store", "s_waitcnt", "s_nop", "scratch"]
        print("%-62s %s  total %d" % (short[:62], " ".join("%s %d" % (k, c[k]) for k in order if c[k]), sum(c.values())))


if __name__ == "__main__":
    main()
