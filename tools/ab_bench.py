#!/usr/bin/env python3
"""A/B timing of two builds of the HIP library on ONE box (boxes differ by +-2 %): runs `bench.py --layers` with
_lib.HIP_LIBRARY pointing at csrc/<name> for every name given.
usage: python tools/ab_bench.py libgsa_hip.so libgsa_hip_x.so  -> gpurun_out/ab_<name>.log/.err"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    code = ("import os, sys; sys.path.insert(0, %r); sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--layers', "
            "'--no-cpu-baseline', '--no-secondary']; import torch; from gan_segmentation_amd import _lib; "
            "_lib.HIP_LIBRARY = os.path.join(os.path.dirname(_lib.HIP_LIBRARY), %r); import bench; bench.main()" % (ROOT, name))
    tag = name.replace(".so", "")
    with open(os.path.join(ROOT, "gpurun_out", "ab_%s.log" % tag), "w") as o, open(os.path.join(ROOT, "gpurun_out", "ab_%s.err" % tag), "w") as e:
        subprocess.run([sys.executable, "-c", code], stdout=o, stderr=e, cwd=ROOT, timeout=300)
