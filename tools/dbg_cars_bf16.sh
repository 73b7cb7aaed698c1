for d in 0 64 0 64; do
  GSA_DBG=$d timeout -k 10 200 python - <<PY
import os, sys, json, io, contextlib
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--gan', 'cars', '--batch', '4', '--precision', 'bf16', '--steps', '40', '--warmup', '5', '--no-cpu-baseline', '--no-secondary']
import torch
from gan_segmentation_amd import _lib
_lib.HIP_LIBRARY = os.path.join(os.path.dirname(_lib.HIP_LIBRARY), 'libgsa_hip_stamp.so')
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("GSA_DBG=%s cars bf16 b4: %.1f pairs/s %.3f ms/step" % (os.environ.get("GSA_DBG"), d["value"], d["ms_per_step"]), flush=True)
PY
done
