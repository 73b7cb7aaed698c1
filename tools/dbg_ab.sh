#!/bin/bash
# timing-only A/B of the diagnostic build's GSA_DBG bits (make dbg): prints pairs/s, ms per step and the serialized kernel sum per setting
for d in "$@"; do
  GSA_DBG=$d timeout -k 10 200 python tools/ab_bench.py libgsa_hip_stamp.so
  python - "$d" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_libgsa_hip_stamp.log").read().strip().splitlines()[-1])
print("GSA_DBG=%s pairs/s %.1f ms/step %.3f kernel_ms %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["whole_path"]["kernel_ms_per_step"]), flush=True)
PY
done
