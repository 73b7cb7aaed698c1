#!/bin/bash
# One GPU-box pass over the current build: the GPU test suite, the default bench line and the bf16 share of configs[4] (eager).
#   gpurun -- bash tools/round_check.sh <tag>     -> gpurun_out/<tag>_gputests.log, <tag>_bench.json, <tag>_cars_bf16.json
tag=${1:-check}
mkdir -p gpurun_out "gpurun_out/$(dirname "$tag")"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "gpurun_out/${tag}_gputests.log" 2>&1; echo "pytest rc=$?" >> "gpurun_out/${tag}_gputests.log"
tail -4 "gpurun_out/${tag}_gputests.log"
timeout -k 10 400 python3 bench.py --steps 20 --no-cpu-baseline > "gpurun_out/${tag}_bench.json" 2> "gpurun_out/${tag}_bench.err"
timeout -k 10 200 python3 bench.py --gan cars --batch 4 --precision bf16 --no-secondary --no-cpu-baseline > "gpurun_out/${tag}_cars_bf16.json" 2>> "gpurun_out/${tag}_bench.err"
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open("gpurun_out/%s_bench.json" % tag))
print("ffhq b8 fp32: %.1f pairs/s  runs %s  oracle %s" % (d["value"], d["runs"], d["output"]["matches_oracle"]))
print("secondary:", [(s["workload"][9:30], s["value"], s["graph"][:8]) for s in d["secondary"]], " generate job %.1f pairs/s" % d["end_to_end"]["generate_job_pairs_per_s"])
c = json.load(open("gpurun_out/%s_cars_bf16.json" % tag))
print("cars b4 bf16 eager: %.1f pairs/s  runs %s" % (c["value"], c["runs"]))
PY
