#!/bin/bash
# A/B of environment switches on ONE box: `bench.py --layers` once per setting, summary through tools/blayers.py.
#   tools/ab_env.sh <tag> <layer filter> VAR=a VAR=b VAR=c,OTHER=d ...   -> gpurun_out/<tag>_<setting>.log/.err  (a setting may hold several VAR=value, comma-separated)
tag=$1; flt=$2; shift 2
mkdir -p gpurun_out "gpurun_out/$(dirname "$tag")"
for kv in "$@"; do
  env ${kv//,/ } python3 bench.py --steps 20 --warmup 3 --repeats 3 --layers --no-cpu-baseline --no-secondary > "gpurun_out/${tag}_${kv}.log" 2> "gpurun_out/${tag}_${kv}.err" || { echo "FAILED $kv"; tail -5 "gpurun_out/${tag}_${kv}.err"; exit 1; }
  echo "== $kv"
  python3 tools/blayers.py "gpurun_out/${tag}_${kv}.log" "gpurun_out/${tag}_${kv}.err" "$flt"
done
