# One round's measurement set (run on the GPU box: gpurun -- bash tools/profile_round.sh r02): default bench line, rocprofv3
# kernel statistics of the same command (serialized: GSA_SIDE_LEVELS=0, so that the per-kernel averages are the standalone
# durations bench.py's roofline pass measures), the PMC passes, and the same for the bf16 share of configs[4].
set -o pipefail
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 400 python bench.py > gpurun_out/prof/${tag}_bench.json 2> gpurun_out/prof/bench.err && echo bench ok &&
GSA_SIDE_LEVELS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o fp32 -- python bench.py --steps 10 --warmup 2 --settle-steps 0 --repeats 1 --no-cpu-baseline --no-secondary > gpurun_out/prof/${tag}_bench_under_rocprof.json 2> gpurun_out/prof/stats.err && echo stats ok &&
timeout -k 10 300 python bench.py --gan cars --batch 4 --precision bf16 --no-secondary > gpurun_out/prof/${tag}_cars_bf16_bench.json 2>> gpurun_out/prof/bench.err && echo cars bench ok &&
GSA_SIDE_LEVELS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o cars_bf16 -- python bench.py --gan cars --batch 4 --precision bf16 --steps 10 --warmup 2 --settle-steps 0 --repeats 1 --no-cpu-baseline --no-secondary > gpurun_out/prof/${tag}_cars_bf16_bench_under_rocprof.json 2>> gpurun_out/prof/stats.err && echo cars stats ok &&
bash tools/pmc.sh pmc &&
bash tools/pmc.sh pmc_cars_bf16 --gan cars --batch 4 --precision bf16 &&
# BASELINE.json configs[2]'s workload on one card: `main.py generate`, 10 000 FFHQ samples, batch 32 (and the batch-8 form), files on disk
(timeout -k 10 300 python tools/generate_to_disk.py ffhq 10000 32 > gpurun_out/prof/${tag}_generate_10000_b32.txt 2>&1; echo "generate b32 rc=$?") &&
(timeout -k 10 300 python tools/generate_to_disk.py ffhq 10000 8 > gpurun_out/prof/${tag}_generate_10000_b8.txt 2>&1; echo "generate b8 rc=$?")
