set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 300 python bench.py > gpurun_out/prof/r01_bench.json 2> gpurun_out/prof/bench.err && echo bench ok &&
GSA_SIDE_LEVELS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o fp32 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof/r01_bench_under_rocprof.json 2> gpurun_out/prof/stats.err && echo stats ok &&
timeout -k 10 300 python bench.py --gan cars --batch 4 --precision bf16 > gpurun_out/prof/r01_cars_bf16_bench.json 2>> gpurun_out/prof/bench.err && echo cars bench ok &&
GSA_SIDE_LEVELS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o cars_bf16 -- python bench.py --gan cars --batch 4 --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof/r01_cars_bf16_bench_under_rocprof.json 2>> gpurun_out/prof/stats.err && echo cars stats ok &&
bash tools/pmc.sh pmc &&
bash tools/pmc.sh pmc_cars_bf16 --gan cars --batch 4 --precision bf16 &&
# the rows after the hot path: JPEG encoder of the dataset writer (8f-1) and the decoder training step (8f-3)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o jpeg -- python tools/jpeg_bench.py --iters 50 > gpurun_out/prof/r01_jpeg_bench.txt 2>> gpurun_out/prof/stats.err && echo jpeg stats ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o train -- python tools/train_bench.py > gpurun_out/prof/r01_train_bench_under_rocprof.txt 2>> gpurun_out/prof/stats.err && echo train stats ok &&
timeout -k 10 300 python tools/train_bench.py > gpurun_out/prof/r01_train_bench.txt 2>> gpurun_out/prof/bench.err && echo train bench ok
