#!/bin/bash
# pairs/s of the fused generate step over the batch size (FFHQ fp32 unless told otherwise): tools/batch_sweep.sh [gan] [precision] -> gpurun_out/sweep_<gan>_<precision>.txt
gan=${1:-ffhq}; prec=${2:-fp32}
mkdir -p gpurun_out
out=gpurun_out/sweep_${gan}_${prec}.txt
: > $out
for b in 1 2 4 8 16 32; do
  timeout -k 10 200 python3 bench.py --gan $gan --precision $prec --batch $b --steps 30 --warmup 5 --repeats 3 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%s %s batch %2d: %8.1f pairs/s  %.3f ms/step  runs %s  graph %s' % ('$gan', '$prec', $b, d['value'], d['ms_per_step'], d['runs'], d['config']['graph']))" | tee -a $out
done
