#!/bin/bash
# Timing-only decomposition of conv3x3_bf16_lean (make dbg; WRONG results): cars bf16 batch 4 with parts of the kernel switched off.
#   GSA_DBG bits: 64 no output stores, 128 no staging, 256 no operand reads / MFMAs, 512 no activation loads, 1024 no weight DMA
mkdir -p gpurun_out/r5
export GSA_HIP_LIBRARY=libgsa_hip_stamp.so
for d in 0 64 128 256 512 1024 1728 1984; do
  GSA_DBG=$d timeout -k 10 200 python3 bench.py --gan cars --batch 4 --precision bf16 --steps 20 --warmup 3 --repeats 3 --layers --no-cpu-baseline --no-secondary > gpurun_out/r5/dbgbf_$d.log 2> gpurun_out/r5/dbgbf_$d.err || { echo "FAILED $d"; tail -3 gpurun_out/r5/dbgbf_$d.err; continue; }
  echo "== GSA_DBG=$d $(python3 tools/blayers.py gpurun_out/r5/dbgbf_$d.log gpurun_out/r5/dbgbf_$d.err bf16_lean | grep -E 'cvt_7|main_6.b|g.512.conv_2|g.256.conv_2|g.64.conv_2' | awk '{printf "%s %s  ", $1, $2}')"
done
