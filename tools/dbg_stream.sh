#!/bin/bash
# Timing-only decomposition of conv3x3_wino_stream (make dbg; WRONG results): FFHQ batch 8 with parts of the kernel switched off after its
# first two items.  GSA_DBG bits: 64 no output stores, 128 no staging, 256 no patch reads / transform / MFMAs, 512 no activation loads, 1024 no weight DMA
mkdir -p gpurun_out/r5
export GSA_HIP_LIBRARY=libgsa_hip_stamp.so
for d in 0 64 128 256 512 1024 1536 1728 1984; do
  GSA_DBG=$d timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --repeats 3 --layers --no-cpu-baseline --no-secondary > gpurun_out/r5/dbgst_$d.log 2> gpurun_out/r5/dbgst_$d.err || { echo "FAILED $d"; tail -2 gpurun_out/r5/dbgst_$d.err | cut -c1-200; continue; }
  echo "== GSA_DBG=$d $(python3 tools/blayers.py gpurun_out/r5/dbgst_$d.log gpurun_out/r5/dbgst_$d.err wino_stream | grep -E 'g.(32|64|128|256).conv_2|cvt_6' | awk '{printf "%s %s  ", $1, $2}')"
done
