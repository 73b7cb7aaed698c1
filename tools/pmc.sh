#!/bin/bash
# Collects PMC counters for the bench in separate passes (run on the GPU box via gpurun).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
run() { name=$1; shift; GSA_SIDE_LEVELS=0 timeout -k 10 280 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc -o $name -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/$name.json 2> gpurun_out/pmc/$name.err; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
ls gpurun_out/pmc | head -20
