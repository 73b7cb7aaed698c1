#!/bin/bash
# Collects PMC counters for the bench in separate passes (run on the GPU box via gpurun).
#   tools/pmc.sh [outdir under gpurun_out/] [extra bench.py arguments...]
set -o pipefail
out=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out
run() { name=$1; shift; GSA_SIDE_LEVELS=0 timeout -k 10 280 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$out -o $name -- python bench.py --steps 2 --warmup 1 --settle-steps 0 --repeats 1 --no-cpu-baseline --no-secondary $EXTRA > gpurun_out/$out/$name.json 2> gpurun_out/$out/$name.err; echo "$name rc=$?"; }
EXTRA="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
ls gpurun_out/$out | head -20
