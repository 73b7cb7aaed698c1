#!/bin/bash
# Diagnostic PMC passes (SQ memory-path counters) over one bench step; run on the GPU box via gpurun.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcsq
run() { name=$1; shift; GSA_SIDE_LEVELS=0 timeout -k 10 280 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmcsq -o $name -- python bench.py --steps 1 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-secondary > gpurun_out/pmcsq/$name.json 2> gpurun_out/pmcsq/$name.err; echo "$name rc=$?"; }
run vmem SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL &&
run lds SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY
