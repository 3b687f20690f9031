#!/bin/bash
# full-size C5 under rocprofv3 --pmc (its own pass, kernel trace only): matrix-pipe and LDS activity per autocov launch
# usage (on the GPU box): bash tools/c5_initseq_pmc.sh <outdir>
set -e
out=${1:-gpurun_out/c5ipmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$out/pmc -o c5 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/secondary_bench.py c5dev=100000256,256,10000 > $GRAFT_REPO_ROOT/$out/c5dev_pmc.log 2> $GRAFT_REPO_ROOT/$out/c5dev_pmc.err
