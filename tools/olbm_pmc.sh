#!/bin/bash
# olbm at n = 2e7, p = 256 under rocprofv3 --pmc: where the waves of olbm_mfma_kernel spend their cycles
# usage (on the GPU box): bash tools/olbm_pmc.sh <outdir>
set -e
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/olbmpmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/p1 -o o --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/secondary_bench.py c5dev=20000256,256,10000 > $out/p1.log 2> $out/p1.err
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES -d $out/p2 -o o --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/secondary_bench.py c5dev=20000256,256,10000 > $out/p2.log 2> $out/p2.err
