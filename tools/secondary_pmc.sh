#!/bin/bash
# C2 (stream kernel) and C4 (temper run kernel) under rocprofv3 --pmc: vector-issue utilisation of the latency-bound kernels
# usage (on the GPU box): bash tools/secondary_pmc.sh <outdir>
set -e
out=${1:-gpurun_out/secpmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$out/pmc -o sec --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/secondary_bench.py c2 c4 > $GRAFT_REPO_ROOT/$out/sec.log 2> $GRAFT_REPO_ROOT/$out/sec.err
