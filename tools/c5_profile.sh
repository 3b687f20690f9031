#!/bin/bash
# full-size C5 (n = 1e8 x 256 columns generated on the device) under rocprofv3: per-launch kernel trace + stats
# usage (on the GPU box): bash tools/c5_profile.sh <outdir>
set -e
out=${1:-gpurun_out/c5prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
MCMCHIP_TRACE=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/trace -o c5 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/secondary_bench.py c5dev=100000256,256,10000 > $GRAFT_REPO_ROOT/$out/c5dev.log 2> $GRAFT_REPO_ROOT/$out/c5dev.err
