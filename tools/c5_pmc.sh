#!/bin/bash
# full-size C5 under rocprofv3 --pmc FETCH_SIZE (its own pass, kernel trace only): HBM bytes per launch
# usage (on the GPU box): bash tools/c5_pmc.sh <outdir>
set -e
out=${1:-gpurun_out/c5pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$out/fetch -o c5 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/secondary_bench.py c5dev=100000256,256,10000 > $GRAFT_REPO_ROOT/$out/c5dev_fetch.log 2> $GRAFT_REPO_ROOT/$out/c5dev_fetch.err
