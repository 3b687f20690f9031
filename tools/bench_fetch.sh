#!/bin/bash
# bench.py under rocprofv3 --pmc FETCH_SIZE only (HBM read bytes of the density kernel).  usage: bash tools/bench_fetch.sh <outdir>
set -e
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/benchfetch}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/fetch.log 2> $out/fetch.err
