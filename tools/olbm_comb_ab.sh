#!/bin/bash
# olbm comb kernel at full C5 size: timing experiments (MCMCHIP_OLBM_EXPERIMENT bits: 1 loads from the cache, 2 one barrier
# placement for all waves)          usage (on the GPU box): bash tools/olbm_comb_ab.sh <outfile> <exper values...>
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/comb_ab.txt}; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  echo "MCMCHIP_OLBM_EXPERIMENT=$e" >> $out
  MCMCHIP_OLBM_EXPERIMENT=$e timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tests/secondary_bench.py c5olbm=100000256,256,10000 2>&1 | grep config | cut -c1-200 >> $out || exit 1
done
