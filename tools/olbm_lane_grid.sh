#!/bin/bash
# olbm_lane.hip against the kernels that served before (MCMCHIP_OLBM_LANE=0) over (n, len, p); run on the GPU box from the
# repository root; writes gpurun_out/lane_grid.txt
out=gpurun_out/lane_grid.txt
: > $out
for shape in "100000000 10000" "100000000 1000" "100000000 100000" "100000000 1000000" "10000000 1000" "1000000 1000" "200000 512"; do
  for mode in lane before; do
    if [ $mode = before ]; then export MCMCHIP_OLBM_LANE=0; else unset MCMCHIP_OLBM_LANE; fi
    echo "== n len = $shape, $mode" >> $out
    timeout -k 10 200 python tools/olbm_cols_probe.py $shape 4,5,6,8,9,16,24,32,48,64 2>&1 | grep "kernel " >> $out || exit 1
  done
done
