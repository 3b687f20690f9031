#!/bin/bash
# lane kernel, round-robin units against contiguous shares, over (n, len); run on the GPU box from the repository root
out=gpurun_out/lane_grid.txt
: > $out
for shape in "100000000 10000" "100000000 1000" "100000000 100000" "100000000 1000000" "10000000 1000" "1000000 1000" "200000 512"; do
  for mode in units contiguous; do
    if [ $mode = contiguous ]; then export MCMCHIP_OLBM_LANE_PLAN=1; else unset MCMCHIP_OLBM_LANE_PLAN; fi
    echo "== $shape plan=$mode" >> $out
    timeout -k 10 200 python tools/olbm_cols_probe.py $shape 9,16,32,48,64 2>&1 | grep kernel >> $out || exit 1
  done
done
