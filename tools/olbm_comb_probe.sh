#!/bin/bash
# olbm at full C5 size: comb route against the two-stream kernel, teeth sweep, kernel stats under rocprofv3
# usage (on the GPU box): bash tools/olbm_comb_probe.sh <outdir>
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/combprobe}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tests/secondary_bench.py
for t in 5 3 2 1; do
  echo "teeth $t" >> $out/sweep.txt
  MCMCHIP_OLBM_TEETH=$t timeout -k 10 200 python3 $S c5olbm=100000256,256,10000 >> $out/sweep.txt 2>&1 || exit 1
done
echo "two-stream kernel (MCMCHIP_OLBM_COMB=0)" >> $out/sweep.txt
MCMCHIP_OLBM_COMB=0 timeout -k 10 200 python3 $S c5olbm=100000256,256,10000 >> $out/sweep.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o c5 --output-format csv -- python3 $S c5olbm=100000256,256,10000 > $out/prof.log 2> $out/prof.err
