#!/bin/bash
# olbm comb kernel at full C5 size under rocprofv3 --pmc (each pass a run of its own): matrix-pipe utilisation and clock,
# instruction mix, HBM bytes; and the cached-loads timing experiment
# usage (on the GPU box): bash tools/olbm_comb_pmc.sh <outdir>
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/combpmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tests/secondary_bench.py
if [ -z "$SKIP_EXPER" ]; then
echo "loads from the cache (MCMCHIP_OLBM_EXPERIMENT=1; results wrong by construction)" > $out/exper.txt
MCMCHIP_OLBM_EXPERIMENT=1 timeout -k 10 200 python3 $S c5olbm=100000256,256,10000 >> $out/exper.txt 2>&1 || exit 1
fi
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $out/util -o c5 --output-format csv -- python3 $S c5olbm=100000256,256,10000 > $out/util.log 2> $out/util.err || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY -d $out/inst -o c5 --output-format csv -- python3 $S c5olbm=100000256,256,10000 > $out/inst.log 2> $out/inst.err || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o c5 --output-format csv -- python3 $S c5olbm=100000256,256,10000 > $out/fetch.log 2> $out/fetch.err || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o c5 --output-format csv -- python3 $S c5olbm=100000256,256,10000 > $out/write.log 2> $out/write.err
for d in util inst fetch write; do python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/$d olbm_comb_kernel > $out/$d.json; rm -rf $out/$d; done
