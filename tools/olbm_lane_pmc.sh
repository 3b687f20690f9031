#!/bin/bash
# olbm lane kernel (n = 1e8, len = 1e4, first p columns of one resident series) under rocprofv3 --pmc, each pass a run of
# its own: matrix-pipe utilisation and clock, waits, LDS conflicts, HBM bytes.
# usage (on the GPU box): bash tools/olbm_lane_pmc.sh <p> <outdir>
p=${1:-64}
out=$GRAFT_REPO_ROOT/${2:-gpurun_out/lanepmc$p}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/olbm_cols_probe.py
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $out/util -o c5 --output-format csv -- python3 $S 100000000 10000 $p > $out/util.log 2> $out/util.err || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY -d $out/inst -o c5 --output-format csv -- python3 $S 100000000 10000 $p > $out/inst.log 2> $out/inst.err || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o c5 --output-format csv -- python3 $S 100000000 10000 $p > $out/fetch.log 2> $out/fetch.err || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o c5 --output-format csv -- python3 $S 100000000 10000 $p > $out/write.log 2> $out/write.err
for d in util inst fetch write; do python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/$d olbm_lane_kernel > $out/$d.json; rm -rf $out/$d; done
