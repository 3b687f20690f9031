#!/bin/bash
# bench.py under rocprofv3, each pass a run of its own: kernel trace + stats, then the PMC passes (HBM fetch bytes, HBM
# write bytes, fp64 MFMA / VALU utilisation).  usage (on the GPU box): bash tools/bench_profile.sh <outdir>
set -e
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/benchprof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/stats -o b --output-format csv -- $B > $out/stats.log 2> $out/stats.err
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o b --output-format csv -- $B > $out/fetch.log 2> $out/fetch.err
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o b --output-format csv -- $B > $out/write.log 2> $out/write.err
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $out/util -o b --output-format csv -- $B > $out/util.log 2> $out/util.err
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM -d $out/inst -o b --output-format csv -- $B > $out/inst.log 2> $out/inst.err
ls -R $out | head -60
