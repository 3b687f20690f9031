#!/bin/bash
# the C5 block (initseq centred / not centred, olbm) under rocprofv3 --kernel-trace --stats, and one PMC pass for the HBM
# fetch bytes of the initseq kernels.  usage (on the GPU box): bash tools/c5_uncentred_profile.sh <outdir>
set -e
out=$GRAFT_REPO_ROOT/${1:-gpurun_out/c5prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/stats -o c5 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/secondary_only.py c5 > $out/stats.log 2> $out/stats.err
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o c5 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/secondary_only.py c5 > $out/fetch.log 2> $out/fetch.err
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open(out + '/summary.txt', 'w') as w:
    w.write("rocprofv3 --kernel-trace --stats -- python3 tools/secondary_only.py c5 (2 centred initseq calls, 2 not centred, centre in place, olbm): kernels by total time\n")
    for r in rows[:40]:
        w.write("%-110s calls %6s total_ms %10.3f avg_us %10.1f\n" % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
    g = glob.glob(out + '/fetch/**/*counter_collection.csv', recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(g[0])):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'autocov_kernel' in r['Kernel_Name']:
            k = 'CENTRE' if ', true>(' in r['Kernel_Name'] else 'plain'        # autocov_kernel<NACC, FIRST, PAIR, CENTRE>
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value']) * 2048        # gfx950: FETCH_SIZE counts 2 KB units (MI355X_MICROARCH.md)
    w.write("\nHBM fetch bytes of autocov_kernel launches (FETCH_SIZE x 2048 B), summed over the two calls of each kind:\n")
    for k, (n, b) in agg.items():
        w.write("  %-7s %4d launches  %.1f GB (input: 2 x 204.8 GB)\n" % (k, n, b / 1e9))
PY
rm -rf $out/stats $out/fetch
cat $out/summary.txt | head -30
