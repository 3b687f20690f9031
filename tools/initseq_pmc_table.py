#!/usr/bin/env python3
"""Per-launch table of the autocov kernels from one rocprofv3 --kernel-trace --pmc pass (tools/c5_initseq_pmc.sh).
usage: python tools/initseq_pmc_table.py <counter_collection.csv> [nblk of a full pass] [GB per column]"""
import csv
import re
import sys
from collections import defaultdict

f = sys.argv[1]
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 763
gb = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
ctr, meta = defaultdict(dict), {}
for r in csv.DictReader(open(f)):
    if "autocov_kernel" in r["Kernel_Name"]:
        d = ctr[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = r
print("%-22s %5s %8s %8s %6s %7s %6s %6s %6s" % ("kernel<NACC,FIRST,PAIR>", "items", "ms", "ms/item", "GHz", "mfma %", "TB/s", "wait %", "lds %"))
tot = 0.0
for d in sorted(ctr, key=int):
    r, c = meta[d], ctr[d]
    name = re.search(r"autocov_kernel<([^>]*)>", r["Kernel_Name"]).group(1)
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    tot += ms
    wg = int(r["Grid_Size"]) // 256
    full = wg >= nblk
    items = wg // nblk if full else wg
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                         # summed over the 8 XCDs
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)      # SIMD-cycles with the matrix pipe busy / SIMD-cycles of the launch
    lds = c.get("SQ_ACTIVE_INST_LDS", 0) / (4 * 256 * cyc)
    print("%-22s %5d %8.3f %8.4f %6.2f %7.1f %6s %6.1f %6.1f" % (name, items, ms, ms / max(items, 1) if full else 0, cyc / ms / 1e6,
          100 * busy, "%.2f" % (items * gb / ms) if full else "-", 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * lds))
print("total %.2f ms" % tot)
