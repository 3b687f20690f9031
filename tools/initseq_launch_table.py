#!/usr/bin/env python3
"""Per-launch table of the autocov kernels from a rocprofv3 --kernel-trace csv (and optionally a --pmc pass).
usage: python tools/initseq_launch_table.py <kernel_trace.csv> [<counter_collection.csv>] [len]
A counter file that holds FETCH_SIZE (a pass of its own, tools/c5_pmc.sh) adds the HBM bytes per launch instead (raw KB x 1024
x 2, the gfx950 correction; launches matched by order, which is deterministic).  TFLOP/s counts the reference's 2 flop per
lag and element (initseq.c:31-37); since round 3 the kernel spends one multiply-add per lag PAIR, i.e. executes half of it."""
import csv
import re
import sys
from collections import defaultdict

trace = sys.argv[1]
pmc = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".csv") else None
n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 100000256
ctr = defaultdict(lambda: defaultdict(float))
if pmc:
    for r in csv.DictReader(open(pmc)):
        if "autocov_kernel" in r["Kernel_Name"]:
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = [r for r in csv.DictReader(open(trace)) if "autocov_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fetch = None
if pmc and any("FETCH_SIZE" in c for c in ctr.values()):
    fetch = [ctr[k]["FETCH_SIZE"] * 1024 * 2 / 1e9 for k in sorted(ctr, key=int)]
    pmc = None
print("%5s %5s %6s %9s %8s %8s %7s %7s" % ("NACC", "first", "items", "ms", "ms/item", "TFLOP/s", "TB/s", "nblk") +
      ("  %6s %6s %6s %6s" % ("GHz", "mfma%", "lds%", "wait%") if pmc else "") + ("  %8s %8s" % ("alg. GB", "HBM GB") if fetch else ""))
tot = 0.0
execd = 0
for i, r in enumerate(rows):
    m = re.search(r"autocov_kernel<(\d+), (\w+)(?:, (\w+))?>", r["Kernel_Name"])
    nacc, first = int(m.group(1)), m.group(2) == "true"
    if m.group(3):
        nacc *= 2                                   # round 3: an accumulator = 16 lag pairs = 32 lags
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    items = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    nblk = int(r["Grid_Size_Y"])
    tot += ms
    full = nblk > 100
    if full:
        execd += 16 * nacc * items
    line = "%5d %5s %6d %9.3f %8.4f" % (nacc, "y" if first else "n", items, ms, ms / items)
    if full:
        line += " %8.1f %7.2f %7d" % (2.0 * n * 16 * nacc * items / (ms * 1e-3) / 1e12, 8.0 * n * items / (ms * 1e-3) / 1e12, nblk)
    else:
        line += " %8s %7s %7d" % ("-", "-", nblk)
    if pmc:
        c = ctr.get(r["Dispatch_Id"], {})
        if c:
            ghz = c["GRBM_GUI_ACTIVE"] / (ms * 1e-3) / 1e9
            # SQ_BUSY_CYCLES is summed over SEs/XCDs; MFMA busy is per SIMD-cycle: normalise by 1024 SIMDs x kernel cycles
            cyc = c["GRBM_GUI_ACTIVE"]
            line += "  %6.2f %6.1f %6.1f %6.1f" % (ghz, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                                                  100 * c["SQ_ACTIVE_INST_LDS"] / (cyc * 256) if "SQ_ACTIVE_INST_LDS" in c else -1,
                                                  100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else -1)
    if fetch and i < len(fetch):
        line += "  %8.2f %8.2f" % (8.0 * n * items * (nblk > 100 and 1.0 or 0.0) / 1e9, fetch[i])
    print(line)
print("lag-columns executed in full-length launches: %d" % execd)
if fetch:
    print("HBM GB over all launches: %.1f" % sum(fetch[:len(rows)]))
print("total autocov ms: %.2f over %d launches; span %.2f ms" % (tot, len(rows), (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) * 1e-6))
