"""Per-launch PMC counter sums of one kernel from a rocprofv3 --pmc --kernel-trace csv pass.
usage: python tools/pmc_summary.py <dir of the pass> <kernel name substring>   -> JSON on stdout"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def summarize(d, sub):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    sums, launches = defaultdict(float), set()
    for f in cc:
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
                launches.add(r["Dispatch_Id"])
    durs = []
    for f in kt:
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    n = max(len(launches), 1)
    return {"kernel_substring": sub, "launches": len(launches),
            "avg_launch_ms": sum(durs) / len(durs) if durs else None,
            "per_launch_counters": {k: v / n for k, v in sorted(sums.items())}}


if __name__ == "__main__":
    print(json.dumps(summarize(sys.argv[1], sys.argv[2]), indent=1))
