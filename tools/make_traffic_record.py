#!/usr/bin/env python3
"""profiles/rN_pmc_traffic.json from the FETCH_SIZE and WRITE_SIZE passes of tools/bench_profile.sh, stamped with the hash
of the kernel's sources (tools/kernel_source_hash.py) so that bench.py quotes it only for the kernel it was measured on.
usage: python tools/make_traffic_record.py <benchprof dir | prefix of the <prefix>_{fetch,write,stats}.json summaries that
tools/pmc_summary.py wrote on the box> > profiles/r3_pmc_traffic.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_source_hash import logdens_kernel_hash  # noqa: E402
from pmc_summary import summarize  # noqa: E402

d = sys.argv[1]
n, dim, chains = 1000000, 64, 65536
if os.path.isdir(d):
    fetch = summarize(os.path.join(d, "fetch"), "logdens_mfma_kernel")
    write = summarize(os.path.join(d, "write"), "logdens_mfma_kernel")
    stats = summarize(os.path.join(d, "stats"), "logdens_mfma_kernel")
else:
    fetch, write, stats = (json.load(open("%s_%s.json" % (d, k))) for k in ("fetch", "write", "stats"))
fkb = fetch["per_launch_counters"]["FETCH_SIZE"]
wkb = write["per_launch_counters"]["WRITE_SIZE"]
nunits = 254
alg = 8 * (n * dim + dim * chains + nunits * chains)
rec = {
    "workload": {"n": n, "d": dim, "chains_per_gpu": chains},
    "kernel_source_sha256": logdens_kernel_hash(ROOT),
    "commands": ["rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary",
                 "rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- (the same)",
                 "(tools/bench_profile.sh; per-launch sums by tools/pmc_summary.py; this file by tools/make_traffic_record.py)"],
    "launches": fetch["launches"],
    "avg_launch_ms": stats["avg_launch_ms"],
    "avg_launch_ms_under_pmc": fetch["avg_launch_ms"],
    "FETCH_SIZE_KB_per_launch_raw": fkb,
    "WRITE_SIZE_KB_per_launch": wkb,
    "gfx950_correction": "FETCH_SIZE counts the 128-B requests of a wide (16 B/lane) streaming read at 64 B (MI355X_MICROARCH.md, HBM): raw value doubled; WRITE_SIZE is exact",
    "hbm_bytes_per_launch": int(2 * fkb * 1024 + wkb * 1024),
    "algorithmic_bytes_per_launch": alg,
}
print(json.dumps(rec, indent=1))
