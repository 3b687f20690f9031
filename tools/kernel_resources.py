#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py mcmc_amd/csrc/initseq_first_a.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print("%-70s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "occ", "scratch"))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name:
        continue
    print("%-70s %5d %5d %6d %6d %4d %7d" % (name[:70], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
                                           r.get("SGPRs Spill", -1), r.get("Occupancy [waves/SIMD]", -1),
                                           r.get("ScratchSize [bytes/lane]", -1)))
