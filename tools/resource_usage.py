#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy table of libsiftmi.so's device code (hipcc -Rpass-analysis=kernel-resource-usage,
no GPU needed):  python tools/resource_usage.py > profiles/r03/resource_usage.txt"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "sift_pyocl_amd", "csrc"), "resource-usage"], capture_output=True, text=True)
text = out.stdout + out.stderr
KEYS = {"VGPRs": "vgpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
        "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
rows, cur = [], None
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+):\s*(\d+)", line)
    if m and cur is not None and m.group(1).strip() in KEYS:
        cur[KEYS[m.group(1).strip()]] = int(m.group(2))
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")


def demangle(n):
    if not filt:
        return n
    d = subprocess.run([filt, n], capture_output=True, text=True).stdout.strip()
    return d.split("(")[0].replace("siftk::", "").replace("void ", "")


print("%-66s %5s %5s %8s %4s %7s %7s %7s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "sgprSpl", "vgprSpl", "LDS"))
for r in sorted(rows, key=lambda r: demangle(r["name"])):
    print("%-66s %5d %5d %8d %4d %7d %7d %7d" % (demangle(r["name"])[:66], r.get("vgpr", 0), r.get("sgpr", 0), r.get("scratch", 0),
                                                 r.get("occ", 0), r.get("sgpr_spill", 0), r.get("vgpr_spill", 0), r.get("lds", 0)))
print("\n%d kernels; with scratch: %s" % (len(rows), ", ".join(demangle(r["name"]) for r in rows if r.get("scratch", 0)) or "none"))
