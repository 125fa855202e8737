#!/usr/bin/env python
"""Summarise a tools/collect_profiles.sh output directory: per-kernel calls / average duration from
rocprofv3 --stats, and FETCH_SIZE / WRITE_SIZE per launch from the two PMC passes.

HBM-traffic correction (MI355X_MICROARCH.md, section HBM): on gfx950 FETCH_SIZE reports half of the bytes
of wide coalesced streaming reads; other access widths are uncalibrated, so this script calibrates on
kernels of this very run whose byte counts are known: minmax_kernel (16-byte loads, reads exactly 4 B per
pixel) and shrink_kernel / blur writes (4-byte or 8-byte stores of a known plane)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
image_pixels = int(sys.argv[2]) if len(sys.argv) > 2 else 4096 * 4096        # of the profiled command's frames (calibration of FETCH_SIZE)


def _fingerprint():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sift_pyocl_amd import _lib
    return _lib.source_fingerprint()


def short(name):
    name = name.replace("siftk::", "").replace("void ", "")
    return name.split("(")[0]


stats = list(csv.DictReader(open(glob.glob(os.path.join(d, "kt", "*kernel_stats.csv"))[0])))
print("== rocprofv3 --kernel-trace --stats (bench.py --steps 10 --warmup 2): per-kernel totals")
print("%-46s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
tot = sum(float(r["TotalDurationNs"]) for r in stats)
fam = collections.defaultdict(lambda: [0, 0.0])
for r in stats:
    n = short(r["Name"])
    print("%-46s %7s %12.1f %12.2f %7.2f" % (n[:46], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                              100 * float(r["TotalDurationNs"]) / tot))
    key = "blur (all instances)" if n.startswith("blur_") else n
    fam[key][0] += int(r["Calls"]); fam[key][1] += float(r["TotalDurationNs"])
print("\n== families")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-46s calls %6d  total %10.1f us  avg %9.2f us  %5.1f %%" % (k, c, t / 1e3, t / c / 1e3, 100 * t / tot))


# per (kernel, grid) durations from the kernel trace: the octave-0 (largest grid) rows of the blur instances are
# what bench.py's roofline object is computed from (hipEvent) -- compare avg_us here with its avg_launch_us
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, "kt", "*kernel_trace.csv"))[0])))
# a launch is keyed by (kernel, grid width in threads, grid threads): the marching blur asks for the same NUMBER of workgroups
# on every plane it takes (1024 / 768), so at 16384^2 four octaves share a grid size -- the width (one strip of workgroups
# per 256 columns) tells them apart
per = collections.defaultdict(list)
for r in trace:
    per[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("\n== kernel trace by (kernel, grid width, grid threads): calls, avg_us, min_us")
b0_t = b0_n = 0
bigg = {}
# full resolution = the marching instances on their widest grid (the tile kernel only ever sees small octaves)
is_full = lambda n: n.startswith("blur_team") or n.startswith("blur_march")
for (n, gx, g) in per:
    if is_full(n):
        bigg[n] = max(bigg.get(n, (0, 0)), (gx, g))
for (n, gx, g), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("%-40s %7d %10d %6d %10.2f %10.2f" % (n[:40], gx, g, len(v), sum(v) / len(v), min(v)))
    if is_full(n) and (gx, g) == bigg[n]:
        b0_t += sum(v); b0_n += len(v)
if b0_n:
    print("full-resolution blur launches: %d calls, avg %.2f us" % (b0_n, b0_t / b0_n))


def pmc(sub, counter):
    f = glob.glob(os.path.join(d, sub, "*counter_collection.csv"))
    if not f:
        return {}
    # grid width of every dispatch from the kernel trace of the same pass (the counter rows carry the total only)
    width = {}
    for t in glob.glob(os.path.join(d, sub, "*kernel_trace.csv")):
        for r in csv.DictReader(open(t)):
            if "Dispatch_Id" in r:
                width[r["Dispatch_Id"]] = int(r["Grid_Size_X"])
    agg = collections.defaultdict(list)
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Grid_Size"])] += float(r["Counter_Value"])
    for (n, disp, g), v in per.items():
        agg[(n, "%d/%s" % (width.get(disp, 0), g))].append(v)      # key: "grid width/grid threads"
    return agg


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
print("\n== FETCH_SIZE / WRITE_SIZE per launch (KiB as reported, averaged over launches of the same grid)")
print("%-40s %14s %6s %14s %14s" % ("kernel", "width/grid", "n", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB"))
keys = sorted(set(fetch) | set(write), key=lambda k: -sum(fetch.get(k, [0])))
for k in keys[:40]:
    fv = fetch.get(k, []); wv = write.get(k, [])
    print("%-40s %14s %6d %14.0f %14.0f" % (k[0][:40], k[1], max(len(fv), len(wv)), sum(fv) / max(len(fv), 1), sum(wv) / max(len(wv), 1)))

# ---- traffic of the dominant kernel family (blur), corrected: reads x2 (calibrated on minmax_kernel, whose
# 16-byte loads read exactly 4 B/pixel and are reported at 1/2), writes x1 (blur / shrink stores are exact)
import json
cal = [v for (n, g), vs in fetch.items() if n == "minmax_kernel" for v in vs]
# full-resolution launches only: the marching instances on their LARGEST grid (octave 0), exactly the launches bench.py's
# roofline object times -- selected by kernel name and grid, never by byte counts (the launch that also writes the next
# octave's plane 0 stores 1.25 planes and would otherwise be the only one "near the maximum")
pmc_grid = {}
gkey = lambda g: tuple(int(x) for x in g.split("/"))
for (n, g) in set(fetch) | set(write):
    if is_full(n):
        pmc_grid[n] = max(pmc_grid.get(n, (0, 0)), gkey(g))
sel = lambda n, g: is_full(n) and gkey(g) == pmc_grid.get(n, (-1, -1))
images_f = sum(len(v) for (n, g), v in fetch.items() if n == "minmax_kernel")
images_w = sum(len(v) for (n, g), v in write.items() if n == "minmax_kernel")
nb = sum(len(v) for (n, g), v in fetch.items() if sel(n, g))
fb = sum(sum(v) for (n, g), v in fetch.items() if sel(n, g)) * 1024.0
wb = sum(sum(v) for (n, g), v in write.items() if sel(n, g)) * 1024.0
nw = sum(len(v) for (n, g), v in write.items() if sel(n, g))
if nb and nw:
    out = {"kernel_family": "blur, full-resolution (octave 0) launches", "launches_fetch_pass": nb, "launches_write_pass": nw,
           "images_fetch_pass": images_f, "images_write_pass": images_w, "launches_per_image": 6,
           "complete": bool(nb == 6 * images_f and nw == 6 * images_w),
           "library_fingerprint": _fingerprint(),
           "fetch_size_bytes_per_launch_reported": fb / nb, "read_correction": 2.0,
           "write_size_bytes_per_launch": wb / nw,
           "traffic_bytes_per_launch": 2.0 * fb / nb + wb / nw,
           "calibration": {"minmax_kernel_fetch_KiB_reported": sum(cal) / max(len(cal), 1), "expected_KiB": image_pixels * 4 // 1024},
           "rocprof_avg_launch_us_all_blur_launches": fam["blur (all instances)"][1] / fam["blur (all instances)"][0] / 1e3,
           # the launches bench.py's roofline brackets, from the kernel trace of the --stats pass (bench.py: roofline.frac_rocprof)
           "rocprof_avg_launch_us_full_resolution": (b0_t / b0_n) if b0_n else None, "rocprof_full_resolution_launches": b0_n}
    print("\n== blur family traffic per launch (corrected):", json.dumps(out))
    json.dump(out, open(os.path.join(d, "blur_traffic.json"), "w"), indent=1)
