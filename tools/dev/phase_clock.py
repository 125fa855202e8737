"""dev: where the waves of the per-keypoint kernels spend their time (a build of the library with -DSIFT_PHASE_CLOCK, made
on first use as /tmp/libsiftmi_ph.so; k_keypoint.hpp: PhaseClock).  python tools/dev/phase_clock.py [size] [white|smooth] [octaves] [name=value ...]"""
import os, sys, shutil, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = os.path.join(ROOT, "sift_pyocl_amd")
ph = "/tmp/libsiftmi_ph.so"
if not os.path.exists(ph):      # the instrumented build (30 s): never kept inside the package
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-DSIFT_PHASE_CLOCK", os.path.join(pkg, "csrc", "siftmi.hip"), "-o", ph],
                          stderr=subprocess.DEVNULL)
shutil.copy(os.path.join(pkg, "libsiftmi.so"), "/tmp/libsiftmi_keep.so")
shutil.copy(ph, os.path.join(pkg, "libsiftmi.so"))
try:
    import numpy as np, torch
    import sift_pyocl_amd as sp
    from sift_pyocl_amd import _lib
    from util import smooth_noise
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    kind = sys.argv[2] if len(sys.argv) > 2 else "white"
    octaves = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    img = smooth_noise((size, size)) if kind == "smooth" else np.random.default_rng(0).random((size, size), dtype=np.float32)
    t = torch.from_numpy(img).cuda()
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=octaves or None)
    for kv in sys.argv[4:]:
        n, v = kv.split("="); plan.set_option(n, int(v))
    L = _lib.lib()
    L.siftmi_dev_phase.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    for _ in range(3): k = plan.keypoints(t)
    L.siftmi_dev_phase(None, 1)
    N = 10
    for _ in range(N): k = plan.keypoints(t)
    out = (C.c_uint64 * 32)()
    L.siftmi_dev_phase(out, 0)
    v = [int(x) for x in out]
    print("%s %d^2, %d octaves, %d keypoints; per image, shader-clock cycles summed over the waves of both groups" % (kind, size, plan.octave_max, len(k)))
    names = {0: ["prologue (fold table, barrier)", "keypoint set-up, first loads requested", "wait for the batch's neighbours (HBM / L2)",
                 "evaluation + vote atomics", "owners: masks -> segments", "voters: rank -> pool store", "owners: ordered sums, reset",
                 "smoothing, peaks, park", "final flush", "", "", ""],
             16: ["prologue (fold table, pool init, barrier)", "window set-up, thresholds", "row intervals", "row look-up + next batch's loads requested",
                  "wait for the batch's neighbours (HBM / L2)", "evaluation", "routing a-b: atomics, masks, prefix, entries", "routing c: entries -> ranks -> pool stores",
                  "ordered sums", "normalise, quantise, record", "hand-out ticket", ""]}
    for base, title in ((0, "orientation_kernel"), (16, "descriptor_kernel (wave form)")):
        a = v[base:base + 16]
        tot = sum(a[:12]) or 1
        kp, nb = a[13] / N, a[14] / N
        print("== %s: %.0f keypoints, %.0f batches per image; wave time summed %.1f Mcycles per image, longest wave %.0f cycles (%.1f us at 2.1 GHz)"
              % (title, kp, nb, a[12] / N / 1e6, a[15], a[15] / 2100.0))
        for i in range(12):
            if a[i]: print("   %-62s %5.1f %%   %8.0f cycles per keypoint  %7.0f per batch" % (names[base][i], 100.0 * a[i] / tot, a[i] / N / max(kp, 1), a[i] / N / max(nb, 1)))
finally:
    shutil.copy("/tmp/libsiftmi_keep.so", os.path.join(pkg, "libsiftmi.so"))
