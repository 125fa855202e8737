"""dev: clocks and package power (rocm-smi, every 0.2 s) while the headline call loops for 4 s:  python tools/dev/clk_probe.py
(round 5: sclk 2394-2395 MHz, mclk 2000 MHz, 1136-1139 W -- the loop runs at the top clock, below the power limit)"""
import subprocess, threading, time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sift_pyocl_amd as sp
t = torch.from_numpy(np.random.default_rng(0).random((4096, 4096), dtype=np.float32)).cuda()
plan = sp.SiftPlan(shape=(4096, 4096), dtype=np.float32, octave_max=3)
for _ in range(10): plan.keypoints(t)
stop = False
samples = []
def poll():
    while not stop:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout
        samples.append(o.strip()[:600])
        time.sleep(0.2)
th = threading.Thread(target=poll); th.start()
t0 = time.time()
while time.time() - t0 < 4: plan.keypoints(t)
stop = True; th.join()
for s in samples[-4:]: print(s)
