"""Shader clock and package power while the bench workload runs (rocm-smi sampled from a side thread) -- development aid for reading
roofline.valu_pipe_frac, which prices an instruction against a 2.4 GHz clock."""
import os, re, subprocess, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snail_amd import FPSCamera, HostBVH, scenes
from snail_amd.render import DistributedRenderer
from snail_amd.scene import Scene
tv = scenes.scene_by_name("atrium"); h = HostBVH.build(tv)
cam = FPSCamera(*scenes.atrium_camera()).camera()
rnd = DistributedRenderer(Scene(h, 0), 1920, 1080)
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out); pw = re.search(r"Power \(W\): ([0-9.]+)", out)
        samples.append((time.perf_counter(), int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        time.sleep(1.0)
for _ in range(200): rnd.render(cam)
rnd.flush()
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 15.0:
    for _ in range(2000): rnd.render(cam)
    rnd.flush(); n += 2000
el = time.perf_counter() - t0
stop = True; th.join()
print("frames %d in %.2f s = %.4f ms/frame = %.0f Mrays/s" % (n, el, el / n * 1e3, 2088960 * n / el / 1e6))
for t, c, p in samples: print("t=%5.1f s  sclk %5d MHz  power %6.1f W" % (t - t0, c, p))
