"""What the HOST can issue per second on the multi-GPU route: ONE share of an N-rank plan of the 1080p frame through the whole
per-frame chain (packet-list launch with fused depth shading -> RCCL gather (one rank: to itself) -> rank-0 scatter), on one GPU.
At N ranks the device work per frame shrinks N-fold while the host work per frame does not: this is the bound of the strong-scaling
points.  Usage: python tools/host_rate.py [N ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.render import DistributedRenderer
from snail_amd.scene import Scene
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tv = scenes.scene_by_name("atrium"); h = HostBVH.build(tv)
cam = FPSCamera(*scenes.atrium_camera()).camera()
sc = Scene(h, 0)
resx, resy = 1920, 1080
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    for graph in (False,):
        rnd = DistributedRenderer(sc, resx, resy, 0, 1, force_collective=True, plan_ranks=n, plan_rank=min(1, n - 1))
        for _ in range(40): rnd.render(cam)
        rnd.flush(); torch.cuda.synchronize()
        K = 400
        t0 = time.perf_counter()
        for _ in range(K): rnd.render(cam)
        t1 = time.perf_counter()
        rnd.flush(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("plan for %d ranks, one share (%d packets): host %.1f us per frame to enqueue, %.1f us per frame end to end -> at most %.1f Grays/s for the %d-GPU frame"
              % (n, rnd.n_real, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, 2088960 / ((t2 - t0) / K) / 1e9, n), flush=True)
dist.destroy_process_group()
