"""What ONE rank of an N-rank strong-scaling run sustains: one share of an N-rank plan of the 1080p frame through the whole per-frame
chain (packet-list launch with fused depth shading -> RCCL gather (one rank: to itself) -> rank-0 scatter) on one GPU, for several
numbers of frames in flight.  At N ranks a rank's launch holds 8160 / N packets -- at N = 8 one wave per SIMD -- so a frame's time is
its heaviest packet's latency and throughput = frames in flight / that latency, until the host's issue rate binds.
Usage: python tools/host_rate.py N slots [queues [frames_per_launch]]   (one configuration per process: the queue count is fixed at HIP initialisation)"""
import os, sys, time
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[3] if len(sys.argv) > 3 else str(max(8, slots))
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
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
rnd = DistributedRenderer(sc, resx, resy, 0, 1, force_collective=True, plan_ranks=n, plan_rank=min(1, n - 1), slots=slots, frames_per_launch=batch)
for _ in range(60): rnd.render(cam)
rnd.flush(); torch.cuda.synchronize()
best = None
for rep in range(3):
    K = 600
    t0 = time.perf_counter()
    for _ in range(K): rnd.render(cam)
    t1 = time.perf_counter()
    rnd.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    r = ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6)
    best = r if best is None or r[1] < best[1] else best
print("plan for %d ranks, one share (%d packets), %d launches in flight x %d frames per launch, %s hw queues: host %.1f us per frame to enqueue, %.1f us per frame end to end -> %.1f Grays/s for the %d-GPU frame if every rank keeps this pace"
      % (n, rnd.n_real, slots, rnd.batch, os.environ["GPU_MAX_HW_QUEUES"], best[0], best[1], 2088960 / best[1] / 1e3, n), flush=True)
dist.destroy_process_group()
