"""Where the host's per-frame time goes on the multi-GPU route (cProfile over 2000 frames of one 8-rank share; one rank, nccl)."""
import os, sys, cProfile, pstats
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
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
rnd = DistributedRenderer(sc, 1920, 1080, 0, 1, force_collective=True, plan_ranks=8, plan_rank=1)
for _ in range(100): rnd.render(cam)
rnd.flush(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): rnd.render(cam)
pr.disable()
rnd.flush(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
dist.destroy_process_group()
