"""What the multi-GPU route costs on top of the traversal, measured with ONE rank on one GPU (nccl backend, force_collective):
tile plan + packet-list launch + depth shading + dist.gather (to itself) + scatter, against the plain single-GPU frame."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from snail_amd import FPSCamera, HostBVH, scenes
from snail_amd.render import DistributedRenderer
from snail_amd.scene import Scene
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tv = scenes.scene_by_name("atrium"); h = HostBVH.build(tv)
cam = FPSCamera(*scenes.atrium_camera()).camera()
sc = Scene(h, 0)
SLOTS = int(os.environ["SLOTS"]) if "SLOTS" in os.environ else None    # default: 4 frames in flight alone, 3 beside the collective
for label, kw in (("single-GPU route (hit records in frame layout)", dict(slots=SLOTS)), ("multi-GPU route, rgb8 payload", dict(force_collective=True, slots=SLOTS)),
                  ("multi-GPU route, rgb8 payload, gather on the slot stream", dict(force_collective=True, slots=SLOTS, inline_collective=True)),
                  ("multi-GPU route, hits payload", dict(force_collective=True, payload="hits", slots=SLOTS))):
    rnd = DistributedRenderer(sc, 1920, 1080, 0, 1, **kw)
    for rep in range(3):
        for _ in range(30): rnd.render(cam)
        rnd.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): rnd.render(cam)
        host = (time.perf_counter() - t0) / 200 * 1e3      # the host's share: enqueueing only (it blocks when a slot's previous gather is not done)
        rnd.flush(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
    print("%-50s %.4f ms/frame = %.0f Mrays/s   (host enqueue loop %.4f ms/frame)" % (label, ms, 2088960 / ms / 1e3, host))
dist.destroy_process_group()
