"""Helper of tests/test_gpu_parity.py::test_rccl_code_path_single_rank (run as a child process): the multi-GPU route of
DistributedRenderer -- tile plan, packet-list launch, depth shading, per-frame dist.gather over the nccl (= RCCL) backend, rank-0
scatter -- with ONE rank, so that the RCCL calls themselves run on a single-GPU box.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from snail_amd import FPSCamera, HostBVH, scenes
from snail_amd.render import DistributedRenderer
from snail_amd.scene import Scene
from tests import oracle_lib as O


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    name, resx, resy = "atrium:0.05", 640, 368
    tv = scenes.scene_by_name(name)
    h = HostBVH.build(tv)
    cam = FPSCamera(*scenes.atrium_camera()).camera()
    sc = Scene(h, 0)
    osc = O.OracleScene(tv)
    t_ref = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE)
    out = {}
    for payload, inline in (("rgb8", True), ("rgb8", False), ("hits", False)):
        rnd = DistributedRenderer(sc, resx, resy, 0, 1, payload=payload, force_collective=True, inline_collective=inline)
        assert rnd.inline == inline and rnd.nslots == (4 if inline else 3)
        for _ in range(7):
            rnd.render(cam)
        fr = rnd.flush()
        dist.barrier()
        torch.cuda.synchronize()
        if payload == "rgb8":
            want = O.shade_depth(t_ref[0]).reshape(resy, resx, 3)
            out["rgb8_equal"] = out.get("rgb8_equal", True) and bool(np.array_equal(fr.cpu().numpy(), want))
        else:
            out["hits_equal"] = bool(np.array_equal(fr.t.cpu().numpy().view(np.uint32), t_ref[0].view(np.uint32)) and np.array_equal(fr.tri_id.cpu().numpy(), t_ref[3]))
        if True:   # a moving camera: after flush() the renderer's frame is the LAST camera's, whatever the slots did (both payloads: one frame buffer per slot)
            rng = np.random.default_rng(11)
            bmin, bmax = h.bbox()
            ctr, ext = (bmin + bmax) * 0.5, (bmax - bmin)
            cams = [FPSCamera((ctr + (rng.random(3) - 0.5) * ext * 0.6).astype(np.float32), rng.random() * 6.28, (rng.random() - 0.5)).camera() for _ in range(6)]
            for cm in cams:
                rnd.render(cm)
            fr = rnd.flush()
            torch.cuda.synchronize()
            last = osc.render_primary(cams[-1].as_array13(), resx, resy, mode=O.MODE_IEEE)
            if payload == "rgb8":
                want = O.shade_depth(last[0]).reshape(resy, resx, 3)
                out["moving_equal"] = out.get("moving_equal", True) and bool(np.array_equal(fr.cpu().numpy(), want))
            else:
                out["moving_hits_equal"] = bool(np.array_equal(fr.t.cpu().numpy().view(np.uint32), last[0].view(np.uint32)) and np.array_equal(fr.tri_id.cpu().numpy(), last[3]))
        t0 = time.perf_counter()
        for _ in range(30):
            rnd.render(cam)
        rnd.flush()
        torch.cuda.synchronize()
        out[payload + ("_inline" if inline else "") + "_ms_per_frame"] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
    # multi-frame launches on the route: 7 frames of a moving camera, 3 per launch (one collective and one chunked scatter per frame), the
    # last batch partial -- after flush() the renderer's frame is the LAST camera's
    rnd = DistributedRenderer(sc, resx, resy, 0, 1, force_collective=True, frames_per_launch=3)
    assert rnd.batch == 3
    rng = np.random.default_rng(12)
    bmin, bmax = h.bbox()
    ctr, ext = (bmin + bmax) * 0.5, (bmax - bmin)
    cams = [FPSCamera((ctr + (rng.random(3) - 0.5) * ext * 0.6).astype(np.float32), rng.random() * 6.28, (rng.random() - 0.5)).camera() for _ in range(7)]
    stb = sc.new_stats()
    for cm in cams:
        rnd.render(cm, stats=stb)
    fr = rnd.flush()
    torch.cuda.synchronize()
    refs = [osc.render_primary(cm.as_array13(), resx, resy, mode=O.MODE_IEEE) for cm in cams]
    out["batched_equal"] = bool(np.array_equal(fr.cpu().numpy(), O.shade_depth(refs[-1][0]).reshape(resy, resx, 3)))
    out["batched_stats_equal"] = bool(np.array_equal(stb.cpu().numpy().astype(np.uint64), sum(r[4] for r in refs)))
    # every frame of a full batch, not just the last one
    rnd.render(cams[0]); rnd.render(cams[1]); rnd.render(cams[2])
    rnd.flush(); torch.cuda.synchronize()
    slot = (rnd.step - 1) % rnd.nslots
    out["batched_all_frames_equal"] = bool(all(np.array_equal(rnd.framesB_rgb8[slot][k].cpu().numpy(), O.shade_depth(refs[k][0]).reshape(resy, resx, 3)) for k in range(3)))
    # an UNEVEN 3-rank plan rendered share by share through the product route (real packets traced once each, pad entries skipped by
    # the scatter): the shares' frames add up to the oracle's frame, and their TreeStats -- reduce_stats() over this one-rank group
    # is the identity -- add up to the oracle's counters (src/node.cpp:358-359)
    resx2, resy2 = 250, 130
    ref2 = osc.render_primary(cam.as_array13(), resx2, resy2, mode=O.MODE_IEEE)
    want2 = O.shade_depth(ref2[0]).reshape(resy2, resx2, 3)
    total = np.zeros(4, dtype=np.int64)
    union = np.zeros((resy2, resx2, 3), dtype=np.uint8)
    sizes = []
    for share in range(3):
        rnd = DistributedRenderer(sc, resx2, resy2, 0, 1, force_collective=True, plan_ranks=3, plan_rank=share, rank0_share=0.5)
        sizes.append(len(rnd.plan.packets[share]))
        st = sc.new_stats()
        rnd.render(cam, stats=st)
        fr = rnd.flush()
        torch.cuda.synchronize()
        total += rnd.reduce_stats(st).cpu().numpy()
        union |= fr.cpu().numpy()
    out["uneven_sizes"] = sizes
    out["uneven_frame_equal"] = bool(len(set(sizes)) > 1 and np.array_equal(union, want2))
    out["uneven_stats_equal"] = bool(np.array_equal(total.astype(np.uint64), ref2[4]))
    # what bench.py --gpus N adds to its line at N > 1 (the self-audit), over the same RCCL group: a one per rank all-reduced on the device, every
    # rank's device record through all_gather_object, the collective alone (batched and single-frame payload buffers)
    one = torch.ones(1, dtype=torch.int64, device="cuda")
    dist.all_reduce(one)
    props = torch.cuda.get_device_properties(0)
    everyone = [None]
    dist.all_gather_object(everyone, {"rank": 0, "uuid": str(getattr(props, "uuid", ""))})
    gb = []
    for fpl in (1, 4):
        rnd = DistributedRenderer(sc, resx, resy, 0, 1, force_collective=True, frames_per_launch=fpl)
        for _ in range(2 * fpl):
            rnd.render(cam)
        rnd.flush()
        gb.append(rnd.measure_gather(5))
    out["audit_ok"] = bool(int(one.item()) == 1 and everyone[0]["uuid"] != "" and all(g and g["ms"] > 0 and g["bytes_per_collective"] == 0 for g in gb))
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
