"""The N>1 data path on CPU: world_size 2 over gloo.  Each rank holds the packet-major planes of its own
tiles (cut out of an oracle frame -- the HIP kernel cannot run here), the per-frame collective
(render.gather_planes, the same call the GPU path makes over RCCL) brings them to rank 0, which rebuilds
the frame; it must equal the full oracle frame byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


from tests.util import frame_to_packets  # noqa: E402


def packets_to_frame(planes, xy, frame):
    resy, resx = frame.shape
    for i, (x, y) in enumerate(xy.tolist()):
        h, w = min(16, resy - y), min(16, resx - x)
        frame[y:y + h, x:x + w] = planes[i].reshape(16, 16)[:h, :w]


def _worker(rank, world, port, resx, resy, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from snail_amd import render as R
    from tests import oracle_lib as O
    from tests import util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        name = "atrium:0.02"
        tv, hb, osc = util.scene_pair(name)
        cam = util.camera_for(name, tv)
        t, u, v, tid, _ = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE, threads=2)
        plan = R.ShardPlan.make(resx, resy, world)
        xy = plan.padded_packets(rank)
        local = torch.from_numpy(np.stack([frame_to_packets(t, xy), frame_to_packets(u, xy), frame_to_packets(v, xy),
                                           frame_to_packets(tid, xy).view(np.float32)], axis=0))
        got = R.gather_planes(local, rank, world)
        # TreeStats travel as one SUM-reduce (src/node.cpp:358-359): each rank's counters over ITS tiles add up to the frame's
        mine = np.zeros(4, dtype=np.int64)
        for x, y, w, h in plan.tiles[plan.owner == rank].tolist():
            mine += osc.render_primary(cam.as_array13(), resx, resy, rect=(x, y, w, h), mode=O.MODE_IEEE, threads=1)[4].astype(np.int64)
        total = R.reduce_stats(torch.from_numpy(mine), rank, world)
        if rank == 0:
            whole = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE, threads=2)[4].astype(np.int64)
            if not np.array_equal(total.numpy(), whole):
                q.put(("stats mismatch %s vs %s" % (total.numpy(), whole), 0, [])); return
        # the rgb8 payload of the GPU path: shaded bytes, asynchronous gather (same call DistributedRenderer makes)
        bgr_full = O.shade_depth(t).reshape(resy, resx, 3)
        bgr_local = torch.from_numpy(np.stack([frame_to_packets(bgr_full[:, :, c], xy) for c in range(3)], axis=2).copy())
        glist = [torch.empty_like(bgr_local) for _ in range(world)] if rank == 0 else None
        work = dist.gather(bgr_local, glist, dst=0, async_op=True)
        work.wait()
        if rank == 0:
            fb = np.zeros((resy, resx, 3), np.uint8)
            for r in range(world):
                g = glist[r].numpy(); rxy = plan.padded_packets(r)
                for c in range(3):
                    plane = fb[:, :, c].copy(); packets_to_frame(g[:, :, c], rxy, plane); fb[:, :, c] = plane
            if fb.tobytes() != bgr_full.tobytes():
                q.put(("rgb8 frame mismatch", 0, [])); return
        if rank == 0:
            ft = np.full((resy, resx), np.nan, np.float32); fu = ft.copy(); fv = ft.copy(); fi = np.full((resy, resx), -1, np.int32)
            for r in range(world):
                g = got[r].numpy()
                rxy = plan.padded_packets(r)
                packets_to_frame(g[0], rxy, ft); packets_to_frame(g[1], rxy, fu); packets_to_frame(g[2], rxy, fv)
                packets_to_frame(g[3].view(np.int32), rxy, fi)
            ok = (ft.tobytes() == t.tobytes() and fu.tobytes() == u.tobytes() and fv.tobytes() == v.tobytes() and fi.tobytes() == tid.tobytes())
            q.put(("ok" if ok else "frame mismatch", plan.padded, [len(p) for p in plan.packets]))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("resx,resy,world", [(320, 192, 2), (250, 130, 2), (250, 130, 3)])
def test_two_rank_tile_gather_rebuilds_the_frame(resx, resy, world):
    """world_size 2 -- and 3 on a frame whose tiles do not divide evenly: the ranks' shards are padded to one size for the collective and
    rank 0's scatter skips the pad entries."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, resx, resy, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    status, padded, sizes = q.get(timeout=10)
    assert status == "ok", status
    assert sum(sizes) == ((resx + 15) // 16) * ((resy + 15) // 16) and padded == max(sizes)
