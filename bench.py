#!/usr/bin/env python3
"""bench.py -- the measurement contract of this repo.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE FRAME of primary rays through the hot path (RayGenerator + SafeInv + packet BVH traversal
+ ray/triangle intersection -> hit records), inputs (SoA BVH, camera) resident in HBM before the timed
region.  Workload at N=1 = BASELINE.json configs[1]: the sponza stand-in `atrium` (263 K triangles; the
reference checkout lacks sponza.obj) at 1920x1080, primary rays.  At N>1 the frame has
N times the pixels (same camera, both axes scaled by sqrt(N) and rounded to the 16x64 tile grid), cut
into the reference's 16x64 tiles, dealt to the ranks by shuffled round-robin, traced with one launch per
rank; the hit records stay in each rank's HBM, are shaded to RGB8 with the reference's depth shading
(gVals[1], src/scene_trace.cpp:128-137 + ConvColor) and the RGB8 tiles are gathered to rank 0 with one RCCL
collective per frame -- what a render node returns in the reference (src/node.cpp:336-349) -- asynchronously,
overlapping the next frame's traversal; shading, gather and the rank-0 scatter are inside the timed region
("scaling": "weak": per-GPU work is fixed as N grows).

Mrays = rays launched (every lane of every traced packet, hit or miss), as TreeStats::TracingRays counts
them (src/scene_trace.cpp:116-117): frames are padded to whole 16x16 packets (1920x1080 -> 1920x1088).

roofline: bound "hbm"; achieved = algorithmic bytes per launch / mean kernel duration, where algorithmic
bytes = sum over the launch's rays of B_alg(ray) = 32*V_n + 64*V_t + 16 (node boxes and
triangles that ray tests in a single-ray cache-less walk, 16-B hit record) -- V_n, V_t counted once,
outside the timed region, by the device accounting kernel; kernel duration from HIP events recorded on
the launch stream around every timed launch.  NOTE: the packet algorithm fetches a node once per 256
rays (one scalar load per wavefront), so achieved may legitimately EXCEED the HBM peak: the real HBM
traffic is far below the single-ray algorithmic bytes (profiles/).

cpu_baseline: the oracle (kind "port": this repo's CPU restatement of the reference's packet algorithm,
SSE arithmetic mode = what the reference executes on x86) timed on this box's host cores, rank 0, N=1
only, on whole frames of the same workload until ~15 CPU-core-seconds are spent."""
from __future__ import annotations

import os
# four frames in flight want four hardware queues of their own: with the runtime's default of 4 queues per process, streams share
# queues with each other and with the runtime's own work (measured: 20.5 G with 3 streams / 4 queues, 22.1 G with 4 streams / >= 6
# queues, 5 or more streams slower again; profiles/README.md).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
CPU_THREADS_CAP = 16


def pmc_traffic(workload_key: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic.json: FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KB -> bytes)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        return d.get(workload_key)
    except Exception:
        return None


def frame_size_for(n_gpus: int):
    """N x the pixels of 1920x1080, same aspect, on the 16 x 64 tile grid."""
    s = math.sqrt(n_gpus)
    resx = int(round(1920 * s / 16.0)) * 16
    resy = int(round(1080 * s / 8.0)) * 8
    return resx, resy


def cpu_baseline(tv, cam, resx, resy):
    """Rank 0, N=1 only.  The ONLY place bench.py touches oracle/ -- as the reported CPU baseline."""
    from tests import oracle_lib as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, CPU_THREADS_CAP))      # the GPU box grants a 16-CPU share per GPU
    osc = O.OracleScene(tv)
    cam13 = cam.as_array13()
    osc.render_primary(cam13, resx, resy, rect=(0, 0, resx, 64), mode=O.MODE_SSE, threads=cores)   # warm-up strip
    frames, spent, times = 0, 0.0, []
    while frames < 1 or (spent * cores < 15.0 and frames < 50):
        t0 = time.perf_counter()
        osc.render_primary(cam13, resx, resy, mode=O.MODE_SSE, threads=cores)
        dt = time.perf_counter() - t0
        times.append(dt); spent += dt; frames += 1
    times.sort()
    med = times[len(times) // 2]
    rays = resx * ((resy + 15) // 16 * 16)
    return {"value": round(rays / med / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full frame(s) of the same workload (%dx%d, %d rays each), oracle SSE mode, median frame time %.3f s, %d threads"
                      % (frames, resx, resy, rays, med, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--scene", default="atrium")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal only: all ranks share GPU 0 and the per-frame gather is staged through host memory")
    ap.add_argument("--streams", type=int, default=0, help="frames in flight (HIP streams); 0 = the renderer's default (4); 1 = strictly serial frames")
    ap.add_argument("--event-every", type=int, default=8, help="bracket every n-th traversal launch with HIP events (kernel time of the roofline object)")
    ap.add_argument("--feedback-order", type=int, default=1, help="1 (default) = dispatch packets heaviest first by the node visits of an earlier frame (DistributedRenderer feedback_order)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    rehearsal = world > 1 and args.backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from snail_amd import FPSCamera, HostBVH, scenes
    from snail_amd.render import DistributedRenderer
    from snail_amd.scene import Scene

    # ---- scene (host build, once; replicated into every GPU's HBM) ----
    tv = scenes.scene_by_name(args.scene, scenes_dir=os.path.join(ROOT, "scenes"))
    t0 = time.perf_counter()
    hbvh = HostBVH.build(tv)
    build_s = time.perf_counter() - t0
    if args.scene.startswith("stress"):
        cam = FPSCamera(*scenes.stress_camera()).camera()
    elif args.scene.startswith("atrium"):
        cam = FPSCamera(*scenes.atrium_camera()).camera()
    else:   # an OBJ dropped into scenes/ (e.g. the real sponza.obj): the survey's far camera looking at the whole model
        from snail_amd import survey_camera
        cam = survey_camera(tv)
    scene = Scene(hbvh, local_rank)
    resx, resy = frame_size_for(world)
    rnd = DistributedRenderer(scene, resx, resy, rank, world, slots=args.streams if args.streams > 0 else None, stage_cpu=rehearsal,
                              feedback_order=bool(args.feedback_order))
    total_rays = rnd.rays_per_frame() if world > 1 else resx * ((resy + 15) // 16 * 16)

    # ---- algorithmic bytes (outside the timed region) ----
    acc = scene.account_primary(cam, resx, resy) if rank == 0 else None
    if rank == 0:
        b_alg = (32.0 * float(acc[1]) + 64.0 * float(acc[2])) / float(acc[0]) + 16.0
    else:
        b_alg = 0.0
    launch_rays = total_rays if world == 1 else rnd.plan.padded * 256

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rnd.render(cam)
    rnd.flush()
    barrier()

    # ---- timed region: EXACTLY K steps (frames are pipelined over args.streams HIP streams, see DistributedRenderer) ----
    # HIP events bracket every `--event-every`-th launch on the stream it is launched on (an event record is a barrier packet in the
    # stream: bracketing every launch costs ~2 % of the frame rate; the average is taken over steps / event_every launches)
    every = max(1, args.event_every)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % every == 0 else None for i in range(args.steps)]
    t0 = time.perf_counter()
    for e in ev:
        rnd.render(cam, events=e)
    rnd.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    timed = [e for e in ev if e is not None]
    kern_ms = sum(e0.elapsed_time(e1) for e0, e1 in timed) / max(1, len(timed))

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_rays * args.steps / elapsed / 1e6
        achieved = launch_rays * b_alg / (kern_ms * 1e-3) / 1e9
        tr = pmc_traffic("%s_%dx%d_n%d" % (args.scene, resx, resy, world))
        traffic = round(tr["bytes_per_launch"] / (kern_ms * 1e-3) / 1e9, 1) if tr else None
        hit_frac = float(torch.isfinite(rnd.frame.t).float().mean().item()) if rnd.frame is not None else float((rnd.frame_rgb8.amax(dim=2) > 0).float().mean().item())
        out = {
            "metric": "Mrays/sec (primary)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: gloo, ranks share one GPU -- not a measurement)",
            "config": {"workload": "%s (%d tris, sponza.obj stand-in) %dx%d primary rays, hit records (t,u,v,triId)" % (args.scene, hbvh.n_tris, resx, resy),
                       "rays_per_step": total_rays, "packets": "16x16 px = 1 wavefront", "bvh_nodes": hbvh.n_nodes, "bvh_depth": hbvh.depth,
                       "bvh_build_s": round(build_s, 3), "hit_fraction": round(hit_frac, 5), "frames_in_flight": rnd.nslots, "packet_order": "heaviest first (node visits of an earlier frame)" if rnd.feedback else "built-in region interleave",
                       "traversal_stack": "VGPR pair per wave (lane i = slot i), at most bvh_depth+1 = %d slots; LDS 0 B/wave in the main kernel (3328 B/wave only in the deferred M_EXACT pass)" % (hbvh.depth + 1),
                       "parallelism": "tiles16x64-roundrobin-x%d + depth-shade + per-frame RCCL gather of rgb8 tiles to rank 0 (overlapped with the next frames)" % world if world > 1 else "single-gpu"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_bytes_per_launch": tr["bytes_per_launch"] if tr else None,
                         "traffic_source": tr["source"] if tr else None, "kernel": "dev::k_primary", "kernel_ms": round(kern_ms, 5),
                         "alg_bytes_per_ray": round(b_alg, 1), "rays_per_launch": launch_rays,
                         "compulsory_bytes_per_ray": round((32.0 * hbvh.n_nodes + 64.0 * hbvh.n_tris) / launch_rays + 16.0, 2),
                         # what actually bounds this kernel (profiles/README.md): the SIMDs' VALU pipes.  One wave64 VALU instruction per
                         # 2 cycles per SIMD, 1024 SIMDs, 2.4 GHz; instruction count per launch from the committed PMC pass
                         "valu_pipe_frac": round(tr["valu_insts_per_launch"] * 2.0 / (1024 * ms_per_step * 1e-3 * 2.4e9), 4) if tr and world == 1 and "valu_insts_per_launch" in tr else None,
                         "note": "achieved = single-ray algorithmic bytes (32*V_n+64*V_t+16 per ray) / kernel time; the packet kernel fetches a node once per 256 rays, so this can exceed the HBM peak"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tv, cam, resx, resy)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
