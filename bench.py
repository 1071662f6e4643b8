#!/usr/bin/env python3
"""bench.py -- the measurement contract of this repo.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|3|4|5] [--reflections] [--arith auto|host_sse|ieee] [--scaling strong|weak] [--rank0-share F]
                  [--camera-path static|orbit|dolly]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Both forms work for N > 1: started WITHOUT torch.distributed.run (no WORLD_SIZE in the environment), `bench.py --gpus N` starts its own N ranks
-- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD
process, before this process has imported torch or touched a GPU -- relays rank 0's JSON line and exits with the child's status.
The documented multi-GPU sweep: N = 1, 2, 4, 8 with the defaults (throughput; several frames per launch and per collective) and again with
`--frames-per-launch 1` (latency: one frame per launch and per collective); every N > 1 line carries `roofline`, `frames_in_flight` and the
measured latency of a lone launch through the whole route (`config.lone_launch_ms`).

A "step" is ONE FRAME through the hot path (RayGenerator + SafeInv + packet BVH traversal + ray/triangle
intersection -> hit records), inputs (BVH, camera) resident in HBM before the timed region.

Workloads (BASELINE.json `configs`; the reference checkout lacks sponza.obj, BASELINE.md section 2, so `atrium`
-- 263 K triangles -- stands in for it and `stress` -- 996 K triangles -- for abrams + lancia):
  --config 1 (default)  atrium 1920x1080, primary rays                      = the metric's configuration
  --config 3            atrium 1920x1080, primary + one point light's shadow packets (Scene::RayTrace, simple shading)
  --config 4            atrium 3840x2160, primary rays
  --config 5            stress-1M 1920x1080, primary rays
At N = 1 the frame's hit records (t, u, v, triId) are the output.  At N > 1 the SAME frame (default `--scaling strong`: the
metric is quoted at 1920x1080 on 1/2/4/8 GPUs, and the reference's server cuts one fixed frame for its n render nodes,
src/server.cpp:224-265, benchmark.txt:91-99) is cut into the reference's 16x64 tiles, dealt to the ranks by shuffled round-robin,
every rank traces the packets of its tiles with one launch and shades them to RGB8 (gVals[1] depth shading + ConvColor, fused into
the traversal kernel; config 3: the staged light pipeline), and ONE RCCL gather per frame returns the RGB8 tiles to rank 0 -- what
a render node returns in the reference (src/node.cpp:336-349) -- which scatters them into the frame; shading, gather and scatter
are inside the timed region.  `--rank0-share F` = the fraction of a fair tile share rank 0 renders itself (the reference's server
renders nothing: 0).  `--scaling weak` = the round-1 mode: N x the pixels at N GPUs (both axes x sqrt N).

Arithmetic: the reference's x86 build computes Inv / RSqrt / FastInv with rcpps / rsqrtps + one Newton step (veclib/sse/base.h:84-92), and north_star's
parity bar is against THAT path.  `--arith auto` (default) therefore runs the kernels in SNAIL_ARITH_HOST_SSE -- the host CPU's two instructions reproduced on
the device from verified tables, results equal to the reference's SSE path bit for bit -- and falls back to veclib's scalar definitions (SNAIL_ARITH_IEEE,
host-independent results, ~2 % faster) only on a host whose instructions cannot be tabulated; `config.arith` says which ran, `roofline.other_arith` carries
the other arithmetic's rate for the same K steps (N = 1), and `verified` = the timed region's output hashed equal to the frame the oracle renders on this box after the
timed region, in the timed arithmetic (`verification.live_oracle`), and to the committed digest of the oracle's frame where one exists (`verification.committed`).  The renderer's output
buffers are overwritten with all-ones after the warm-up steps (outside the timed region), so what is hashed was written by the timed launches.

Mrays = rays launched (every lane of every traced packet, hit or miss), as TreeStats::TracingRays counts them
(src/scene_trace.cpp:116-117): frames are padded to whole 16x16 packets (1920x1080 -> 1920x1088); config 3 adds the shadow
lanes with N.L > 0 (:554-557).

Before the W warm-up steps the pipeline runs untimed for `--settle-ms` (default 30 ms; the frame count is measured per workload by a probe and reported as
config.settle_frames / settle_measured_ms): from an
idle start the part's shader clock ramps for tens of milliseconds (profiles/README.md), and a short timed region would measure that ramp.  W warm-up
steps, the barrier + synchronize, and EXACTLY K timed steps follow as the contract says.

Launches: frames are pipelined over 4 HIP streams (`--streams`; on one GPU 8 for the frame with the mirrored bounce and 3 for a burst of at most 64 primary frames,
such as the round driver's 20 steps: measured, profiles/r5_streams.txt, r5_burst.txt), and `--frames-per-launch B` frames (default: enough for a launch to hold a 1080p frame's worth of packets, at least 4 on one GPU and 2 on several, i.e. 4 / 2 / 4 / 8 at
N = 1 / 2 / 4 / 8 -- 2 for the dolly camera; always 1 for config 3)
share ONE launch -- the heaviest packets of all of them first, one tail and one set of launch overheads (and, at N > 1, one collective)
for B frames; results are those of B single-frame launches.

roofline (one denominator everywhere: the step time, i.e. whole-frame throughput -- the per-launch HIP-event duration `kernel_ms` is
reported too, but with four launches overlapping it is not a stand-alone time; counters are per frame: the PMC passes profile one-frame
launches):
  bound "valu_issue": what the profiles show binds this kernel -- the SIMDs' vector issue (one wave64 VALU instruction per 2 cycles
      per SIMD, 1024 SIMDs, 2.4 GHz = 1228.8 G wave-instructions/s).  achieved = VALU wave-instructions per launch (rocprofv3
      SQ_INSTS_VALU of this workload, profiles/traffic.json) / step time.  frac <= 1 by construction.
  hbm_frac_traffic     measured HBM bytes per launch (rocprofv3 FETCH_SIZE x2 (gfx950) + WRITE_SIZE, profiles/traffic.json) / step time / 8 TB/s
  hbm_frac_packet_alg  PACKET-level algorithmic bytes -- 32 B x node visits + 64 B x triangle records fetched + 16 B x 256 hit records
                       per packet, counted on the device by the diagnostic build of the kernel -- / step time / 8 TB/s
  alg_single_ray_*     SURVEY.md section 8(d)'s single-ray, cache-less figure 32*V_n + 64*V_t + 16 per ray: informational only (a packet
                       fetches a node once per 256 rays, so this is not a lower bound on anything the kernel does and exceeds the peak)
cpu_baseline: the oracle's primary path written four lanes wide with SSE intrinsics (kind "port", variant "sse4": one SSE quad per __m128, as
the reference's veclib f32x4 code runs, rcpps / rsqrtps arithmetic; bit-identical to the scalar restatement, whose rate is reported beside
it) timed on this box's host cores, rank 0, N=1 only, on whole frames of the same workload until ~15 CPU-core-seconds are spent."""
from __future__ import annotations

import os
# four frames in flight want four hardware queues of their own: with the runtime's default of 4 queues per process, streams share
# queues with each other and with the runtime's own work (measured: 20.5 G with 3 streams / 4 queues, 22.1 G with 4 streams / >= 6
# queues, 5 or more streams slower again; profiles/README.md).  Must be set before the HIP runtime initialises.
# The frame with the mirrored bounce is nine dependent launches on its stream (primary, deferred, mirrored-ray generation, mirrored packets, deferred, their shadow
# packets, deferred, the primary hits' shadow packets, deferred): more frames in flight fill the gaps between them -- 8 streams on 16 queues 11.6 G against 10.8 G with
# 4 on 8, 7 on 8 11.3 G, 8 on 8 9.1 G (round 5, profiles/r5_streams.txt); every other workload is fastest with 4 streams, on 8 or 16 queues alike.
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if "--reflections" in sys.argv else "8")
import argparse
import json
import math
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
VALU_PEAK_GINST = 1024 * 2.4 / 2.0   # wave64 VALU instructions per ns: 1024 SIMDs x 2.4 GHz / 2 cycles each (MI355X_MICROARCH.md, cycle constants)
VALU_CYCLES_MEASURED = 2.5   # cycles per wave64 VALU instruction a SIMD sustains at 6 waves (tools/micro/visit_rate.hip, valu_banks.hip)
CPU_THREADS_CAP = 16

CONFIGS = {
    1: dict(scene="atrium", res=(1920, 1080), lights=0, what="primary rays, hit records (t,u,v,triId)"),
    3: dict(scene="atrium", res=(1920, 1080), lights=1, what="primary + 1 point light's shadow packets (Scene::RayTrace simple shading), rgb8 frame"),
    4: dict(scene="atrium", res=(3840, 2160), lights=0, what="primary rays, hit records (t,u,v,triId)"),
    5: dict(scene="stress", res=(1920, 1080), lights=0, what="primary rays, hit records (t,u,v,triId)"),
}


def kernel_source_sha16() -> str:
    """Hash of the DEVICE sources and build inputs (the kernels: snail_dev.inc, included by snail_hip.hip once per arithmetic; the GPU builder; the
    table rule the kernels share with the host; the Makefile's flags) the PMC counters of profiles/traffic.json were measured on (tools/make_traffic.py stores it beside them):
    a kernel edit without a fresh counter pass makes the bench line say `counters_stale: true` instead of pricing the new kernel with the
    old instruction count."""
    import hashlib
    h = hashlib.sha256()
    for f in ("snail_dev.inc", "lbvh.inc", "host_sse.h", "Makefile"):
        with open(os.path.join(ROOT, "snail_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def self_launch(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks as a CHILD process (never an exec, and before this
    process has imported torch or touched a GPU); the children inherit stdout, so rank 0's JSON line is this command's output."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what the host driver of this pool supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def orbit_cameras(scenes, FPSCamera, scene_name, n, step_deg=0.2, sweep_deg=20.0):
    """--camera-path orbit: the view turns by `step_deg` every frame, back and forth over +-sweep_deg around the scene's bench view, so
    that no frame is traced with a dispatch order derived from its own costs (the static view's perfect prediction)."""
    pos, ang, pitch = scenes.stress_camera() if scene_name.startswith("stress") else scenes.atrium_camera()
    period = int(round(4 * sweep_deg / step_deg))
    cams = []
    for i in range(min(n, period)):
        k = i % period
        tri = k if k <= period // 4 else (period // 2 - k if k <= 3 * period // 4 else k - period)     # 0 .. +q .. -q .. 0
        cams.append(FPSCamera(pos, ang + math.radians(tri * step_deg), pitch).camera())
    return cams


def dolly_cameras(scenes, FPSCamera, scene_name, n, step=0.01, sweep=4.0):
    """--camera-path dolly: the camera's POSITION advances by `step` scene units along its viewing direction every frame, forth and back
    over `sweep` units (the reference's viewer translates its camera every frame it is steered, src/rtracer.cpp:76-148).  Every frame then has
    an origin of its own: a new origin-relative node array (dev::k_rel_nodes + an entry of the scene's 16-entry cache), and dispatch orders
    that are predictions from an older view."""
    import numpy as np
    pos, ang, pitch = scenes.stress_camera() if scene_name.startswith("stress") else scenes.atrium_camera()
    front = np.asarray(FPSCamera(pos, ang, pitch).camera().front, dtype=np.float64)
    half = int(round(sweep / step))
    cams = []
    for i in range(min(n, 2 * half)):
        k = i if i <= half else 2 * half - i          # 0 .. half .. 1
        cams.append(FPSCamera((np.asarray(pos, dtype=np.float64) + front * (k * step)).astype(np.float32), ang, pitch).camera())
    return cams


def pmc_counters(workload_key: str):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 passes of this same command
    (profiles/traffic.json): HBM bytes = FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, and
    SQ_INSTS_VALU.  Both are properties of (kernel build, scene, camera), not of the run: they are re-measured whenever the kernel changes."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
    except Exception:
        return None
    tr = d.get(workload_key)
    if tr is not None:
        tr = dict(tr)
        # (another build of the library loaded through SNAIL_LIB_PATH -- tools/variant.sh -- is never what the counters were measured on)
        tr["_stale"] = d.get("_kernel_sha16") != kernel_source_sha16() or bool(os.environ.get("SNAIL_LIB_PATH"))
    return tr


def verify_outputs(rnd, scene_name, resx, resy, config, arith, lights):
    """What the timed path produced, against the committed digests of the oracle's frame of this workload (tests/golden/oracle_full_size.json;
    made by tests/golden/full_size.py -- `ieee` in the build container, `host_sse` per host CPU): every output buffer the renderer's launches
    wrote (one per slot and per frame of a multi-frame launch) must hold the same frame (compared on the device), and that frame's SHA-256
    must be the committed one.  Returns {"verified": True | False | None, ...}; None = no committed digest for this workload / CPU."""
    import ctypes as C
    import hashlib
    import numpy as np
    import torch
    from snail_amd import _lib
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()
    bufs = rnd.output_buffers()
    if not bufs:
        return {"verified": None, "note": "no output buffer on this rank"}
    hits = hasattr(bufs[0], "tri_id")
    planes = (lambda f: (f.t, f.u, f.v, f.tri_id)) if hits else (lambda f: (f,))
    same = all(torch.equal(a, b) for f in bufs[1:] for a, b in zip(planes(f), planes(bufs[0])))
    got = dict(zip(("sha_t", "sha_u", "sha_v", "sha_id"), (sha(x) for x in planes(bufs[0])))) if hits else {"sha_bgr" if lights else "sha_depth_bgr": sha(bufs[0])}
    res = {"verified": None, "buffers": len(bufs), "buffers_identical": bool(same), "digest": got}
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_full_size.json")))
    except Exception as e:
        res["note"] = "tests/golden/oracle_full_size.json: %s" % e
        return res
    key = "%s_%dx%d_c%s" % (scene_name, resx, resy, config)
    if arith == "ieee":
        sec, where = gold.get("ieee", {}), "ieee"
    else:
        tab = np.zeros(3 * 4096, dtype=np.uint32)
        _lib.check(_lib.lib().snail_host_sse_tables(tab.ctypes.data_as(C.c_void_p)), "snail_host_sse_tables")
        cpu = hashlib.sha256(tab.tobytes()).hexdigest()[:16]
        sec, where = gold.get("host_sse", {}).get(cpu, {}), "host_sse[%s]" % cpu
    want = sec.get(key)
    if want is None:
        res["note"] = "no committed digest for %s in section %s" % (key, where)
        return res
    res["committed"] = bool(all(want.get(k) == v for k, v in got.items()))
    res["verified"] = bool(same and res["committed"])
    res["against"] = "tests/golden/oracle_full_size.json %s.%s" % (where, key)
    return res


def live_oracle_check(verify, tv, cam, resx, resy, arith, tables_given, lights7, reflections, scalar_sse_digest=None):
    """The timed arithmetic checked LIVE (rank 0, outside the timed region; the second place bench.py touches oracle/, again as the checker): the
    oracle renders this workload's frame here and now -- ORC_MODE_IEEE; ORC_MODE_SSE = this CPU's own rcpps / rsqrtps when the library computes with
    this host's tables; ORC_MODE_TABLE over the tables in force when `--arith-tables` named another CPU -- and its SHA-256s must be those of the
    buffers the timed launches wrote (`verify["digest"]`).  So `verified` never depends on a digest having been committed for this CPU: the
    committed digests (tests/golden/oracle_full_size.json) remain as the second, independent check.  Folds the result into `verify`."""
    import hashlib
    import numpy as np
    if not verify or not verify.get("digest"):
        return verify
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    try:
        from tests import oracle_lib as O
        if arith == "ieee":
            mode, how = O.MODE_IEEE, "ORC_MODE_IEEE"
        elif tables_given:
            import ctypes as C
            from snail_amd import _lib
            tab = np.zeros(3 * 4096, dtype=np.uint32)
            _lib.check(_lib.lib().snail_host_sse_tables(tab.ctypes.data_as(C.c_void_p)), "snail_host_sse_tables")
            O.set_tables(tab)
            mode, how = O.MODE_TABLE, "ORC_MODE_TABLE over the tables in force (--arith-tables %s)" % tables_given
        else:
            mode, how = O.MODE_SSE, "ORC_MODE_SSE (this CPU's rcpps / rsqrtps)"
        got = verify["digest"]
        threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), CPU_THREADS_CAP))
        t0 = time.perf_counter()
        if "sha_bgr" in got:
            osc = O.OracleScene(tv)
            frame, _ = osc.render_whitted(cam.as_array13(), resx, resy, lights7, mode=mode, threads=threads, reflections=reflections)
            want = {"sha_bgr": sha(frame)}
        elif scalar_sse_digest is not None and mode == O.MODE_SSE and "sha_t" in got:
            want = dict(scalar_sse_digest)          # the frame cpu_baseline() rendered for its scalar timing, hashed there
            how += ", the frame of the cpu_baseline leg"
        else:
            osc = O.OracleScene(tv)
            t, u, v, tid, _ = osc.render_primary(cam.as_array13(), resx, resy, mode=mode, threads=threads)
            want = {"sha_t": sha(t), "sha_u": sha(u), "sha_v": sha(v), "sha_id": sha(tid)} if "sha_t" in got else {"sha_depth_bgr": sha(O.shade_depth(t, mode=mode).reshape(resy, resx, 3))}
        live = all(want.get(k) == v for k, v in got.items())
        verify["live_oracle"] = bool(live)
        verify["live_oracle_how"] = "%s, %d threads, %.2f s" % (how, threads, time.perf_counter() - t0)
    except Exception as e:      # (the oracle is test infrastructure: its absence must not fail a measurement -- the committed digests still stand)
        verify["live_oracle"] = None
        verify["live_oracle_how"] = "not run: %s" % e
    checks = [c for c in (verify.get("live_oracle"), verify.get("committed")) if c is not None]
    verify["verified"] = None if not checks else bool(all(checks) and verify.get("buffers_identical", False))
    if verify.get("buffers_identical") is False:
        verify["verified"] = False
    return verify


def have_digest(scene_name, resx, resy, config, arith) -> bool:
    """Is there a committed digest for this workload in this arithmetic (host_sse: for the CPU tables in force)?  The same answer on every rank of a node."""
    import ctypes as C
    import hashlib
    import numpy as np
    from snail_amd import _lib
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_full_size.json")))
    except Exception:
        return False
    key = "%s_%dx%d_c%s" % (scene_name, resx, resy, config)
    if arith == "ieee":
        return key in gold.get("ieee", {})
    tab = np.zeros(3 * 4096, dtype=np.uint32)
    if _lib.lib().snail_host_sse_tables(tab.ctypes.data_as(C.c_void_p)) != 0:
        return False
    return key in gold.get("host_sse", {}).get(hashlib.sha256(tab.tobytes()).hexdigest()[:16], {})


def weak_frame_size(n_gpus: int, res):
    """N x the pixels, same aspect, on the 16 x 64 tile grid (--scaling weak)."""
    s = math.sqrt(n_gpus)
    return int(round(res[0] * s / 16.0)) * 16, int(round(res[1] * s / 8.0)) * 8


def cpu_baseline(tv, cam, resx, resy, digest_out=None):
    """Rank 0, N=1 only.  The ONLY place bench.py touches oracle/ -- as the reported CPU baseline: the oracle's primary path written four
    lanes wide with SSE intrinsics (oracle/snail_sse4.inc, one SSE quad per __m128 as the reference's f32x4 code runs; pinned bit-exactly to
    the scalar restatement by tests/test_oracle_semantics.py), rcpps / rsqrtps arithmetic, std::thread workers over the packets."""
    from tests import oracle_lib as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, CPU_THREADS_CAP))      # the GPU box grants a 16-CPU share per GPU
    osc = O.OracleScene(tv)
    cam13 = cam.as_array13()
    osc.render_primary_sse4(cam13, resx, resy, rect=(0, 0, resx, 64), threads=cores)   # warm-up strip
    frames, spent, times = 0, 0.0, []
    while frames < 1 or (spent * cores < 15.0 and frames < 200):
        t0 = time.perf_counter()
        osc.render_primary_sse4(cam13, resx, resy, threads=cores)
        dt = time.perf_counter() - t0
        times.append(dt); spent += dt; frames += 1
    times.sort()
    med = times[len(times) // 2]
    rays = resx * ((resy + 15) // 16 * 16)
    # the scalar restatement beside it (one frame): what rounds 1-2 reported as the baseline
    t0 = time.perf_counter()
    st, su, sv, sid, _ = osc.render_primary(cam13, resx, resy, mode=O.MODE_SSE, threads=cores)
    scalar_s = time.perf_counter() - t0
    if digest_out is not None:       # the same frame serves live_oracle_check(): what the timed host_sse buffers must hash to on THIS CPU
        import hashlib
        import numpy as np
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        digest_out.update({"sha_t": sha(st), "sha_u": sha(su), "sha_v": sha(sv), "sha_id": sha(sid)})
    return {"value": round(rays / med / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "variant": "sse4",
            "scalar_port_value": round(rays / scalar_s / 1e6, 3),
            "sample": "%d full frame(s) of the same workload (%dx%d, %d primary rays each), median frame time %.4f s, %d threads; the oracle's primary path "
                      "four lanes wide in SSE intrinsics (one SSE quad per __m128, rcpps / rsqrtps + Newton as the reference's veclib), bit-identical to the "
                      "scalar restatement (which ran one frame in %.3f s here)" % (frames, resx, resy, rays, med, cores, scalar_s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs[] entry (1 = the metric's)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: strong = the same frame cut for N ranks (the metric); weak = N x the pixels")
    ap.add_argument("--rank0-share", type=float, default=1.0, help="N > 1: fraction of a fair tile share that rank 0 (which also gathers and scatters) renders; the reference's server renders nothing = 0")
    ap.add_argument("--scene", default=None, help="override the config's scene (e.g. an OBJ dropped into scenes/, or atrium:0.05)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-check", action="store_true", help="skip the live oracle check of the timed output (rank 0, after the timed region); `verified` then rests on the committed digests alone")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal only: all ranks share GPU 0 and the per-frame gather is staged through host memory")
    ap.add_argument("--streams", type=int, default=0, help="frames in flight (HIP streams); 0 = the renderer's default (4) except on one GPU: 8 -- on 16 hardware queues -- for --reflections, 3 for a burst of at most 64 primary frames; 1 = strictly serial frames")
    ap.add_argument("--event-every", type=int, default=8, help="bracket every n-th traversal launch with HIP events (roofline.kernel_ms)")
    ap.add_argument("--feedback-order", type=int, default=1, help="1 (default) = dispatch packets heaviest first by the node visits of an earlier frame (DistributedRenderer feedback_order)")
    ap.add_argument("--frames-per-launch", type=int, default=0, help="trace this many frames (1..8) with one launch -- and, at N > 1, move them with one collective (DistributedRenderer frames_per_launch); 0 = enough for a launch to hold 8160 packets (one 1080p frame), at most 8, at least 4 on one GPU and 2 on several: 4 / 2 / 4 / 8 at N = 1 / 2 / 4 / 8; config 3 always 1")
    ap.add_argument("--lone-frames", type=int, default=12, help="frames (N > 1: launches through the whole route) traced one at a time after the timed region (lone_frame_ms / lone_launch_ms); 0 = skip")
    ap.add_argument("--camera-path", default="static", choices=["static", "orbit", "dolly"], help="dolly = the camera's position advances 0.01 units along its viewing direction every step (every frame needs a new origin-relative node array); orbit = the camera turns 0.2 degrees every step (dispatch orders are then predictions from an older view, re-derived every --order-refresh frames of a slot, inside the timed region)")
    ap.add_argument("--settle-ms", type=float, default=30.0, help="untimed frames for this many milliseconds BEFORE the W warm-up steps: the part's clocks ramp for tens of ms after an idle start (2.09 -> 1.91 -> 2.2 GHz over the first 600 frames, profiles/README.md), which a 20-step timed region would otherwise measure instead of the kernel; 0 = off; reported as config.settle_ms")
    ap.add_argument("--dry-run", action="store_true", help="start the ranks, rendezvous, one all-reduce over the chosen backend, print {dry_run, ranks} and exit: checks the launch path without a GPU (with --backend gloo)")
    ap.add_argument("--reflections", action="store_true", help="config 3 only: + the one mirrored bounce of gVals[7] (Scene::TraceReflection: mirrored packets with per-ray origins through the same RayTrace)")
    ap.add_argument("--arith", default="auto", choices=["auto", "ieee", "host_sse"], help="arithmetic of the path's approximate operations (include/snail_hip.h): host_sse = veclib's SSE definitions as this host's CPU executes them (rcpps / rsqrtps reproduced on the device + Newton), i.e. the reference's x86 results bit for bit -- the arithmetic north_star's parity bar is about; ieee = veclib's scalar definitions (host-independent results, ~2 %% faster); auto (default) = host_sse, or ieee on a host whose instructions cannot be tabulated (config.arith says which ran)")
    ap.add_argument("--arith-tables", default=None, help="with --arith host_sse: compute with the committed rcpps / rsqrtps tables of a NAMED CPU (tests/golden/rcp_tables.npz: xeon_skylake_sp, epyc_9575f; or a .npy file of uint32[3, 4096]) instead of this host's own -- the same bits on any host (snail_arith_set_tables)")
    ap.add_argument("--order-refresh", type=int, default=16, help="frames of a slot between two derivations of its dispatch order while the camera moves (DistributedRenderer order_refresh)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as the plain command: become the launcher.  Nothing GPU-related has been imported or initialised in this process.
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    rehearsal = world > 1 and args.backend == "gloo"
    if args.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
        tt = torch.ones(1, dtype=torch.float64, device="cpu" if (rehearsal or world == 1) else torch.device("cuda", local_rank))
        if world > 1:
            dist.all_reduce(tt)
        if rank == 0:
            print(json.dumps({"dry_run": True, "ranks": int(tt.item()), "n_gpus": world, "backend": args.backend if world > 1 else None}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d over RCCL needs %d GPUs, this node shows %d (--backend gloo = a rehearsal on one GPU)" % (world, world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
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

    cfg = CONFIGS[args.config]
    scene_name = args.scene or cfg["scene"]
    # ---- scene (host build, once; replicated into every GPU's HBM) ----
    tv = scenes.scene_by_name(scene_name, scenes_dir=os.path.join(ROOT, "scenes"))
    t0 = time.perf_counter()
    hbvh = HostBVH.build(tv)
    build_s = time.perf_counter() - t0
    if scene_name.startswith("stress"):
        cam = FPSCamera(*scenes.stress_camera()).camera()
    elif scene_name.startswith("atrium"):
        cam = FPSCamera(*scenes.atrium_camera()).camera()
    else:   # an OBJ dropped into scenes/ (e.g. the real sponza.obj): the survey's far camera looking at the whole model
        from snail_amd import survey_camera
        cam = survey_camera(tv)
    scene = Scene(hbvh, local_rank)
    arith_note = None
    if args.arith_tables:
        if args.arith == "ieee":
            raise SystemExit("--arith-tables needs --arith host_sse")
        args.arith = "host_sse"
        from snail_amd.scene import set_arith_tables
        set_arith_tables(np.load(args.arith_tables) if args.arith_tables.endswith(".npy") else np.load(os.path.join(ROOT, "tests", "golden", "rcp_tables.npz"))[args.arith_tables])
    if args.arith == "auto":      # the reference's own arithmetic where this host's CPU can be reproduced (every x86 CPU seen so far), veclib's scalar one otherwise
        from snail_amd._lib import SnailError
        try:
            scene.set_arith("host_sse")
            args.arith = "host_sse"
        except SnailError as e:
            args.arith, arith_note = "ieee", "host_sse is not available here: %s" % e
    scene.set_arith(args.arith)
    resx, resy = cfg["res"] if (world == 1 or args.scaling == "strong") else weak_frame_size(world, cfg["res"])
    lights7 = None
    if cfg["lights"]:
        bmin, bmax = hbvh.bbox()
        c, e = (bmin + bmax) * 0.5, (bmax - bmin)
        lights7 = np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)   # one point light above the nave
    # frames per launch, unless given: enough for a launch to hold at least one 1080p frame's worth of packets (8160), at most 8; at least 4 on one GPU (round 5,
    # profiles/r5_shapes.txt: 4 streams x 4 frames against 4 x 2 -- the driver's 20-step run +3..6 %, the long run +1.4 %, 4K +0.5 %, stress-1M -0.7 %), 2 on several
    # and for the dolly camera (every frame walks its own origin-relative node array: 16 of them in flight instead of 8 cost 3.5 %, profiles/r5_dolly.txt)
    per_rank = ((resx + 15) // 16) * ((resy + 15) // 16) / float(world)
    auto_fpl = int(min(8, max(4 if (world == 1 and args.camera_path != "dolly") else 2, math.ceil(8160.0 / max(1.0, per_rank)))))
    # launches in flight, unless given: the renderer's default (4; 3 with an asynchronous gather) -- except on one GPU: 8 for the frame with the mirrored bounce (nine
    # dependent launches per frame: profiles/r5_streams.txt), and 3 for a BURST of primary frames (a timed region of at most 64 frames: 20 frames are five four-frame
    # launches, and three in flight finish them 4.5 % sooner and far more evenly than four -- eight alternating runs on one box, 25.0-25.9 against 23.6-25.3 Grays/s,
    # profiles/r5_burst.txt; in a long run four are ahead by 0-2 %)
    auto_slots = None
    if world == 1:
        if args.reflections and cfg["lights"]: auto_slots = 8
        elif not cfg["lights"] and args.steps <= 64: auto_slots = 3
    rnd = DistributedRenderer(scene, resx, resy, rank, world, slots=args.streams if args.streams > 0 else auto_slots, stage_cpu=rehearsal,
                              feedback_order=bool(args.feedback_order), lights7=lights7, reflections=bool(args.reflections and cfg["lights"]), rank0_share=args.rank0_share,
                              frames_per_launch=args.frames_per_launch if args.frames_per_launch > 0 else auto_fpl, order_refresh=args.order_refresh)
    primary_rays = rnd.rays_per_frame() if world > 1 else resx * ((resy + 15) // 16 * 16)
    # the cameras of the timed steps: one fixed view, or a view that turns every step (built here, outside the timed region)
    moving = args.camera_path != "static" and not args.scene
    path = ((orbit_cameras if args.camera_path == "orbit" else dolly_cameras)(scenes, FPSCamera, scene_name, args.steps + args.warmup + 4096) if moving else [cam])
    cam_at = lambda i: path[i % len(path)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- rays per step (config 3 counts the shadow lanes too) and the diagnostics, all outside the timed region ----
    st = scene.new_stats()
    rnd.render(cam, stats=st)
    rnd.flush()
    barrier()
    tot = rnd.reduce_stats(st) if world > 1 else st
    total_rays = int(tot.cpu().numpy()[2]) if (rank == 0 and cfg["lights"]) else primary_rays
    node_visits = int(tot.cpu().numpy()[1]) if rank == 0 else 0
    # the two diagnostic passes (N = 1), outside the timed region AND before the settle / warm-up frames (round 5: they used to sit between the W warm-up steps
    # and the timed region -- several milliseconds of an almost idle GPU right before a 1.7 ms timed region, which then measured the part's clock ramp: on this
    # round's boxes the driver's 20-step command read 23.2-23.5 Grays/s that way and 26.0-26.3 when timed straight after warm frames)
    acc = pk = None
    if rank == 0 and world == 1:
        acc = scene.account_primary(cam, resx, resy)                      # single-ray accounting walk, informational
        pk = scene.packet_costs(cam, resx, resy)                          # per-packet {visits, ..., triangle records fetched, ...}
    barrier()
    # settle: untimed frames of the run's own camera path for `--settle-ms`.  The count comes from a measured probe of THIS workload (two
    # rounds of the pipeline, rank 0's clock) and is the same on every rank -- the ranks' collectives must pair up -- by a MAX all-reduce
    settle_frames, settle_measured_ms = 0, 0.0
    if args.settle_ms > 0:
        probe = 2 * rnd.nslots * rnd.batch
        for i in range(probe):          # every slot's first frame (scratch allocations, first dispatch orders), untimed
            rnd.render(cam_at(i))
        rnd.flush()
        barrier()
        t1 = time.perf_counter()
        for i in range(2 * probe):
            rnd.render(cam_at(i))
        rnd.flush()
        barrier()
        per_frame_ms = (time.perf_counter() - t1) * 1e3 / (2 * probe)
        want = min(4000, int(args.settle_ms / max(per_frame_ms, 1e-3)))
        if world > 1:
            tt = torch.tensor([want], dtype=torch.int64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            want = int(tt.item())
        settle_frames = max(rnd.batch, want // rnd.batch * rnd.batch)
        t1 = time.perf_counter()
        for i in range(settle_frames):
            rnd.render(cam_at(3 * probe + i))
        rnd.flush()
        torch.cuda.synchronize()
        settle_measured_ms = (time.perf_counter() - t1) * 1e3
    warm0 = settle_frames + (6 * rnd.nslots * rnd.batch if args.settle_ms > 0 else 0)     # the path goes on where the settle frames stopped
    # (what the timed loop needs is made BEFORE the warm-up steps, so that nothing but the barrier + synchronize stands between them and the timed region)
    every = -(-max(1, args.event_every) // rnd.batch) * rnd.batch
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % every == 0 else None for i in range(args.steps)]
    cams_timed = [cam_at(warm0 + args.warmup + i) for i in range(args.steps)]
    for i in range(args.warmup):
        rnd.render(cam_at(warm0 + i))
    rnd.flush()
    if rank == 0 and not rehearsal:
        rnd.poison_outputs()      # what verify_outputs() hashes after the timed region was written INSIDE it: the settle / warm-up frames' results are gone
    barrier()

    # ---- timed region: EXACTLY K steps (frames are pipelined over the renderer's HIP streams, see DistributedRenderer) ----
    # HIP events bracket every `--event-every`-th launch on the stream it is launched on (an event record is a barrier packet in the
    # stream: bracketing every launch costs ~2 % of the frame rate; the average is taken over steps / event_every launches)
    # (a launch of B frames keeps ONE pair, that of its first frame: the stride is a multiple of B)
    t0 = time.perf_counter()
    for e, c in zip(ev, cams_timed):
        rnd.render(c, events=e)
    rnd.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # ---- what the timed region produced (rank 0), before anything else overwrites the renderer's buffers ----
    verify = None
    if len(path) > 1:    # a moving camera: every slot holds another view -- send one more round of the FIXED view through the same renderer (same
        for _ in range(rnd.nslots * rnd.batch):      # launch form, the dispatch orders it has arrived at) and check that, after the clock has stopped
            rnd.render(cam)
        rnd.flush()
        barrier()
    vkey = "%d%s" % (args.config, "r" if (args.reflections and cfg["lights"]) else "")
    if rank == 0:
        verify = verify_outputs(rnd, scene_name, resx, resy, vkey, args.arith, cfg["lights"])
        verify["buffers_poisoned_before_timed_region"] = not rehearsal      # (DistributedRenderer.poison_outputs: nothing a settle / warm-up frame wrote survives into this check)
        verify["what"] = ("one more round of the fixed view through the timed renderer after the timed region (the timed frames each hold another view)" if len(path) > 1
                          else "the last frame each slot traced inside the timed region")
    timed = [e for e in ev if e is not None]
    durs = []
    for e0, e1 in timed:
        try:
            durs.append(e0.elapsed_time(e1))
        except Exception:       # a pair the renderer did not record (never the case with the stride above; not worth a failed run)
            pass
    kern_ms = sum(durs) / max(1, len(durs))

    # ---- one launch at a time through the whole route (N > 1): trace -> gather -> scatter of `frames_per_launch` frames with nothing
    # else in flight, host clock of rank 0 between a barrier and the completed frame -- the latency side of the throughput figure ----
    lone_launch_ms = None
    if world > 1 and args.lone_frames > 0:
        ts = []
        for _ in range(args.lone_frames):
            barrier()
            t1 = time.perf_counter()
            for _k in range(rnd.batch):
                rnd.render(cam)
            rnd.flush()
            ts.append((time.perf_counter() - t1) * 1e3)
        ts.sort()
        lone_launch_ms = ts[len(ts) // 2]
        barrier()

    # ---- N > 1: the line audits its own ranks -- what the process group itself reports (a one per rank, all-reduced over the group the
    # collectives use), the GPU each rank sits on (distinct UUIDs unless this is the one-GPU rehearsal), whether every rank's renderer
    # switches streams the fast way, and the collective alone (rank 0's inbound bytes / time) ----
    audit = None
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device="cpu" if rehearsal else torch.device("cuda", local_rank))
        dist.all_reduce(one)
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "device_index": local_rank, "uuid": str(getattr(props, "uuid", "")), "name": props.name, "raw_stream_switch": bool(rnd._raw_stream_switch)}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        gb = rnd.measure_gather(20)
        if rank == 0:
            uu = [e["uuid"] for e in everyone]
            audit = {"rccl_ranks_seen": int(one.item()), "ranks": everyone, "devices_distinct": len(set(uu)) == world and all(uu),
                     "raw_stream_switch_all": all(e["raw_stream_switch"] for e in everyone), "gather": gb}

    # ---- one frame at a time (N = 1): the latency a frame has when nothing else is in flight ----
    lone_ms = None
    if rank == 0 and world == 1 and args.lone_frames > 0:
        ts = []
        for _ in range(args.lone_frames):
            torch.cuda.synchronize()
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            rnd.render(cam, events=e)
            rnd.flush()
            ts.append(e[0].elapsed_time(e[1]))
        ts.sort()
        lone_ms = ts[len(ts) // 2]

    # ---- the same K steps with ONE frame per launch (N = 1, when the timed region used several): the rate without the batching ----
    fpl1 = None
    if rank == 0 and world == 1 and rnd.batch > 1 and args.lone_frames > 0:
        r1 = DistributedRenderer(scene, resx, resy, 0, 1, slots=rnd.nslots, feedback_order=bool(args.feedback_order),
                                 frames_per_launch=1, order_refresh=args.order_refresh)
        for i in range(max(8, settle_frames + args.warmup)):      # the same settle + warm-up as the timed region had
            r1.render(cam_at(i))
        r1.flush()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for c in cams_timed:
            r1.render(c)
        r1.flush()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        fpl1 = {"value": round(primary_rays * args.steps / dt1 / 1e6, 2), "ms_per_step": round(dt1 * 1e3 / args.steps, 5), "frames_per_launch": 1,
                "frames_in_flight": r1.nslots, "note": "the same %d steps after the timed region with one frame per launch (not the headline)" % args.steps}

    # ---- the same K steps in the OTHER arithmetic (N = 1): both modes' rates in one line ----
    other = None
    if rank == 0 and world == 1 and args.lone_frames > 0:
        other_name = "host_sse" if args.arith == "ieee" else "ieee"
        try:
            scene.set_arith(other_name)
            r2 = DistributedRenderer(scene, resx, resy, 0, 1, slots=rnd.nslots, feedback_order=bool(args.feedback_order), lights7=lights7,
                                     reflections=bool(args.reflections and cfg["lights"]), frames_per_launch=rnd.batch, order_refresh=args.order_refresh)
            for i in range(max(8 * rnd.batch, (settle_frames + args.warmup) // rnd.batch * rnd.batch)):      # the same settle + warm-up as the timed region had
                r2.render(cam_at(i))
            r2.flush()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for c in cams_timed:
                r2.render(c)
            r2.flush()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            other = {"arith": other_name, "value": round(total_rays * args.steps / dt2 / 1e6, 2), "ms_per_step": round(dt2 * 1e3 / args.steps, 5),
                     "note": "the same %d steps after the timed region in the other arithmetic (not the headline)" % args.steps}
        except Exception as e:      # a host whose rcpps / rsqrtps cannot be tabulated: reported, not fatal
            other = {"arith": other_name, "value": None, "note": str(e)}
        scene.set_arith(args.arith)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        step_s = ms_per_step * 1e-3
        value = total_rays * args.steps / elapsed / 1e6
        key = "%s_%dx%d_n%d_c%d%s%s" % (scene_name, resx, resy, world, args.config, "r" if (args.reflections and cfg["lights"]) else "", "_sse" if args.arith == "host_sse" else "")
        tr = pmc_counters(key)
        shared_note = ""
        if tr is None and world > 1 and args.scaling == "strong":
            # N ranks share the SAME frame: its counters are those of the one-GPU pass of this workload (the walk of a packet does not depend
            # on which rank traces it; the depth-shading epilogue of the tile route adds < 1 % vector instructions), priced against N GPUs' peak
            tr = pmc_counters("%s_%dx%d_n1_c%d%s%s" % (scene_name, resx, resy, args.config, "r" if (args.reflections and cfg["lights"]) else "", "_sse" if args.arith == "host_sse" else ""))
            shared_note = " [N = %d: the frame's counters from the one-GPU pass, peak = %d x one GPU's]" % (world, world)
        valu = tr.get("valu_insts_per_launch") if tr else None
        traffic = tr.get("bytes_per_launch") if tr else None
        peak_ginst = VALU_PEAK_GINST * world
        hbm_peak = HBM_PEAK_GBS * world
        roof = {"bound": "valu_issue", "achieved": round(valu / step_s / 1e9, 1) if valu else None, "peak": round(peak_ginst, 1), "unit": "Gwaveinst/s",
                "frac": round(valu / step_s / 1e9 / peak_ginst, 4) if valu else None,
                "counters_stale": bool(tr.get("_stale")) if tr else None,
                # what a SIMD of this part sustains on the visit's own instruction mix at 6 waves (tools/micro/visit_rate.hip: 118-121 cycles per
                # 47-instruction visit; pure v_fma_f32 / v_mul_f32 streams 2.45-2.65 cycles per instruction, tools/micro/valu_banks.hip)
                "frac_of_measured_issue_rate": round(valu / step_s / 1e9 / (peak_ginst * 2.0 / VALU_CYCLES_MEASURED), 4) if valu else None,
                "measured_issue_rate_note": "peak above = one wave64 VALU instruction per 2 cycles per SIMD (MI355X_MICROARCH.md); measured on this part: one per %.1f cycles (profiles/r3_valu_microbench.txt)" % VALU_CYCLES_MEASURED,
                "traffic": traffic * rnd.batch if traffic else None,
                "traffic_unit": "HBM bytes per launch of %d frame(s) (rocprofv3 PMC of a one-frame launch: 2 x FETCH_SIZE + WRITE_SIZE, x frames per launch)" % rnd.batch,
                "frames_per_launch": rnd.batch, "traffic_bytes_per_frame": traffic,
                "hbm_peak_GBs": hbm_peak, "hbm_frac_traffic": round(traffic / step_s / 1e9 / hbm_peak, 4) if traffic else None,
                "valu_insts_per_launch": valu * rnd.batch if valu else None, "valu_insts_per_frame": valu,
                "lanes_live_per_valu_inst": tr.get("lanes_live_per_valu_inst") if tr else None, "kernels": tr.get("kernels") if tr else None, "counters_source": (tr.get("source") + shared_note) if tr else "no PMC pass committed for workload key %s" % key,
                "kernel": tr.get("kernel", "dev::k_primary") if tr else "dev::k_primary", "kernel_ms": round(kern_ms, 5), "kernel_ms_note": "HIP events around one launch (%d frame(s)) on its own stream while %d launches are in flight: overlapped, not a stand-alone duration; every fraction here uses ms_per_step (per frame) and per-frame counters" % (rnd.batch, rnd.nslots),
                "denominator_ms": round(ms_per_step, 5), "lone_frame_ms": round(lone_ms, 5) if lone_ms is not None else None,
                "one_frame_per_launch": fpl1, "other_arith": other}
        if pk is not None:
            visits, fetched = int(pk[:, 0].sum()), int(pk[:, 4].sum())
            pbytes = 32 * visits + 64 * fetched + 16 * 256 * len(pk)
            roof.update({"packet_alg_bytes_per_launch": pbytes * rnd.batch, "packet_alg_bytes_per_frame": pbytes, "packet_node_visits": visits, "packet_tri_records_fetched": fetched,
                         "hbm_frac_packet_alg": round(pbytes / step_s / 1e9 / hbm_peak, 4),
                         "compulsory_bytes_per_frame": 32 * hbvh.n_nodes + 64 * hbvh.n_tris + 16 * primary_rays})
        if acc is not None:
            b_alg = (32.0 * float(acc[1]) + 64.0 * float(acc[2])) / float(acc[0]) + 16.0
            roof.update({"alg_single_ray_bytes_per_ray": round(b_alg, 1),
                         "alg_single_ray_frac_of_hbm_peak": round(primary_rays * b_alg / step_s / 1e9 / hbm_peak, 4),
                         "alg_single_ray_note": "SURVEY 8(d)'s single-ray cache-less bytes; the packet kernel fetches a node once per 256 rays, so this is informational and may exceed 1"})
        if rnd.frame is not None:
            hit_frac = float(torch.isfinite(rnd.frame.t).float().mean().item())
        else:
            hit_frac = float((rnd.frame_rgb8.amax(dim=2) > 0).float().mean().item())
        if world == 1:
            par = "single-gpu"
        else:
            par = "tiles16x64-roundrobin-x%d (rank-0 share %.2f) + %s + per-frame RCCL gather of rgb8 tiles to rank 0 (overlapped with the next frames)" % (
                world, args.rank0_share, "light pipeline" if cfg["lights"] else "depth-shade")
        out = {
            "metric": "Mrays/sec (primary)" if not cfg["lights"] else ("Mrays/sec (primary + mirrored + shadow)" if args.reflections else "Mrays/sec (primary + shadow)"), "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: gloo, ranks share one GPU -- not a measurement)",
            "config": {"workload": "BASELINE config %d: %s (%d tris%s) %dx%d %s" % (args.config, scene_name, hbvh.n_tris,
                                                                                     ", sponza.obj stand-in" if scene_name.startswith("atrium") else "", resx, resy, cfg["what"]),
                       "baseline_config": args.config, "reflections": bool(args.reflections and cfg["lights"]), "arith": args.arith, "arith_note": arith_note, "arith_tables": args.arith_tables or ("host CPU" if args.arith == "host_sse" else None), "rays_per_step": total_rays, "primary_rays_per_step": primary_rays, "node_visits_per_step": node_visits,
                       "packets": "16x16 px = 1 wavefront", "bvh_nodes": hbvh.n_nodes, "bvh_depth": hbvh.depth,
                       "bvh_build_s": round(build_s, 3), "hit_fraction": round(hit_frac, 5), "frames_in_flight": rnd.nslots * rnd.batch, "frames_per_launch": rnd.batch, "launches_in_flight": rnd.nslots,
                       "lone_launch_ms": round(lone_launch_ms, 5) if lone_launch_ms is not None else None,
                       "lone_launch_note": ("host clock of rank 0 around %d frame(s) = one launch + one collective + scatter with nothing else in flight" % rnd.batch) if lone_launch_ms is not None else None,
                       "settle_ms": args.settle_ms, "settle_frames": settle_frames, "settle_measured_ms": round(settle_measured_ms, 2),
                       "camera_path": args.camera_path if len(path) > 1 else "static", "order_refresh": rnd.order_refresh if rnd.feedback else None,
                       "ranks": world, "rccl_ranks_seen": audit["rccl_ranks_seen"] if audit else None, "devices_distinct": audit["devices_distinct"] if audit else None,
                       "raw_stream_switch": audit["raw_stream_switch_all"] if audit else None, "gather_alone": audit["gather"] if audit else None,
                       "rank_devices": [{k: e[k] for k in ("rank", "device_index", "uuid")} for e in audit["ranks"]] if audit else None,
                       "backend": ("gloo (rehearsal)" if rehearsal else "nccl (RCCL)") if world > 1 else None,
                       "packet_order": "heaviest first (node visits of an earlier frame)" if rnd.feedback else "built-in region interleave",
                       "traversal_stack": "VGPR pair per wave (lane i = slot i), at most bvh_depth = %d slots; LDS 0 B/wave in the main kernel (3328 B/wave only in the deferred M_EXACT pass)" % hbvh.depth,
                       "packets_per_rank": [len(p) for p in rnd.plan.packets] if world > 1 else None,
                       "parallelism": par},
            "roofline": roof,
            "verified": verify.get("verified") if verify else None, "verification": verify,
        }
        scalar_digest = {}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tv, cam, resx, resy, digest_out=scalar_digest)
        else:
            out["cpu_baseline"] = None
        if verify is not None and not args.no_live_check:
            verify = live_oracle_check(verify, tv, cam, resx, resy, args.arith, args.arith_tables, lights7, bool(args.reflections and cfg["lights"]), scalar_digest or None)
            out["verified"], out["verification"] = verify.get("verified"), verify
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and verify and verify.get("verified") is False:
        raise SystemExit("bench.py: the timed path's output does NOT match the committed digest: %s" % json.dumps(verify))


if __name__ == "__main__":
    main()
