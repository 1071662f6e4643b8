"""Stability run: many frames over 6 HIP streams mixing every device entry point on ONE scene handle (primary frames, packet lists,
fused shading, the staged config-3 pipeline with and without the mirrored bounce, generic ray / shadow batches), every result checked
against the one computed alone at the start.  Exercises the round-robin scratch slots and their events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.scene import Scene, Context, ShadowContext
from snail_amd import render as R
name = "atrium:0.05"
tv = scenes.scene_by_name(name); h = HostBVH.build(tv)
cam = FPSCamera(*scenes.atrium_camera()).camera()
sc = Scene(h, 0)
resx, resy = 640, 368
bmin, bmax = h.bbox(); c, e = (bmin + bmax) * 0.5, (bmax - bmin)
lights = np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)
plan = R.ShardPlan.make(resx, resy, 2)
xy = torch.from_numpy(plan.padded_packets(1)).cuda()
ref_frame = sc.trace_primary(cam, resx, resy)
ref_pk = sc.trace_packets(cam, resx, resy, xy)
ref_bgr = sc.trace_packets_shaded(cam, resx, resy, xy)
ref_w = sc.render_whitted(cam, resx, resy, lights)
ref_wr = sc.render_whitted(cam, resx, resy, lights, reflections=True)
torch.cuda.synchronize()
NS = 6
streams = [torch.cuda.Stream() for _ in range(NS)]
bufs = []
for k in range(NS):
    bufs.append(dict(frame=sc.alloc_frame(resx, resy), pk=tuple(torch.empty_like(x) for x in ref_pk), bgr=torch.empty_like(ref_bgr),
                     w=torch.empty_like(ref_w), wr=torch.empty_like(ref_wr)))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
t0 = time.time(); bad = 0
for it in range(iters):
    k = it % NS; st = streams[k]; b = bufs[k]
    with torch.cuda.stream(st):
        op = it % 5
        if op == 0: sc.trace_primary(cam, resx, resy, out=b["frame"], stream=st)
        elif op == 1: sc.trace_packets(cam, resx, resy, xy, out=b["pk"], stream=st)
        elif op == 2: sc.trace_packets_shaded(cam, resx, resy, xy, out=b["bgr"], stream=st)
        elif op == 3: sc.render_whitted(cam, resx, resy, lights, out=b["w"], stream=st)
        else: sc.render_whitted(cam, resx, resy, lights, out=b["wr"], stream=st, reflections=True)
    if it % 60 == 59:      # check everything written so far, then clear the buffers
        torch.cuda.synchronize()
        for b in bufs:
            ok = (torch.equal(b["frame"].t, ref_frame.t) or it < NS * 5) and True
        for kk, b in enumerate(bufs):
            for nm, got, want in (("frame.t", b["frame"].t, ref_frame.t), ("frame.id", b["frame"].tri_id, ref_frame.tri_id), ("pk.t", b["pk"][0], ref_pk[0]),
                                  ("pk.id", b["pk"][3], ref_pk[3]), ("bgr", b["bgr"], ref_bgr), ("w", b["w"], ref_w), ("wr", b["wr"], ref_wr)):
                if not torch.equal(got, want):
                    # a buffer not yet written in this window holds zeros: only count mismatches of written buffers
                    if got.abs().sum().item() != 0:
                        bad += 1; print("MISMATCH", it, kk, nm)
        for b in bufs:
            b["frame"].t.zero_(); b["frame"].tri_id.zero_()
            for x in b["pk"]: x.zero_()
            b["bgr"].zero_(); b["w"].zero_(); b["wr"].zero_()
        torch.cuda.synchronize()
print("%d launches over %d streams in %.1f s, mismatches: %d" % (iters, NS, time.time() - t0, bad))
sys.exit(1 if bad else 0)
