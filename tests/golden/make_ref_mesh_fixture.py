"""Fixtures that put the REFERENCE'S OWN MESHES in front of the HIP kernels on the GPU box (which has no /root/reference).

Runs in the build container only:  python tests/golden/make_ref_mesh_fixture.py [lancia feline barracks]

Per mesh it reads /root/reference/scenes/<name>.obj through the product's ingest (snail_amd.scenes.load_obj: the reference loader's face / Repair /
FlipNormals order rules, sscanf("%f") number parsing) and stores
  tests/golden/<name>_tris.npz   the post-ingest float32 triangle soup [n, 3, 3] -- DATA: the vertex coordinates of the reference's asset in
                                 the order its loader yields them (no source text), what BVH::Construct is handed;
  tests/golden/ref_meshes.json   per mesh: triangle / node count, depth, FNV of the oracle's tree, and for the survey's far camera at
                                 1920 x 1080 the SHA-256 of the oracle's t / u / v / triId planes, hits, sum(triId), sum(t),
                                 TreeStats and the config-3 frame (one light, with and without the mirrored bounce) -- in ORC_MODE_IEEE (CPU
                                 independent) and in ORC_MODE_SSE on THIS CPU (keyed by the sha256[:16] of the tables the product library
                                 took from it, as tests/golden/oracle_full_size.json does), plus the numbers the survey recorded from the
                                 reference itself for that mesh (tests/golden/survey_digests.json), which the ORC_MODE_SSE digests made
                                 on the survey's CPU reproduce.
The GPU tests (tests/test_gpu_ref_meshes.py) compare the HIP frames with these digests, with the oracle live on the box, and -- with the
survey CPU's rcpps / rsqrtps tables given to the library -- with the survey's numbers directly."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from snail_amd import scenes, survey_camera      # noqa: E402
from tests import oracle_lib as O                # noqa: E402
from tests.golden.full_size import bench_light, host_table_key, sha   # noqa: E402

REF = "/root/reference/scenes"
OUT = os.path.join(HERE, "ref_meshes.json")
RES = (1920, 1080)


def digests(osc, cam13, mode, threads=8):
    resx, resy = RES
    t, u, v, tid, st = osc.render_primary(cam13, resx, resy, mode=mode, threads=threads)
    hit = np.isfinite(t)
    out = {"sha_t": sha(t), "sha_u": sha(u), "sha_v": sha(v), "sha_id": sha(tid), "stats": [int(x) for x in st], "hits": int(hit.sum()),
           "sum_id": int(tid[hit].astype(np.int64).sum()), "sum_t": round(float(t[hit].astype(np.float64).sum()), 3)}
    acc = osc.account_primary(cam13, resx, resy, mode=mode)
    out["hits_padded"] = int(acc[3])
    lights = bench_light(osc.nodes[0]["bmin"], osc.nodes[0]["bmax"])
    for key, refl in (("c3", False), ("c3r", True)):
        frame, wst = osc.render_whitted(cam13, resx, resy, lights, mode=mode, threads=threads, reflections=refl)
        out[key] = {"sha_bgr": sha(frame), "stats": [int(x) for x in wst]}
    return out


def main(names):
    d = json.load(open(OUT)) if os.path.exists(OUT) else {}
    d["what"] = ("oracle digests of the reference's own meshes (tests/golden/make_ref_mesh_fixture.py, build container): survey camera, 1920x1080; "
                 "ieee = ORC_MODE_IEEE; host_sse[key] = ORC_MODE_SSE on the CPU whose product-library tables hash to key; survey = what SURVEY.md "
                 "section 8(c) recorded from the reference itself on the build container's CPU")
    survey = json.load(open(os.path.join(HERE, "survey_digests.json")))
    key = host_table_key()
    for name in names:
        tv = scenes.load_obj(os.path.join(REF, name + ".obj"))
        np.savez_compressed(os.path.join(HERE, name + "_tris.npz"), tris=tv)
        osc = O.OracleScene(tv)
        cam13 = survey_camera(tv).as_array13()
        e = d.setdefault(name, {})
        e.update({"tris": int(len(tv)), "nodes": int(len(osc.nodes)), "depth": int(osc.depth), "fnv_nodes": "%016x" % osc.fnv_nodes(),
                  "fnv_tris": "%016x" % osc.fnv_tris(), "sha_tris_npz": hashlib.sha256(np.ascontiguousarray(tv).tobytes()).hexdigest(),
                  "camera": [float(x) for x in cam13], "res": list(RES), "survey": survey.get(name, {})})
        e["ieee"] = digests(osc, cam13, O.MODE_IEEE)
        e.setdefault("host_sse", {})[key] = digests(osc, cam13, O.MODE_SSE)
        print(name, len(tv), "tris", e["nodes"], "nodes; sse hits / sum_id / sum_t:", e["host_sse"][key]["hits"], e["host_sse"][key]["sum_id"], e["host_sse"][key]["sum_t"],
              "survey:", e["survey"], flush=True)
    json.dump(d, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:] or ["lancia"])
