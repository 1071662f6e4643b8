"""Digests of the ORACLE's frames at the bench's own sizes (tests/golden/oracle_full_size.json): what bench.py's timed path must have
produced.  `python tests/golden/full_size.py ieee` (build container: ORC_MODE_IEEE, CPU-independent) rewrites the "ieee" section;
`python tests/golden/full_size.py host_sse [out.json]` (on the machine whose CPU the digests are for: ORC_MODE_SSE runs this CPU's
rcpps / rsqrtps) writes / merges the section keyed by that CPU's rcpps / rsqrtps tables (host_table_key).

Workloads = bench.py's CONFIGS with bench.py's cameras and light: per workload SHA-256 of the row-major t, u, v, triId planes, of the
gVals[1] depth-shaded B,G,R frame (what the tile-sharded route gathers) and the TreeStats; config 3: SHA-256 of the B,G,R frame."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from snail_amd import FPSCamera, scenes      # noqa: E402
from tests import oracle_lib as O            # noqa: E402

PATH = os.path.join(HERE, "oracle_full_size.json")
WORKLOADS = {"1": ("atrium", 1920, 1080, 0, False), "3": ("atrium", 1920, 1080, 1, False), "3r": ("atrium", 1920, 1080, 1, True),     # bench.py CONFIGS (3r = --config 3 --reflections)
             "4": ("atrium", 3840, 2160, 0, False), "5": ("stress", 1920, 1080, 0, False)}


def host_table_key():
    """sha256[:16] of the host CPU's rcpps / rsqrtps tables as the PRODUCT library took them (snail_host_sse_tables): the key bench.py
    looks its host_sse digests up by (no oracle involved on that side)."""
    import ctypes as C
    from snail_amd._lib import check, lib
    tab = np.zeros(3 * 4096, dtype=np.uint32)
    check(lib().snail_host_sse_tables(tab.ctypes.data_as(C.c_void_p)), "snail_host_sse_tables")
    return hashlib.sha256(tab.tobytes()).hexdigest()[:16]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bench_camera(scene_name):
    return FPSCamera(*(scenes.stress_camera() if scene_name.startswith("stress") else scenes.atrium_camera())).camera()


def bench_light(bmin, bmax):
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    return np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)   # bench.py --config 3's light


def digests(mode, threads=8, configs=("1", "3", "3r", "4", "5")):
    out, cache = {}, {}
    for c in configs:
        name, resx, resy, nl, refl = WORKLOADS[c]
        if name not in cache:
            cache[name] = O.OracleScene(scenes.scene_by_name(name))
        osc = cache[name]
        cam13 = bench_camera(name).as_array13()
        key = "%s_%dx%d_c%s" % (name, resx, resy, c)
        if nl:
            lights = bench_light(osc.nodes[0]["bmin"], osc.nodes[0]["bmax"])
            frame, st = osc.render_whitted(cam13, resx, resy, lights, mode=mode, threads=threads, reflections=refl)
            out[key] = {"sha_bgr": sha(frame), "stats": [int(x) for x in st]}
        else:
            t, u, v, tid, st = osc.render_primary(cam13, resx, resy, mode=mode, threads=threads)
            out[key] = {"sha_t": sha(t), "sha_u": sha(u), "sha_v": sha(v), "sha_id": sha(tid), "stats": [int(x) for x in st], "hits": int(np.isfinite(t).sum()),
                        "sha_depth_bgr": sha(O.shade_depth(t, mode=mode).reshape(resy, resx, 3))}
        print(key, out[key], flush=True)
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "ieee"
    d = json.load(open(PATH)) if os.path.exists(PATH) else {}
    d.setdefault("what", "SHA-256 of the oracle's full-size frames for bench.py's workloads (tests/golden/full_size.py); ieee = ORC_MODE_IEEE, CPU-independent, made in the "
                         "build container; host_sse = ORC_MODE_SSE per CPU (key = sha256[:16] of the tables of snail_host_sse_tables), made on that CPU")
    if which == "ieee":
        d["ieee"] = digests(O.MODE_IEEE)
        json.dump(d, open(PATH, "w"), indent=1)
    else:
        fp = host_table_key()
        sec = {fp: digests(O.MODE_SSE, threads=min(16, os.cpu_count() or 1))}
        dst = sys.argv[2] if len(sys.argv) > 2 else None
        if dst:
            json.dump(sec, open(dst, "w"), indent=1)
        else:
            d.setdefault("host_sse", {}).update(sec)
            json.dump(d, open(PATH, "w"), indent=1)
