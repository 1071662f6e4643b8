"""Records tests/golden/sse_bounds.json: the counts of tests/test_gpu_parity.py::test_sse_path_full_size_counted measured on this box,
with head-room (x86 vendors differ in rcpps / rsqrtps; the bound is 4x the observed count + 64).  Usage: python tools/sse_counts.py [--write]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_gpu_parity import sse_path_counts
out, bounds = {}, {}
for name in ("atrium", "stress"):
    r = sse_path_counts(torch, name)
    out[name] = r
    keys = ("hit_miss_flips", "triId_mismatches", "t_outside_tol_same_tri", "uv_outside_tol_same_tri", "mismatches_not_near_tie")
    bounds[name] = {leg: dict({k + "_max": (0 if leg == "same_rays" and k != "triId_mismatches" else 4 * r[leg][k] + 64) for k in keys}, observed=r[leg]) for leg in r}
    print(name, json.dumps(r))
bounds["note"] = "observed on an MI355X box (host CPU: %s) by tools/sse_counts.py; bound = 4 x observed + 64 (0 for hit/miss flips on the same-rays leg)" % (
    [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0])
if "--write" in sys.argv:
    p = os.path.join(ROOT, "gpurun_out", "sse_bounds.json")
    json.dump(bounds, open(p, "w"), indent=1)
    print("wrote", p)
