"""Records the counts of tests/test_gpu_parity.py::test_sse_path_full_size_counted measured on this box into tests/golden/sse_bounds.json
form, keyed by this host CPU's rcpps / rsqrtps fingerprint (the test asserts equalities on a CPU it knows).
Usage: python tools/sse_counts.py [--write]   (--write: merged copy at gpurun_out/sse_bounds.json, to be committed as tests/golden/sse_bounds.json)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_gpu_parity import sse_path_counts
from tests.test_oracle_pins import rcp_fingerprint
fp = rcp_fingerprint()
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
entry = {"cpu": cpu}
for name in ("atrium", "stress"):
    entry[name] = sse_path_counts(torch, name)
    print(name, json.dumps(entry[name]))
print("fingerprint", fp, cpu)
if "--write" in sys.argv:
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "sse_bounds.json")))
    d["by_cpu"][fp] = entry
    p = os.path.join(ROOT, "gpurun_out", "sse_bounds.json")
    os.makedirs(os.path.dirname(p), exist_ok=True)
    json.dump(d, open(p, "w"), indent=1)
    print("wrote", p)
