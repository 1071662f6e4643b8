"""profiles/traffic.json from PMC summaries (tools/pmc_run.sh): per workload key the HBM bytes per launch of dev::k_primary
(2 x FETCH_SIZE -- the gfx950 correction of MI355X_MICROARCH.md, section HBM -- + WRITE_SIZE, KB -> bytes) and SQ_INSTS_VALU per launch.
Usage: python tools/make_traffic.py key=summary.txt [key=summary.txt ...] [--command "..."]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "traffic.json")
d = json.load(open(path)) if os.path.exists(path) else {}
for a in sys.argv[1:]:
    if "=" not in a: continue
    key, f = a.split("=", 1)
    v = {}
    for line in open(f):
        p = line.split()
        if len(p) >= 4 and p[3].startswith("mean="): v[p[1]] = float(p[3][5:])
    d[key] = {"bytes_per_launch": int(round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)), "valu_insts_per_launch": int(round(v["SQ_INSTS_VALU"])),
              "salu_insts_per_launch": int(round(v.get("SQ_INSTS_SALU", 0))), "smem_insts_per_launch": int(round(v.get("SQ_INSTS_SMEM", 0))),
              "fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"],
              "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE (%.1f KB per dispatch, doubled per the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md section HBM) + --pmc WRITE_SIZE (%.1f KB), "
                        "separate passes with --kernel-trace only, mean over the dev::k_primary<false> dispatches (one frame each) of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-per-launch 1` of this workload; "
                        "SQ_INSTS_VALU from the sq1 pass" % (os.path.basename(f), v["FETCH_SIZE"], v["WRITE_SIZE"])}
    print(key, d[key]["bytes_per_launch"], d[key]["valu_insts_per_launch"])
import hashlib
h = hashlib.sha256()
for f in ("snail_hip.hip", "lbvh.inc", "render_host.inc"):     # bench.py kernel_source_sha16(): the sources these counters were measured on
    h.update(open(os.path.join(ROOT, "snail_amd", "csrc", f), "rb").read())
d["_kernel_sha16"] = h.hexdigest()[:16]
json.dump(d, open(path, "w"), indent=1)
