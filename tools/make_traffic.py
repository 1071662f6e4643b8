"""profiles/traffic.json from PMC summaries: per workload key the HBM bytes per frame (2 x FETCH_SIZE -- the gfx950 correction of MI355X_MICROARCH.md,
section HBM -- + WRITE_SIZE, KB -> bytes) and SQ_INSTS_VALU per frame, plus the hash of the kernel sources they were measured on.
Usage: python tools/make_traffic.py key=summary.txt [key=summary.txt:frames=N ...]
  key=summary.txt             a tools/pmc_run.sh summary: dev::k_primary alone (one frame per dispatch)
  key=summary.txt:frames=N    a tools/pmc_cmd.sh summary (every kernel, N frames profiled): the frame's kernels summed, per-kernel breakdown kept"""
import collections, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "traffic.json")
d = json.load(open(path)) if os.path.exists(path) else {}
for a in sys.argv[1:]:
    if "=" not in a: continue
    key, f = a.split("=", 1)
    frames = None
    if ":frames=" in f:
        f, n = f.split(":frames=")
        frames = int(n)
    if frames is None:
        v = {}
        for line in open(f):
            p = line.split()
            if len(p) >= 4 and p[3].startswith("mean="): v[p[1]] = float(p[3][5:])
        d[key] = {"bytes_per_launch": int(round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)), "valu_insts_per_launch": int(round(v["SQ_INSTS_VALU"])),
                  "salu_insts_per_launch": int(round(v.get("SQ_INSTS_SALU", 0))), "smem_insts_per_launch": int(round(v.get("SQ_INSTS_SMEM", 0))),
                  "lanes_live_per_valu_inst": round(v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"], 2) if "SQ_THREAD_CYCLES_VALU" in v else None,
                  "fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"], "kernel": "dev_sse::k_primary" if key.endswith("_sse") else "dev::k_primary",
                  "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE (%.1f KB per dispatch, doubled per the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md section HBM) + --pmc WRITE_SIZE (%.1f KB), "
                            "separate passes with --kernel-trace only, mean over the k_primary<false> dispatches (one frame each) of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-per-launch 1 --arith ieee|host_sse` of this workload; "
                            "SQ_INSTS_VALU from the sq1 pass" % (os.path.basename(f), v["FETCH_SIZE"], v["WRITE_SIZE"])}
    else:
        per = collections.defaultdict(dict)      # kernel -> counter -> per-frame total
        for line in open(f):
            if "kernel=" not in line: continue
            head, kname = line.rstrip("\n").split("kernel=", 1)
            p = head.split()
            if len(p) >= 4 and p[3].startswith("mean="):
                n = int(p[2][2:]); per[kname.strip()][p[1]] = float(p[3][5:]) * n / frames
        tot = collections.defaultdict(float)
        for k, v in per.items():
            for c, x in v.items(): tot[c] += x
        kern = {k: {"valu_insts": int(round(v.get("SQ_INSTS_VALU", 0))), "salu_insts": int(round(v.get("SQ_INSTS_SALU", 0))),
                    "hbm_bytes": int(round((2.0 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024)),
                    "lanes_live_per_valu_inst": round(v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"], 2) if v.get("SQ_INSTS_VALU") and "SQ_THREAD_CYCLES_VALU" in v else None,
                    "wait_share_of_wave_cycles": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in v else None}
                for k, v in sorted(per.items()) if v.get("SQ_INSTS_VALU", 0) > 0.002 * tot["SQ_INSTS_VALU"]}
        d[key] = {"bytes_per_launch": int(round((2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024)), "valu_insts_per_launch": int(round(tot["SQ_INSTS_VALU"])),
                  "salu_insts_per_launch": int(round(tot["SQ_INSTS_SALU"])), "kernel": " + ".join(sorted(kern)) + " (all kernels of a frame)", "kernels": kern,
                  "source": "profiles/%s: rocprofv3 --pmc passes (tools/pmc_cmd.sh: FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE, SQ_* groups; --kernel-trace only) of %d frames traced one at a time, "
                            "every kernel of the frame summed" % (os.path.basename(f), frames)}
    print(key, d[key]["bytes_per_launch"], d[key]["valu_insts_per_launch"])
h = hashlib.sha256()
for f in ("snail_dev.inc", "lbvh.inc", "host_sse.h", "Makefile"):     # bench.py kernel_source_sha16(): the sources these counters were measured on
    h.update(open(os.path.join(ROOT, "snail_amd", "csrc", f), "rb").read())
d["_kernel_sha16"] = h.hexdigest()[:16]
json.dump(d, open(path, "w"), indent=1)
