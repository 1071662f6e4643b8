"""tests/golden/rcp_tables.npz: the rcpps / rsqrtps tables (uint32[3, 4096]: T_rcp, T_rsqrt for [1, 2), T_rsqrt for [2, 4); snail_amd/csrc/host_sse.h) of
the CPUs seen so far, as the product library took them from each CPU (snail_host_sse_tables), for snail_arith_set_tables / `bench.py --arith-tables`.
  python tests/golden/make_rcp_tables.py NAME      adds / replaces entry NAME with the tables of THE CPU THIS RUNS ON
Entries: xeon_skylake_sp -- the build container (Intel Xeon @ 2.10 GHz; the CPU the survey ran the reference on); epyc_9575f -- the GPU pool's hosts (made
there: the entry equals the expansion of tools/probe/rcp_probe.c's segment dump of that CPU; its key 8c2e10204764ffab is the one bench.py's host_sse digests
were recorded under on those boxes)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from snail_amd.scene import host_sse_tables      # noqa: E402

if __name__ == "__main__":
    name = sys.argv[1]
    path = os.path.join(HERE, "rcp_tables.npz")
    d = dict(np.load(path)) if os.path.exists(path) else {}
    d[name] = host_sse_tables()
    np.savez_compressed(path, **d)
    for k, v in d.items():
        print(k, hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16])
