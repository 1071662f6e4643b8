"""Helper of tests/test_gpu_parity.py::test_non_nested_caller_tree: run as a child process on the WORKBENCH build with
SNAIL_DEBUG_ASSUME_NESTED=1, i.e. with the record-prefetching node loop forced onto a tree whose child boxes stick out of their parents'
(the product library never does that: snail_scene_create routes such a tree to the loop that rescans the inherited range).  Prints one JSON
line saying whether the frame and its TreeStats still equal the oracle's -- they must NOT, or the test scene does not discriminate."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from snail_amd.scene import Scene
from tests import oracle_lib as O
from tests import util


def main():
    name, resx, resy = "atrium:0.02", 328, 200
    tv, hb, _ = util.scene_pair(name)
    hb2 = util.non_nested_tree(hb)
    osc = O.OracleScene.from_arrays(hb2.tris, hb2.nodes, hb2.depth, hb2.perm)
    cam = util.camera_for(name, tv)
    sc = Scene(hb2, 0)
    st = sc.new_stats()
    fr = sc.trace_primary(cam, resx, resy, stats=st)
    torch.cuda.synchronize()
    t, u, v, tid, ost = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE)
    same = bool(np.array_equal(fr.t.cpu().numpy().view(np.uint32), t.view(np.uint32)) and np.array_equal(fr.tri_id.cpu().numpy(), tid)
                and np.array_equal(st.cpu().numpy().astype(np.uint64), ost))
    print(json.dumps({"assume_nested": os.environ.get("SNAIL_DEBUG_ASSUME_NESTED"), "flags": list(sc.flags()), "equal_to_oracle": same}))
    sc.close()


if __name__ == "__main__":
    main()
