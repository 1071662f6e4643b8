"""What the runtime says about dev::k_primary's residency (snail_debug_occupancy)."""
import os as _os
_os.environ.setdefault("SNAIL_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "snail_amd", "libsnailhip_debug.so"))  # workbench build (snail_debug_*)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import _lib
torch.cuda.init(); torch.zeros(1, device="cuda")
out = np.zeros(4, dtype=np.int32)
_lib.check(_lib.lib().snail_debug_occupancy(_lib.ptr(out)), "snail_debug_occupancy")
print("k_primary: occupancy API admits %d blocks of %d wave(s) per CU (= %.1f waves/SIMD); device limit %d blocks per CU; %d CUs" % (
    out[0], out[3], out[0] * out[3] / 4.0, out[1], out[2]))
