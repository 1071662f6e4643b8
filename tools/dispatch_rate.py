"""What the workgroup dispatcher sustains with empty kernels: ms per launch and workgroups per microsecond for the grid shapes
of the primary kernel (one wave per block) and fatter blocks."""
import os as _os
_os.environ.setdefault("SNAIL_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "snail_amd", "libsnailhip_debug.so"))  # workbench build (snail_debug_*)

import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snail_amd import _lib
torch.cuda.init(); torch.zeros(1, device="cuda")
L = _lib.lib()
for blocks, threads in ((8192, 64), (65536, 64), (16384, 256), (65536, 256), (4096, 1024)):
    ms = C.c_float(0)
    _lib.check(L.snail_debug_dispatch_rate(blocks, threads, 20, C.addressof(ms)), "snail_debug_dispatch_rate")
    print("%6d blocks x %4d threads: %.4f ms per launch = %.1f workgroups/us = %.1f waves/us" % (blocks, threads, ms.value, blocks / ms.value / 1e3, blocks * (threads // 64) / ms.value / 1e3))
