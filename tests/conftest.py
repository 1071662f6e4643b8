import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (*.so are git-ignored): build them once, in-tree, before collecting
    need = [os.path.join(ROOT, "snail_amd", "libsnailhip.so"), os.path.join(ROOT, "snail_amd", "libsnailhip_debug.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def reference_scenes():
    """Directory of the reference's OBJ assets; only present in the build container (never on the GPU box)."""
    p = "/root/reference/scenes"
    if not os.path.isdir(p):
        pytest.skip("reference checkout not present")
    return p
