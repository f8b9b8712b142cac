import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """The C-ABI library must exist for every test tier (hipcc cross-compiles gfx950 without a GPU)."""
    csrc = os.path.join(ROOT, "3d-condtional-stable-diffusion_amd", "csrc")
    so = os.path.join(csrc, "libdm3d_hip.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "dm3d.h"))
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", csrc, "-j4"], check=True)
    return so
