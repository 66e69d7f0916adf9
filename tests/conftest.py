import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The package refuses to import without its HIP library (no fallback): build it in-tree if it is not there yet
# (hipcc cross-compiles gfx950 without a GPU).  Equivalent to __graft_entry__.build().
_SO = os.path.join(ROOT, "unet-studio_amd", "libunet_hip.so")
if not os.path.exists(_SO):
    subprocess.check_call(["bash", os.path.join(ROOT, "unet-studio_amd", "csrc", "build.sh")])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
