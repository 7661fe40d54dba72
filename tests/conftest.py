import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The HIP library is git-ignored: in a fresh checkout build it once (what __graft_entry__.build() does) so that the
    ABI tests have something to load.  hipcc cross-compiles for gfx950 without a GPU."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "safe_control_amd", "lib", "libsafe_control_hip.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C",
                         os.path.join(ROOT, "safe_control_amd", "csrc")])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
