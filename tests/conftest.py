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
    """The HIP library is git-ignored: build it (what __graft_entry__.build() does) so that the ABI tests have something
    to load -- and RE-build it after any edit of csrc/ (make is incremental and tracks the headers), so the tests never
    run against a stale library.  hipcc cross-compiles for gfx950 without a GPU.  A failed build stops the session."""
    import shutil
    import subprocess
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return                                               # no compiler on this box: the shipped .so is what there is
    lib = os.path.join(ROOT, "safe_control_amd", "lib", "libsafe_control_hip.so")
    if os.path.exists(lib) and not os.path.isdir(os.path.join(ROOT, "build", "csrc")):
        return                                               # shipped library without its object files (the GPU box: build/ does
                                                             # not travel): nothing was edited there, do not spend minutes recompiling
    for sub in (os.path.join("safe_control_amd", "csrc"), "oracle"):
        rc = subprocess.call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(ROOT, sub)])
        if rc != 0:
            pytest.exit(f"make -C {sub} failed with exit code {rc}: refusing to test a stale library", returncode=2)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
