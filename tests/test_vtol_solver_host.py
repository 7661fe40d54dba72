"""CPU: the per-lane VTOL2D solver of the HIP kernel (safe_control_amd/csrc/mpc_vtol_solver.hpp: stage-wise costate sweep and Riccati
recursion, second-order forward mode through the aero model, the interior point with restoration and slack reset) compiled FOR THE HOST
by tools/vtol_host.cpp and held to the numpy oracle (condensed single shooting, dense Cholesky) on problems of the vtol workload batch.
The host build is a test / debugging aid: nothing on the product path loads it (the product path is the gfx950 kernel,
tests/test_mpcvtol_gpu.py)."""
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


@pytest.fixture(scope="module")
def host():
    import dbg_vtol_host as Dh
    return Dh, Dh.build()


def test_host_build_of_the_lane_solver_follows_the_oracle(host):
    Dh, lib = host
    from oracle import mpc_vtol as V
    from safe_control_amd import workloads as W
    X, up, goal, obs = W.mpc_family_batch("vtol", 4096, 8, seed=0)
    for i in (0, 2):                                                        # #2 runs 22 iterations, #0 27
        uh, sh, ih, zh = Dh.host_solve(lib, X[i], up[i], goal[i], obs[i])
        uo, so, io, info = V.solve(X[i], up[i], goal[i], obs[i], return_info=True)
        assert (sh, ih) == (so, io) and so == 0
        assert np.abs(uh - uo).max() <= 1e-9 and np.abs(zh - info["z"]).max() <= 1e-7


def test_host_build_without_slack_reset_does_not_converge_either(host):
    """The same switches as in the oracle: slack_reset = 0 leaves the cruise problem unsolved after 60 iterations."""
    Dh, lib = host
    from safe_control_amd import workloads as W
    X, up, goal, obs = W.mpc_family_batch("vtol", 4096, 8, seed=0)
    _, st2, it2, _ = Dh.host_solve(lib, X[0], up[0], goal[0], obs[0], slack_reset=2, max_iter=60)
    _, st0, it0, _ = Dh.host_solve(lib, X[0], up[0], goal[0], obs[0], slack_reset=0, max_iter=60)
    assert st2 == 0 and it2 < 60 and (st0 != 0 or it0 > it2)
