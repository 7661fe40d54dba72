"""GPU: kernel 13 on poisoned inputs -- NaN / inf states, inputs, goals and obstacle rows, negative and huge radii, an obstacle on top of the
robot, a state far outside its bound: every instantiation returns (no hang: each solve ends by a test that a NaN cannot pass, or at the
iteration limit), the poisoned problems do not come back 'optimal' with a NaN input, and the clean problems of the same launch are solved as
if the others were not there."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402

DEV = "cuda:0"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fam", ["du", "uni", "si", "di", "kb"])
def test_poisoned_rows_end_and_leave_the_rest_alone(fam):
    name = W.MPC_FAMILIES[fam]
    X, up, goal, obs = (a[:64].copy() for a in W.mpc_family_batch(fam, 64, 8, seed=1))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)     # noqa: E731
    ctl = sca.BatchedMSMPCCBF({"model": name}, io_dtype="f64", max_iter=300)
    u0, s0, i0 = ctl.solve(t(X), t(up), t(goal), t(obs))
    X[0] = np.nan; X[1, 0] = np.inf; up[2] = np.nan; goal[3] = np.inf; obs[4, 0, 0] = np.nan
    obs[5, :, 2] = -1.0; obs[6, :, :2] = X[6, :2]; X[7] = 1e12; up[8] = 1e9; obs[9, :, 2] = 1e6
    if X.shape[1] >= 4:
        X[10, 3] = 50.0
    u, st, it = ctl.solve(t(X), t(up), t(goal), t(obs))
    torch.cuda.synchronize()
    assert (it <= 300).all() and (st[:5] != 0).all()
    ok = st == 0
    assert torch.isfinite(u[ok]).all()
    assert torch.equal(u[11:], u0[11:]) and torch.equal(st[11:], s0[11:]) and torch.equal(it[11:], i0[11:])
