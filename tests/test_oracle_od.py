"""CPU tests of the optimal-decay CBF-QP oracle (oracle/od_cbf_qp.py): exact enumeration vs scipy SLSQP."""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import od_cbf_qp as OD, robots as R
from safe_control_amd import workloads as W


@pytest.mark.parametrize("model", [R.MODEL_DU, R.MODEL_KB_C3BF, R.MODEL_QUAD2D])
def test_enumerator_agrees_with_slsqp(model):
    if model == R.MODEL_DU:
        X, goal, ur, obs = W.du_cbfqp_batch(48, 2, seed=2)
        spec = R.default_spec(model); spec.update(a_max=1.0, w_max=0.5)
    elif model == R.MODEL_QUAD2D:                                 # optimal_decay_cbf_qp.py:38-45,105-115
        Xd, goal, _, obs = W.du_cbfqp_batch(48, 2, seed=2)
        rng = np.random.default_rng(9)
        X = np.zeros((48, 6)); X[:, :2] = Xd[:, :2]; X[:, 2] = rng.uniform(-0.4, 0.4, 48); X[:, 3:5] = rng.uniform(-1.5, 1.5, (48, 2))
        ur = rng.uniform(2.0, 11.0, (48, 2))
        spec = R.default_spec(model)
    else:
        X, goal, ur, obs = W.kb_c3bf_batch(48, 2, seed=2)
        spec = R.default_spec(model)
    from oracle.cbf_qp import input_bounds
    lo, hi = input_bounds(model, spec)
    worst, n_skip = 0.0, 0
    for i in range(48):
        uref = ur[i] * (3.0 if i % 3 == 0 else 1.0)
        r = OD.solve(model, X[i], uref, obs[i, 0], spec)
        assert r["status"] == 0
        fx, gx = R.f(model, X[i], spec), R.g(model, X[i], spec)
        if model in (R.MODEL_DU, R.MODEL_QUAD2D):
            h, hdot, d = R.agent_barrier(model, X[i], obs[i, 0], spec["radius"])
            A, b = d @ gx, d @ fx
            rr = np.array([uref[0], uref[1], 1.0, 1.0]); D = np.array([1, 1, 1e4, 1e4])
            con = lambda x: np.array([A @ x[:2] + b + 1.0 * hdot * x[2] + 0.25 * h * x[3],
                                      x[0] - lo[0], hi[0] - x[0], x[1] - lo[1], hi[1] - x[1]])
        else:
            h, d = R.agent_barrier(model, X[i], obs[i, 0], spec["radius"])
            A, b = d @ gx, d @ fx
            rr = np.array([uref[0], uref[1], 1.0]); D = np.array([1, 1, 1e4])
            con = lambda x: np.array([A @ x[:2] + b + 0.5 * h * x[2], x[0] - lo[0], hi[0] - x[0], x[1] - lo[1], hi[1] - x[1]])
        s = minimize(lambda x: np.sum(D * (x - rr) ** 2), rr.copy(), constraints=[{"type": "ineq", "fun": con}],
                     method="SLSQP", options={"ftol": 1e-15, "maxiter": 500})
        if con(s.x).min() < -1e-7:                        # SLSQP gave up at an infeasible point ("positive directional derivative")
            n_skip += 1
            continue
        worst = max(worst, np.abs(s.x[:2] - r["u"]).max(), abs(s.x[2] - r["omega"][0]))
        assert np.sum(D * (np.concatenate([r["u"], r["omega"]])[: len(D)] - rr) ** 2) <= s.fun + 1e-7
    assert worst < 5e-6 and n_skip <= 6


def test_no_obstacle_is_box_projection():
    spec = R.default_spec(R.MODEL_DU); spec.update(a_max=1.0, w_max=0.5)
    r = OD.solve(R.MODEL_DU, np.array([1.0, 2.0, 0.3, 0.5]), np.array([3.0, -2.0]), None, spec)
    assert r["status"] == 0 and np.allclose(r["u"], [1.0, -0.5]) and np.allclose(r["omega"], [1.0, 1.0])
