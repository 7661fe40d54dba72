"""CPU tests of the Unicycle2D MPC-CBF problem functions (oracle/mpc_cbf_uni.py).

Parity UNPINNED at the IPOPT boundary like oracle/mpc_cbf.py.  Checked here: the restated functions are
self-consistent (finite differences), the CBF rows are the reference's definition through the robot's own step,
and the oracle's solution is a feasible local optimum that scipy SLSQP cannot improve.
"""
import numpy as np
from scipy.optimize import minimize

from oracle import mpc_cbf_uni as U
from oracle import robots as R
from safe_control_amd import workloads as W

P = dict(U.DEFAULTS)


def case(i, seed=0, K=6):
    X, goal, ur, obs = W.du_cbfqp_batch(64, K, seed=seed)
    x0 = X[i].copy(); x0[3] = 0.0
    return x0, goal[i], obs[i]


def test_derivatives_by_finite_differences():
    x0, goal, obs = case(1)
    rng = np.random.default_rng(0)
    n, m = 20, 10 * 6 + 40
    z = rng.uniform(-0.4, 0.4, n); lam = rng.uniform(0, 2, m); up = rng.uniform(-0.2, 0.2, 2)
    ev = U.evaluate(x0, z, up, goal, obs, P, lam, level=2)
    assert ev["g"].shape == (m,)
    h = 1e-6
    f = lambda zz: U.evaluate(x0, zz, up, goal, obs, P, level=0)["f"]
    g = lambda zz: U.evaluate(x0, zz, up, goal, obs, P, level=0)["g"]
    gfd = np.array([(f(z + h * e) - f(z - h * e)) / (2 * h) for e in np.eye(n)])
    Jfd = np.array([(g(z + h * e) - g(z - h * e)) / (2 * h) for e in np.eye(n)]).T
    assert np.abs(gfd - ev["grad"]).max() <= 1e-5 * np.abs(gfd).max()
    assert np.abs(Jfd - ev["J"]).max() <= 1e-6

    def gL(zz):
        e = U.evaluate(x0, zz, up, goal, obs, P, level=1)
        return e["grad"] - e["J"].T @ lam
    Wfd = np.array([(gL(z + h * e) - gL(z - h * e)) / (2 * h) for e in np.eye(n)])
    assert np.abs(Wfd - ev["W"]).max() <= 1e-5 * max(1.0, np.abs(ev["W"]).max())
    assert np.abs(ev["W"] - ev["W"].T).max() < 1e-12


def test_cbf_row_equals_reference_definition():
    """row = h(step(x, u)) - h(x) + alpha h(x), circle barrier with beta = 1.01 (unicycle2D.py:127-145)."""
    x0, goal, obs = case(2)
    rng = np.random.default_rng(1)
    z = rng.uniform(-0.5, 0.5, 20)
    ev = U.evaluate(x0, z, np.zeros(2), goal, obs, P, level=0)
    X = ev["X"]
    spec = R.default_spec(R.MODEL_UNI)
    K = obs.shape[0]

    def h(x, ob):
        return (x[0] - ob[0]) ** 2 + (x[1] - ob[1]) ** 2 - 1.01 * (ob[2] + P["radius"]) ** 2
    for k in range(10):
        x1 = X[k] + 0.05 * np.array([z[2 * k] * np.cos(X[k, 2]), z[2 * k] * np.sin(X[k, 2]), z[2 * k + 1]])
        assert np.abs(x1 - X[k + 1]).max() < 1e-14
        for j in range(K):
            want = h(x1, obs[j]) - h(X[k], obs[j]) + P["alpha"] * h(X[k], obs[j])
            assert abs(ev["g"][k * K + j] - want) <= 1e-11 * max(1.0, abs(want))
    # rollout against the robot model's own step (oracle/robots.py, pinned on the reference by tests/golden)
    x = np.concatenate([X[0][:3], [0.0]])                  # oracle/robots.py keeps Unicycle2D rows 4 wide (last unused)
    for k in range(10):
        x = R.step(R.MODEL_UNI, x, z[2 * k:2 * k + 2], 0.05, spec)
        d = x[:3] - X[k + 1]
        d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi           # the robot wraps its heading
        assert np.abs(d).max() < 1e-12


def test_solution_is_feasible_and_slsqp_cannot_improve_it():
    n_opt = 0
    for i in range(8):
        x0, goal, obs = case(i, seed=5)
        up = np.zeros(2)
        u, st, it, info = U.solve(x0, up, goal, obs, return_info=True)
        if st != 0:
            continue
        n_opt += 1
        z = info["z"]
        ev = U.evaluate(x0, z, up, goal, obs, P, level=0)
        assert ev["g"].min() >= -1e-6
        assert abs(z[0]) <= P["a_max"] + 1e-9 and abs(z[1]) <= P["w_max"] + 1e-9
        if i < 4:
            fun = lambda zz: U.evaluate(x0, zz, up, goal, obs, P, level=0)["f"]
            con = lambda zz: U.evaluate(x0, zz, up, goal, obs, P, level=0)["g"]
            jac = lambda zz: U.evaluate(x0, zz, up, goal, obs, P, level=1)["grad"]
            cjac = lambda zz: U.evaluate(x0, zz, up, goal, obs, P, level=1)["J"]
            r = minimize(fun, z, jac=jac, constraints=[{"type": "ineq", "fun": con, "jac": cjac}],
                         method="SLSQP", options={"ftol": 1e-13, "maxiter": 100})
            assert r.fun >= ev["f"] * (1 - 1e-6) - 1e-6
    assert n_opt >= 6
