"""CPU tests of the MPC-CBF oracle (oracle/mpc_cbf.py).

Parity UNPINNED at the IPOPT boundary (the reference's solver stack is not installable, the NLP is
non-convex, the reference has no test that pins its result).  What can be checked here: the
restated problem functions are self-consistent (finite differences), and the oracle's solution is
a local optimum that an independent solver (scipy SLSQP, same single-shooting functions, same
constant initial guess) also reaches.
"""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import mpc_cbf as M
from safe_control_amd import workloads as W

P = dict(M.DEFAULTS)


def case(i, seed=0, superell=False):
    X, goal, ur, obs = W.du_cbfqp_batch(64, 8, seed=seed)
    o = obs[i].copy()
    if superell:
        o[1] = [X[i, 0] + 2.0, X[i, 1] + 0.5, 0.6, 0.9, 4.0, 0.7, 1.0]
    return X[i], goal[i], o


@pytest.mark.parametrize("superell", [False, True])
def test_derivatives_by_finite_differences(superell):
    x0, goal, obs = case(1, superell=superell)
    rng = np.random.default_rng(0)
    z = rng.uniform(-0.4, 0.4, 20); lam = rng.uniform(0, 2, 140); up = rng.uniform(-0.2, 0.2, 2)
    ev = M.evaluate(x0, z, up, goal, obs, P, lam, level=2)
    h = 1e-6
    f = lambda zz: M.evaluate(x0, zz, up, goal, obs, P, level=0)["f"]
    g = lambda zz: M.evaluate(x0, zz, up, goal, obs, P, level=0)["g"]
    gfd = np.array([(f(z + h * e) - f(z - h * e)) / (2 * h) for e in np.eye(20)])
    Jfd = np.array([(g(z + h * e) - g(z - h * e)) / (2 * h) for e in np.eye(20)]).T
    assert np.abs(gfd - ev["grad"]).max() <= 1e-5 * np.abs(gfd).max()
    assert np.abs(Jfd - ev["J"]).max() <= 1e-6

    def gL(zz):
        e = M.evaluate(x0, zz, up, goal, obs, P, level=1)
        return e["grad"] - e["J"].T @ lam
    Wfd = np.array([(gL(z + h * e) - gL(z - h * e)) / (2 * h) for e in np.eye(20)])
    assert np.abs(Wfd - ev["W"]).max() <= 1e-5 * max(1.0, np.abs(ev["W"]).max())
    assert np.abs(ev["W"] - ev["W"].T).max() < 1e-12


def test_cbf_row_equals_reference_definition():
    """g_cbf = dd_h + (a1+a2) d_h + a1 a2 h with x1 = step(x,u), x2 = step(x1,u) (dynamic_unicycle2D.py:188-238)."""
    from oracle import robots as R
    x0, goal, obs = case(2)
    rng = np.random.default_rng(1)
    z = rng.uniform(-0.5, 0.5, 20)
    ev = M.evaluate(x0, z, np.zeros(2), goal, obs, P, level=0)
    X = ev["X"]
    spec = R.default_spec(R.MODEL_DU)
    for k in range(10):
        u = z[2 * k:2 * k + 2]
        x1 = R.step(R.MODEL_DU, X[k], u, 0.05, spec)          # with the angle wrap
        x2 = R.step(R.MODEL_DU, x1, u, 0.05, spec)
        for j in range(8):
            h = lambda x: (x[0] - obs[j, 0]) ** 2 + (x[1] - obs[j, 1]) ** 2 - 1.01 * (0.25 + obs[j, 2]) ** 2
            hk, h1, h2 = h(X[k]), h(x1), h(x2)
            want = (h2 - 2 * h1 + hk) + 0.3 * (h1 - hk) + 0.0225 * hk
            assert abs(ev["g"][k * 8 + j] - want) <= 1e-9 * max(1.0, abs(want))


@pytest.mark.parametrize("i", [0, 3, 5, 9, 11, 14])
def test_oracle_solution_is_a_local_optimum_slsqp_agrees(i):
    x0, goal, obs = case(i)
    u0, st, it, info = M.solve(x0, np.zeros(2), goal, obs, return_info=True)
    assert st == M.STATUS_OPTIMAL and it < 60
    assert info["g"].min() >= -1e-7
    fun = lambda z: M.evaluate(x0, z, np.zeros(2), goal, obs, P, level=0)["f"]
    con = lambda z: M.evaluate(x0, z, np.zeros(2), goal, obs, P, level=0)["g"]
    jac = lambda z: M.evaluate(x0, z, np.zeros(2), goal, obs, P, level=1)["grad"]
    cjac = lambda z: M.evaluate(x0, z, np.zeros(2), goal, obs, P, level=1)["J"]
    r = minimize(fun, np.zeros(20), jac=jac, constraints=[{"type": "ineq", "fun": con, "jac": cjac}],
                 method="SLSQP", options={"ftol": 1e-12, "maxiter": 300})
    assert abs(info["f"] - r.fun) <= 1e-6 * max(1.0, abs(r.fun))
    assert np.abs(u0 - r.x[:2]).max() <= 1e-5


def test_config3_optima_agree_with_an_independent_solver_on_200_problems():
    """BASELINE config 3 draws (seed 0, first 230 agents => ~200 feasible): scipy SLSQP from the reference's constant
    initial guess (z = 0), with every inequality (CBF rows, speed box, input box) in play, reaches the same cost (1e-7
    relative) and the same first move (1e-5) as the oracle's interior point.  The NLP is non-convex: a different local
    optimum is legitimate, so up to 1 % of the problems may differ -- each of those must still be a KKT point."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from _mpc_checks import kkt_residual, slsqp_from
    from _oracle_pool import mpc_cbf_solve_many
    B = 230
    X, goal, ur, obs = W.du_cbfqp_batch(4096, 8, seed=0)
    X, goal, obs = X[:B], goal[:B], obs[:B]
    up = np.zeros((B, 2))
    u, st, it, z, f = mpc_cbf_solve_many(X, up, goal, obs, workers=min(8, os.cpu_count() or 1))
    opt = np.flatnonzero(st == M.STATUS_OPTIMAL)
    assert len(opt) >= 200
    differ, active = [], 0
    for i in opt:
        ev = lambda zz, lvl: M.evaluate(X[i], zz, up[i], goal[i], obs[i], P, level=lvl)
        r = slsqp_from(ev, np.zeros(20))
        g_s = ev(r.x, 0)["g"]
        assert g_s.min() >= -1e-6                                           # SLSQP's own feasibility tolerance
        active += int((ev(z[i], 0)["g"][:80] < 1e-6).any())
        same = abs(r.fun - f[i]) <= 1e-7 * max(1.0, abs(f[i])) and np.abs(r.x[:2] - u[i]).max() <= 1e-5
        if not same:
            differ.append(i)
            res, gmin, _ = kkt_residual(ev, z[i], active_tol=1e-4)
            assert res <= 1e-3 and gmin >= -1e-7, (i, res, gmin)        # the oracle's point is a KKT point all the same
    assert len(differ) <= 0.01 * len(opt), differ
    assert active >= 25                                                     # the agreement includes problems with active CBF rows


def test_infeasible_start_is_reported():
    """An agent already violating the DT-CBF row at k = 0 beyond the input authority: no feasible point."""
    x0, goal, obs = case(7)
    u0, st, it = M.solve(x0, np.zeros(2), goal, obs)
    assert st == M.STATUS_INFEASIBLE


def test_pad_obstacles_like_update_tvp():
    out = M.pad_obstacles([[1.0, 2.0, 0.3], [3.0, 4.0, 0.5, 0, 0, 0, 0]], 4)
    assert out.shape == (4, 7) and np.all(out[2:] == M.DUMMY_OBS) and np.all(out[0] == [1, 2, 0.3, 0, 0, 0, 0])
    assert np.all(M.pad_obstacles(None, 3) == M.DUMMY_OBS)
    assert M.pad_obstacles(np.zeros((9, 7)), 5).shape == (5, 7)
    with pytest.raises(ValueError):
        M.pad_obstacles([[1.0, 2.0, 0.3, 0.0, 0.0]], 4)
