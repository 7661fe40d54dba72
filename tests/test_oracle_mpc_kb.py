"""CPU: the KinematicBicycle2D-family MPC-CBF oracles (oracle/mpc_gn.py: kb_model; oracle/mpc_kb_state.py: C3BF / DPCBF) beyond the pinned
problem functions of tests/test_oracle_mpc_golden.py: the exact Hessian of the Lagrangian against finite differences, the solver's
optimum against scipy's SLSQP, and the reference scene's closed loop."""
import math

import numpy as np
import pytest

from oracle import mpc_cbf as M
from oracle import mpc_gn as G
from oracle import mpc_kb_state as S


@pytest.mark.parametrize("mk", [S.c3bf_model, S.dpcbf_model])
def test_exact_hessian_of_the_full_state_barrier_problems(mk):
    """W = d/dz (grad f - J' lam) at fixed multipliers: barrier Hessians from the second-order forward mode, second derivatives of the
    bicycle and of step() weighted by the costates (oracle/mpc_kb_state.py: evaluate).  One state is past the v_max clip of step()."""
    mdl = mk()
    rng = np.random.default_rng(1)
    N = 5
    P = S.params(mdl, N)
    for x0 in (np.array([1.0, 1.0, 0.4, 1.5]), np.array([1.0, 1.0, -0.3, 3.45])):
        goal = np.array([4.0, 3.0]); obs = np.array([[2.6, 2.4, 0.4, 0.3, -0.1, 0, 0], [3.5, 0.2, 0.5, 0, 0, 0, 0]])
        lo, hi = mdl["u_lo"], mdl["u_hi"]
        z = np.tile((lo + hi) / 2, N) + rng.normal(size=2 * N) * 0.1 * np.tile(hi - lo, N)
        z[0::2] = np.abs(z[0::2])                                               # accelerating: the second start clips from stage 1 on
        up = np.zeros(2)
        m = S.evaluate(x0, z, up, goal, obs, P, level=0)["g"].shape[0]
        lam = rng.uniform(0, 1, m)
        W = S.evaluate(x0, z, up, goal, obs, P, lam, 2)["W"]

        def gl(zz):
            e = S.evaluate(x0, zz, up, goal, obs, P, level=1)
            return e["grad"] - e["J"].T @ lam
        eps = 1e-6
        Wfd = np.zeros_like(W)
        for i in range(2 * N):
            d = np.zeros(2 * N); d[i] = eps
            Wfd[:, i] = (gl(z + d) - gl(z - d)) / (2 * eps)
        assert np.abs(Wfd - W).max() <= 2e-7 * np.abs(Wfd).max(), (mdl["name"], x0)
        assert np.abs(W - W.T).max() <= 1e-12 * np.abs(W).max()


def _draw(mdl, rng, K=3):
    x0 = np.zeros(4); x0[:2] = rng.uniform(2, 10, 2)
    goal = rng.uniform(2, 10, 2)
    x0[2] = math.atan2(goal[1] - x0[1], goal[0] - x0[0]) + rng.uniform(-0.5, 0.5); x0[3] = rng.uniform(0.6, 2.5)
    obs = np.zeros((K, 7))
    for j in range(K):
        r = rng.uniform(0.2, 0.8); rho = rng.uniform(r + 1.5, 5.0); ph = rng.uniform(-np.pi, np.pi)
        obs[j, :3] = [x0[0] + rho * np.cos(ph), x0[1] + rho * np.sin(ph), r]
    return x0, goal, obs


@pytest.mark.parametrize("mk,ev,sol", [(G.kb_model, G.evaluate, G.solve), (S.c3bf_model, S.evaluate, S.solve)])
def test_solver_reaches_a_local_optimum_slsqp_cannot_improve(mk, ev, sol):
    from scipy.optimize import minimize
    mdl = mk()
    rng = np.random.default_rng(5)
    n_ok = 0
    for t in range(6):
        x0, goal, obs = _draw(mdl, rng)
        N = 6
        P = (G if ev is G.evaluate else S).params(mdl, N)
        up = np.zeros(2)
        u0, st, it, info = sol(mdl, x0, up, goal, obs, N=N, return_info=True)
        if st != M.STATUS_OPTIMAL:
            continue
        assert np.min(info["g"]) >= -1e-6
        fun = lambda z: ev(x0, z, up, goal, obs, P, level=1)
        r = minimize(lambda z: fun(z)["f"], info["z"], jac=lambda z: fun(z)["grad"],
                     constraints=[{"type": "ineq", "fun": lambda z: fun(z)["g"], "jac": lambda z: fun(z)["J"]}],
                     method="SLSQP", options={"ftol": 1e-13, "maxiter": 100})
        assert r.fun >= info["f"] * (1 - 1e-6) - 1e-6
        n_ok += 1
    assert n_ok >= 3


def test_reference_scene_closed_loop_kinematic_bicycle():
    """examples/test_tracking.py --model kb, first 60 control steps with the oracle as position controller: optimal throughout, the
    bicycle accelerates north and keeps its clearance."""
    obs_all = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0], [4.0, 3.5, 1.5],
                        [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8], [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
    mdl = G.kb_model(dict(a_max=0.5, radius=0.5))
    x = np.array([2.0, 2.0, math.pi / 2, 1.0]); up = np.zeros(2)
    n_opt = 0
    for step in range(60):
        near = np.argsort(np.linalg.norm(obs_all[:, :2] - x[:2], axis=1) - obs_all[:, 2])[:5]
        u, st, it = G.solve(mdl, x, up, np.array([2.0, 12.0]), M.pad_obstacles(obs_all[near], 5))
        n_opt += int(st == 0)
        x = G.kb_S(x, u, mdl["spec"], mdl["dt"]); up = u
        assert (np.linalg.norm(obs_all[:, :2] - x[:2], axis=1) - obs_all[:, 2] - 0.5).min() > 0.0
    assert n_opt >= 58 and x[1] > 6.0 and x[3] > 2.0
