"""CPU: oracle/mpc_gn.py (MPC-CBF problem functions of DoubleIntegrator2D / Quad2D with the robot's own step() inside the
barrier, Gauss-Newton Hessian).  Model maps against the reference's own step (tests/golden: integrators.npz, quad2d.npz);
derivatives against finite differences; the solver's optimum against scipy SLSQP."""
import os

import numpy as np
import pytest

from oracle import mpc_cbf as M
from oracle import mpc_gn as G

GD = os.path.join(os.path.dirname(__file__), "golden")
DT = 0.05


def test_step_maps_match_reference():
    g = np.load(os.path.join(GD, "integrators.npz"))
    spec = dict(v_max=1.0, a_max=1.5, radius=0.25)                              # tests/golden/make_golden.py: gen_integrators
    X, U, st = g["DoubleIntegrator2D/X"], g["DoubleIntegrator2D/U"], g["DoubleIntegrator2D/step"]
    n_clip = 0
    for i in range(X.shape[0]):
        np.testing.assert_allclose(G.di_S(X[i, :4], U[i], spec, DT), st[i, :4], rtol=0, atol=1e-13)
        n_clip += int(np.hypot(*(X[i, 2:4] + DT * U[i])) > 1.0)
    assert n_clip > 0                                                           # the speed rescaling branch is exercised
    q = np.load(os.path.join(GD, "quad2d.npz"))
    spec = dict(mass=q["Quad2D/meta"][3], inertia=q["Quad2D/meta"][4], radius=q["Quad2D/meta"][2])
    for X, U, st in zip(q["Quad2D/X"], q["Quad2D/U"], q["Quad2D/step"]):
        xn = G.q2_F(X, U, spec, DT)
        xn[2] = ((xn[2] + np.pi) % (2 * np.pi)) - np.pi                         # step() wraps the pitch; h does not see it
        np.testing.assert_allclose(xn, st, rtol=0, atol=1e-12)


def draw(mdl, rng, K=3):
    nx = mdl["nx"]
    x0 = np.zeros(nx); x0[:2] = rng.uniform(2, 10, 2)
    if mdl["name"] == "DoubleIntegrator2D":
        x0[2:4] = rng.uniform(-0.75, 0.75, 2)
    else:
        x0[2] = rng.uniform(-0.2, 0.2); x0[3:5] = rng.uniform(-0.5, 0.5, 2)
    goal = rng.uniform(2, 10, 2)
    obs = np.zeros((K, 7))
    for j in range(K):
        r = rng.uniform(0.2, 0.8); rho = rng.uniform(r + 0.6, 3.0); ph = rng.uniform(-np.pi, np.pi)
        obs[j, :3] = [x0[0] + rho * np.cos(ph), x0[1] + rho * np.sin(ph), r]
    return x0, goal, obs


@pytest.mark.parametrize("mk", [G.di_model, G.quad2d_model])
def test_first_derivatives_against_finite_differences(mk):
    mdl = mk()
    rng = np.random.default_rng(2)
    N = 5
    P = G.params(mdl, N)
    x0, goal, obs = draw(mdl, rng)
    if mdl["name"] == "DoubleIntegrator2D":
        x0[2:4] = [0.7, 0.75]                                                   # |v| > v_max after a step: the rescaled branch
        obs[1] = [x0[0] + 1.6, x0[1] + 1.2, 0.5, 0.7, 4.0, 0.3, 1.0]            # a superellipsoid
    lo, hi = mdl["u_lo"], mdl["u_hi"]
    z = np.tile((lo + hi) / 2, N) + rng.normal(size=2 * N) * 0.1 * (hi - lo).mean()
    up = (lo + hi) / 2
    ev = G.evaluate(x0, z, up, goal, obs, P, None, 1)
    eps = 1e-6
    gfd = np.zeros(2 * N); Jfd = np.zeros_like(ev["J"])
    for i in range(2 * N):
        d = np.zeros(2 * N); d[i] = eps
        a, b = G.evaluate(x0, z + d, up, goal, obs, P, level=0), G.evaluate(x0, z - d, up, goal, obs, P, level=0)
        gfd[i] = (a["f"] - b["f"]) / (2 * eps); Jfd[:, i] = (a["g"] - b["g"]) / (2 * eps)
    assert np.abs(gfd - ev["grad"]).max() <= 1e-6 * max(1.0, np.abs(gfd).max())
    assert np.abs(Jfd - ev["J"]).max() <= 1e-7
    W = G.evaluate(x0, z, up, goal, obs, P, np.zeros(ev["g"].shape[0]), 2)["W"]
    assert np.all(np.linalg.eigvalsh(W) > 0)                                    # Gauss-Newton cost Hessian: positive definite


@pytest.mark.parametrize("mk", [G.di_model, G.quad2d_model])
def test_solver_reaches_a_local_optimum_slsqp_cannot_improve(mk):
    from scipy.optimize import minimize
    mdl = mk()
    rng = np.random.default_rng(13)
    n_ok = 0
    for t in range(5):
        x0, goal, obs = draw(mdl, rng)
        N = 8
        P = G.params(mdl, N)
        up = (mdl["u_lo"] + mdl["u_hi"]) / 2 if mdl["name"] == "Quad2D" else np.zeros(2)
        u0, st, it, info = G.solve(mdl, x0, up, goal, obs, N=N, return_info=True)
        if st != M.STATUS_OPTIMAL:
            continue
        assert np.min(info["g"]) >= -1e-6
        fun = lambda z: G.evaluate(x0, z, up, goal, obs, P, level=1)
        r = minimize(lambda z: fun(z)["f"], info["z"], jac=lambda z: fun(z)["grad"],
                     constraints=[{"type": "ineq", "fun": lambda z: fun(z)["g"], "jac": lambda z: fun(z)["J"]}],
                     method="SLSQP", options={"ftol": 1e-13, "maxiter": 100})
        assert r.fun >= info["f"] * (1 - 1e-6) - 1e-6
        n_ok += 1
    assert n_ok >= 3


@pytest.mark.parametrize("mk", [G.kb_model, G.di_model, G.quad2d_model])
def test_exact_hessian_against_finite_differences_of_the_lagrangian_gradient(mk):
    """Costate-weighted second derivatives of the dynamics and of step o step (oracle/mpc_gn.py: evaluate, exact_hessian):
    W = d/dz (grad f - J' lam) at fixed multipliers, for the served models and for the bicycle kept for later."""
    mdl = mk()
    rng = np.random.default_rng(0)
    N = 5
    P = G.params(mdl, N, exact_hessian=True)
    nx = mdl["nx"]
    x0 = np.zeros(nx); x0[:2] = [1.0, 1.0]
    if mdl["name"] == "KinematicBicycle2D":
        x0[2], x0[3] = 0.4, 1.5
    elif mdl["name"] == "DoubleIntegrator2D":
        x0[2:4] = [0.7, 0.75]                                                   # speed rescaling active
    else:
        x0[2] = 0.1; x0[3:5] = [0.5, 0.2]
    goal = np.array([4.0, 3.0]); obs = np.array([[2.0, 1.8, 0.4, 0, 0, 0, 0], [3.0, 3.0, 0.5, 0, 0, 0, 0]])
    lo, hi = mdl["u_lo"], mdl["u_hi"]
    z = np.tile((lo + hi) / 2, N) + rng.normal(size=2 * N) * 0.1 * (hi - lo).mean()
    up = (lo + hi) / 2
    m = G.evaluate(x0, z, up, goal, obs, P, level=0)["g"].shape[0]
    lam = rng.uniform(0, 1, m)
    W = G.evaluate(x0, z, up, goal, obs, P, lam, 2)["W"]

    def gl(zz):
        e = G.evaluate(x0, zz, up, goal, obs, P, level=1)
        return e["grad"] - e["J"].T @ lam
    eps = 1e-6
    Wfd = np.zeros_like(W)
    for i in range(2 * N):
        d = np.zeros(2 * N); d[i] = eps
        Wfd[:, i] = (gl(z + d) - gl(z - d)) / (2 * eps)
    assert np.abs(Wfd - W).max() <= 1e-7 * np.abs(Wfd).max()
    assert np.abs(W - W.T).max() <= 1e-12 * np.abs(W).max()
