"""CPU: oracle/mpc_lin.py (MPC-CBF problem functions of the reference's linear models, SingleIntegrator2D and Quad3D).
Model matrices against the reference's own f / g / step (tests/golden/linear_models.npz, integrators.npz); derivatives
against finite differences; the solver's optimum against scipy SLSQP."""
import os

import numpy as np
import pytest

from oracle import mpc_cbf as M
from oracle import mpc_lin as L

GD = os.path.join(os.path.dirname(__file__), "golden")
DT = 0.05


def test_quad3d_matrices_match_reference():
    G = np.load(os.path.join(GD, "linear_models.npz"))
    mdl = L.quad3d_model(dt=DT)
    A, B = L.quad3d_matrices(dict(mass=3.0, Ix=0.5, Iy=0.5, Iz=0.5, L=0.3, nu=0.1))
    for X, U, f, g, st in zip(G["Quad3D/X"], G["Quad3D/U"], G["Quad3D/f"], G["Quad3D/g"], G["Quad3D/step"]):
        np.testing.assert_allclose(A @ X, f, rtol=0, atol=1e-13)
        np.testing.assert_allclose(B, g, rtol=0, atol=1e-15)
        np.testing.assert_allclose(mdl["Ae"] @ X + mdl["Be"] @ U, X + (f + g @ U) * DT, rtol=0, atol=1e-12)   # mpc_cbf.py:138
        np.testing.assert_allclose(mdl["As"] @ X + mdl["Bs"] @ U, st, rtol=0, atol=1e-12)                      # RK4 step (angles unwrapped)
    assert tuple(G["Quad3D/spec"][6:8]) == (10.0, -10.0)


def test_single_integrator_matrices_match_reference():
    G = np.load(os.path.join(GD, "integrators.npz"))
    mdl = L.si_model(dt=DT)
    X, U, st = G["SingleIntegrator2D/X"], G["SingleIntegrator2D/U"], G["SingleIntegrator2D/step"]
    for i in range(X.shape[0]):
        np.testing.assert_allclose(mdl["As"] @ X[i, :2] + mdl["Bs"] @ U[i], st[i, :2], rtol=0, atol=1e-13)


def draw(mdl, rng, K=4, moving=True):
    nx, nu = mdl["nx"], mdl["nu"]
    x0 = np.zeros(nx); x0[:2] = rng.uniform(0, 8, 2)
    if nx == 12:
        x0[2] = rng.uniform(0.5, 2.0); x0[6:8] = rng.uniform(-1.5, 1.5, 2) if moving else 0.0
        x0[3:5] = rng.uniform(-0.1, 0.1, 2)
    goal = np.concatenate([rng.uniform(0, 8, 2), [1.0]])[: mdl["ng"]]
    obs = np.zeros((K, 7))
    for j in range(K):
        rho, ph = rng.uniform(0.9, 3.0), rng.uniform(-np.pi, np.pi)
        obs[j, :3] = [x0[0] + rho * np.cos(ph), x0[1] + rho * np.sin(ph), rng.uniform(0.2, 0.6)]
    if nx == 12 and moving:                                   # one obstacle dead ahead, so that a CBF row binds inside the horizon
        v = x0[6:8]; d = v / max(np.linalg.norm(v), 1e-9)
        obs[0, :3] = [x0[0] + 1.0 * d[0], x0[1] + 1.0 * d[1], 0.4]
        goal[:2] = x0[:2] + 4.0 * d
    return x0, goal, obs


@pytest.mark.parametrize("name", ["si", "quad3d"])
def test_derivatives_against_finite_differences(name):
    mdl = L.si_model() if name == "si" else L.quad3d_model()
    rng = np.random.default_rng(3)
    N = 5
    P = L.params(mdl, N)
    n = N * mdl["nu"]
    x0, goal, obs = draw(mdl, rng, K=3)
    if name == "si":
        obs[1] = [x0[0] + 1.5, x0[1] - 0.5, 0.5, 0.7, 4.0, 0.3, 1.0]           # a superellipsoid (SI accepts them)
    z = rng.normal(size=n) * 0.3
    up = rng.normal(size=mdl["nu"]) * 0.1
    lam = rng.uniform(0, 1, N * 3 + 2 * n)
    ev = L.evaluate(x0, z, up, goal, obs, P, lam, 2)
    eps = 1e-6

    def gl(zz):
        e = L.evaluate(x0, zz, up, goal, obs, P, level=1)
        return e["grad"] - e["J"].T @ lam
    gfd = np.zeros(n); Jfd = np.zeros_like(ev["J"]); Wfd = np.zeros((n, n))
    for i in range(n):
        d = np.zeros(n); d[i] = eps
        a, b = L.evaluate(x0, z + d, up, goal, obs, P, level=0), L.evaluate(x0, z - d, up, goal, obs, P, level=0)
        gfd[i] = (a["f"] - b["f"]) / (2 * eps); Jfd[:, i] = (a["g"] - b["g"]) / (2 * eps)
        Wfd[:, i] = (gl(z + d) - gl(z - d)) / (2 * eps)
    assert np.abs(gfd - ev["grad"]).max() <= 1e-6 * max(1.0, np.abs(gfd).max())
    assert np.abs(Jfd - ev["J"]).max() <= 1e-7
    assert np.abs(Wfd - ev["W"]).max() <= 1e-6 * max(1.0, np.abs(Wfd).max())
    Hc, G, _ = L.condensed(mdl, N)
    np.testing.assert_allclose(L.evaluate(x0, z, up, goal, obs, P, np.zeros_like(lam), 2)["W"], Hc, rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", ["si", "quad3d"])
def test_solver_reaches_a_local_optimum_slsqp_cannot_improve(name):
    from scipy.optimize import minimize
    mdl = L.si_model() if name == "si" else L.quad3d_model()
    rng = np.random.default_rng(11)
    n_act = 0
    for t in range(6):
        x0, goal, obs = draw(mdl, rng)
        N = 8
        P = L.params(mdl, N)
        u0, st, it, info = L.solve(mdl, x0, np.zeros(mdl["nu"]), goal, obs, N=N, return_info=True)
        if st != M.STATUS_OPTIMAL:
            continue
        assert np.min(info["g"]) >= -1e-6
        n_act += int(np.min(info["g"][: N * 4]) < 1e-4)
        fun = lambda z: L.evaluate(x0, z, np.zeros(mdl["nu"]), goal, obs, P, level=1)
        r = minimize(lambda z: fun(z)["f"], info["z"], jac=lambda z: fun(z)["grad"],
                     constraints=[{"type": "ineq", "fun": lambda z: fun(z)["g"], "jac": lambda z: fun(z)["J"]}],
                     method="SLSQP", options={"ftol": 1e-13, "maxiter": 100})
        assert r.fun >= info["f"] * (1 - 1e-6) - 1e-6
    assert n_act >= 1


def test_library_host_side_condensed_matrices_match_oracle():
    """sc_mpclin_build_model (host code of the HIP library, no GPU call) against oracle.mpc_lin.condensed."""
    import safe_control_amd as sca
    from safe_control_amd import _lib
    from safe_control_amd.position_control import mpc_cbf_linear as ML
    from safe_control_amd.robots.linear_models import linear_model
    lib = _lib.load()
    for name, om in (("SingleIntegrator2D", L.si_model()), ("Quad3D", L.quad3d_model())):
        spec = sca.complete_robot_spec({"model": name})
        mdl = linear_model(spec, DT)
        for key in ("Ae", "Be", "As", "Bs"):
            np.testing.assert_allclose(mdl[key], om[key], rtol=0, atol=1e-15)
        for N in (10, 6):
            p = ML.make_params(mdl, mdl["cbf_param"], N, 0.25, _lib.DTYPE_F64)
            blob = ML.build_model_blob(lib, p, mdl)
            Hc, G, _ = L.condensed(om, N)
            nx, nu = mdl["nx"], mdl["nu"]
            n = N * nu
            off = nx * nx + nx * nu + 2 * nx + 2 * nu
            np.testing.assert_allclose(blob[off:off + n * n].reshape(n, n), Hc, rtol=1e-13, atol=1e-12)
            np.testing.assert_allclose(blob[off + n * n:].reshape(4 * N, n), G, rtol=0, atol=1e-15)
    assert lib.sc_mpclin_model_doubles(12, 4, 33) == 0            # nu * horizon > 128
