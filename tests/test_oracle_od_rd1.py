"""CPU tests of the optimal-decay MPC-CBF oracle for the relative-degree-1 models (oracle/od_mpc_rd1.py): the
BASELINE config 5 extension (Unicycle2D + Quad3D, superellipsoid obstacles).  Parity UNPINNED (no reference
counterpart: see the module header); checked here: derivatives against finite differences, the Schur elimination
against the dense Newton step, SLSQP agreement on the optimum, and the limits that tie the extension to pinned code
(rho frozen at 1 by a huge penalty reproduces the plain MPC-CBF row values)."""
import numpy as np
import pytest

from oracle import mpc_cbf as M
from oracle import mpc_cbf_uni as MU
from oracle import mpc_lin as L
from oracle import od_mpc_rd1 as O


def superellipsoids(rng, p, K):
    obs = np.zeros((K, 7))
    for j in range(K):
        rho, phi = rng.uniform(1.6, 3.5), rng.uniform(-np.pi, np.pi)
        obs[j] = [p[0] + rho * np.cos(phi), p[1] + rho * np.sin(phi), rng.uniform(0.3, 0.8), rng.uniform(0.3, 0.8),
                  float(rng.choice([2, 4, 6])), rng.uniform(-3, 3), 1.0]
    return obs


def uni_case(seed, N=6, K=3):
    rng = np.random.default_rng(seed)
    x0 = np.array([*rng.uniform(2, 12, 2), rng.uniform(-3, 3)])
    goal = x0[:2] + rng.uniform(-4, 4, 2)
    obs = superellipsoids(rng, x0, K)
    obs[0, 3:] = 0.0; obs[0, 2] = 0.4                                        # one circle among them
    return x0, goal, obs, O.uni_params(N=N)


def quad_case(seed, N=5, K=3):
    rng = np.random.default_rng(seed)
    x0 = np.zeros(12); x0[0:3] = [*rng.uniform(2, 12, 2), 1.0]; x0[6:8] = rng.uniform(-0.5, 0.5, 2)
    goal = np.array([*(x0[:2] + rng.uniform(-4, 4, 2)), 1.5])
    obs = superellipsoids(rng, x0, K)
    mdl = dict(L.quad3d_model(), circles_only=False)
    return x0, goal, obs, O.lin_params(mdl, N=N)


CASES = {"uni": uni_case, "quad3d": quad_case}


@pytest.mark.parametrize("kind", ["uni", "quad3d"])
def test_derivatives_by_finite_differences(kind):
    x0, goal, obs, P = CASES[kind](1)
    N, nu = P["N"], P.get("nu", 2)
    n = N * nu
    rng = np.random.default_rng(0)
    lo = P["u_lo"] if "u_lo" in P else -np.array([P["a_max"], P["w_max"]])
    hi = P["u_hi"] if "u_hi" in P else np.array([P["a_max"], P["w_max"]])
    zz = np.concatenate([rng.uniform(np.tile(lo, N) * 0.3, np.tile(hi, N) * 0.3), rng.uniform(0.5, 1.6, N)])
    m = O.evaluate(x0, zz, goal, obs, P, level=0)["g"].shape[0]
    lam = rng.uniform(0, 2, m)
    ev = O.evaluate(x0, zz, goal, obs, P, lam, level=2)
    h = 1e-6
    I = np.eye(n + N)
    f = lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["f"]
    g = lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["g"]
    gfd = np.array([(f(zz + h * e) - f(zz - h * e)) / (2 * h) for e in I])
    Jfd = np.array([(g(zz + h * e) - g(zz - h * e)) / (2 * h) for e in I]).T
    assert np.abs(gfd - ev["grad"]).max() <= 1e-5 * max(1.0, np.abs(gfd).max())
    assert np.abs(Jfd - ev["J"]).max() <= 1e-5 * max(1.0, np.abs(Jfd).max())

    def gL(v):
        e = O.evaluate(x0, v, goal, obs, P, level=1)
        return e["grad"] - e["J"].T @ lam
    Wfd = np.array([(gL(zz + h * e) - gL(zz - h * e)) / (2 * h) for e in I])
    assert np.abs(Wfd - ev["W"]).max() <= 2e-5 * max(1.0, np.abs(ev["W"]).max())
    assert np.abs(ev["W"] - ev["W"].T).max() < 1e-12


@pytest.mark.parametrize("kind", ["uni", "quad3d"])
def test_rows_reduce_to_the_plain_rows_at_rho_one(kind):
    """rho = 1 everywhere: the CBF rows are the pinned MPC-CBF rows of the model (d_h + alpha h_k)."""
    x0, goal, obs, P = CASES[kind](2)
    N, nu = P["N"], P.get("nu", 2)
    rng = np.random.default_rng(1)
    z = rng.uniform(-0.3, 0.3, N * nu)
    zz = np.concatenate([z, np.ones(N)])
    base = (L.evaluate if "model" in P else MU.evaluate)(x0, z, np.zeros(nu), goal, obs, dict(P, rterm="du"), level=0)
    got = O.evaluate(x0, zz, goal, obs, P, level=0)
    K = obs.shape[0]
    np.testing.assert_allclose(got["g"][: N * K], base["g"][: N * K], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind,seeds", [("uni", range(6)), ("quad3d", range(4))])
def test_schur_elimination_equals_dense_newton_and_slsqp_agrees(kind, seeds):
    from scipy.optimize import minimize
    n_opt = 0
    for seed in seeds:
        x0, goal, obs, P = CASES[kind](10 + seed, N=6 if kind == "uni" else 5)
        N, nu = P["N"], P.get("nu", 2)
        n = N * nu
        up = np.zeros(nu)
        u, rho, st, it, info = O.solve(x0, up, goal, obs, P, return_info=True)
        ud, rhod, std, itd, infod = O.solve(x0, up, goal, obs, P, return_info=True, linear_algebra="dense")
        assert st == std
        if st != O.STATUS_OPTIMAL:
            continue
        n_opt += 1
        assert np.abs(info["zz"] - infod["zz"]).max() <= 1e-7
        assert info["g"].min() >= -1e-7
        fun = lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["f"]
        con = lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["g"]
        jac = lambda v: O.evaluate(x0, v, goal, obs, P, level=1)["grad"]
        cjac = lambda v: O.evaluate(x0, v, goal, obs, P, level=1)["J"]
        r = minimize(fun, info["zz"], jac=jac, constraints=[{"type": "ineq", "fun": con, "jac": cjac}], method="SLSQP",
                     options={"ftol": 1e-13, "maxiter": 100})
        assert r.fun >= info["f"] * (1 - 1e-6) - 1e-6                      # SLSQP cannot improve the reported optimum
    assert n_opt >= len(list(seeds)) - 1


def test_decay_variable_buys_a_faster_approach_when_the_row_is_active():
    """A unicycle driving at an obstacle in front of its goal: the plain row h(p_k+1) >= (1 - alpha) h(p_k) caps the approach
    speed; the optimal-decay row lets rho_k rise above 1 (paying p_sb (rho - 1)^2) where that lowers the total cost."""
    P = O.uni_params(N=6)
    x0 = np.array([2.0, 2.0, 0.0])
    goal = np.array([6.0, 2.0])
    obs = np.array([[3.2, 2.0, 0.4, 0, 0, 0, 0.0]])
    u, rho0, st, it, info = O.solve(x0, np.zeros(2), goal, obs, P, return_info=True)
    assert st == O.STATUS_OPTIMAL and rho0 > 1.0 + 1e-3
    K = 1
    assert info["g"][: P["N"] * K].min() < 1e-5                              # CBF rows active at the optimum
    # the same problem with the decay variables pinned at 1 by a huge penalty costs more
    u1, rho1, st1, it1, info1 = O.solve(x0, np.zeros(2), goal, obs, dict(P, p_sb1=1e9), return_info=True)
    assert st1 == O.STATUS_OPTIMAL and abs(rho1 - 1.0) < 1e-6
    assert info["f"] < info1["f"] - 1e-3
