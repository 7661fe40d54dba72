"""CPU tests of the optimal-decay MPC-CBF oracle (oracle/od_mpc_cbf.py).

Parity UNPINNED and the reference copy is stale (oracle header): the checks are self-consistency of the
restated problem (finite differences, the row definition of optimal_decay_mpc_cbf.py:291-297), agreement of the
two linear-algebra paths of the solver, and an independent solver (scipy SLSQP) reaching the same optimum.
"""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import mpc_cbf as M
from oracle import od_mpc_cbf as O
from safe_control_amd import workloads as W

P = dict(O.DEFAULTS)


def case(i, seed=0, superell=False, K=8):
    X, goal, ur, obs = W.du_cbfqp_batch(64, K, seed=seed)
    o = obs[i].copy()
    if superell:
        o[1] = [X[i, 0] + 2.0, X[i, 1] + 0.5, 0.6, 0.9, 4.0, 0.7, 1.0]
    return X[i], goal[i], o


@pytest.mark.parametrize("superell", [False, True])
def test_derivatives_by_finite_differences(superell):
    x0, goal, obs = case(1, superell=superell)
    rng = np.random.default_rng(0)
    zz = np.concatenate([rng.uniform(-0.4, 0.4, 20), rng.uniform(0.2, 3.0, 20)])
    lam = rng.uniform(0, 2, 140)
    ev = O.evaluate(x0, zz, goal, obs, P, lam, level=2)
    h = 1e-6
    f = lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["f"]
    g = lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["g"]
    E = np.eye(40)
    gfd = np.array([(f(zz + h * e) - f(zz - h * e)) / (2 * h) for e in E])
    Jfd = np.array([(g(zz + h * e) - g(zz - h * e)) / (2 * h) for e in E]).T
    assert np.abs(gfd - ev["grad"]).max() <= 1e-5 * np.abs(gfd).max()
    assert np.abs(Jfd - ev["J"]).max() <= 1e-6 * max(1.0, np.abs(ev["J"]).max())

    def gL(v):
        e = O.evaluate(x0, v, goal, obs, P, level=1)
        return e["grad"] - e["J"].T @ lam
    Wfd = np.array([(gL(zz + h * e) - gL(zz - h * e)) / (2 * h) for e in E])
    assert np.abs(Wfd - ev["W"]).max() <= 1e-5 * max(1.0, np.abs(ev["W"]).max())
    assert np.abs(ev["W"] - ev["W"].T).max() < 1e-12


def test_row_equals_reference_definition_and_reduces_to_mpccbf():
    """row = dd_h + (a1 rho1 + a2 rho2) d_h + a1 a2 rho1 rho2 h (optimal_decay_mpc_cbf.py:291-297); with rho = 1 and
    the same gains it is the MPC-CBF row (mpc_cbf.py:316-321)."""
    x0, goal, obs = case(3)
    rng = np.random.default_rng(1)
    z = rng.uniform(-0.4, 0.4, 20)
    rho = rng.uniform(0.3, 2.5, (10, 2))
    Pq = dict(P, alpha1=0.15, alpha2=0.12)
    g = O.evaluate(x0, np.concatenate([z, rho.reshape(-1)]), goal, obs, Pq, level=0)["g"]
    X, pe = M.rollout(x0, z, Pq)
    pos = np.vstack([X[:, :2], pe[None]])
    for k in (0, 4, 9):
        for j in (0, 5):
            h0, h1, h2 = (M.barrier(pos[k + t], obs[j], Pq)[0] for t in range(3))
            d_h, dd_h = h1 - h0, h2 - 2 * h1 + h0
            ref = dd_h + (Pq["alpha1"] * rho[k, 0] + Pq["alpha2"] * rho[k, 1]) * d_h + Pq["alpha1"] * Pq["alpha2"] * rho[k, 0] * rho[k, 1] * h0
            assert abs(g[k * 8 + j] - ref) <= 1e-12 * max(1.0, abs(ref))
    g1 = O.evaluate(x0, np.concatenate([z, np.ones(20)]), goal, obs, Pq, level=0)["g"]
    gm = M.evaluate(x0, z, np.zeros(2), goal, obs, dict(M.DEFAULTS, alpha1=0.15, alpha2=0.12), level=0)["g"]
    assert np.abs(g1 - gm).max() < 1e-12


@pytest.mark.parametrize("i", [0, 3, 5, 9])
def test_schur_and_dense_newton_steps_agree(i):
    x0, goal, obs = case(i)
    a = O.solve(x0, np.zeros(2), goal, obs, return_info=True, linear_algebra="schur")
    b = O.solve(x0, np.zeros(2), goal, obs, return_info=True, linear_algebra="dense")
    assert a[2] == b[2]
    if a[2] == O.STATUS_OPTIMAL:
        assert np.abs(a[4]["zz"] - b[4]["zz"]).max() < 1e-6
        assert abs(a[3] - b[3]) <= 2


@pytest.mark.parametrize("i", [0, 3, 5, 9, 11])
def test_oracle_solution_is_a_local_optimum_slsqp_agrees(i):
    x0, goal, obs = case(i)
    u0, rho0, st, it, info = O.solve(x0, np.zeros(2), goal, obs, return_info=True)
    if st != O.STATUS_OPTIMAL:
        pytest.skip("infeasible start")
    zz = info["zz"]
    assert info["g"].min() >= -1e-7
    f = lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["f"]
    jac = lambda v: O.evaluate(x0, v, goal, obs, P, level=1)["grad"]
    cons = {"type": "ineq", "fun": lambda v: O.evaluate(x0, v, goal, obs, P, level=0)["g"],
            "jac": lambda v: O.evaluate(x0, v, goal, obs, P, level=1)["J"]}
    r = minimize(f, zz, jac=jac, constraints=[cons], method="SLSQP", options=dict(maxiter=200, ftol=1e-12))
    assert r.fun >= info["f"] * (1 - 1e-6) - 1e-9                # SLSQP cannot improve the reported optimum
    z0 = np.concatenate([np.zeros(20), np.ones(20)])
    r2 = minimize(f, z0, jac=jac, constraints=[cons], method="SLSQP", options=dict(maxiter=400, ftol=1e-12))
    if r2.success and abs(r2.fun - info["f"]) <= 1e-6 * max(1.0, abs(info["f"])):
        assert np.abs(r2.x[:2] - u0).max() < 1e-3


def test_decay_rates_move_when_rows_are_active():
    """With the reference's slow default decay (a1 = a2 = 0.01) rows are active in the batch cases and the solver
    trades the decay penalty against them: rho leaves omega = 1, every row still holds."""
    moved = 0
    for i in (0, 3, 5, 9):
        x0, goal, obs = case(i)
        u0, rho0, st, it, info = O.solve(x0, np.zeros(2), goal, obs, return_info=True)
        assert st == O.STATUS_OPTIMAL
        assert info["g"].min() >= -1e-7
        moved += np.abs(info["zz"][20:] - 1.0).max() > 1e-3
    assert moved >= 2
