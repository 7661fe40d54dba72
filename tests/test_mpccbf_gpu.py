"""GPU tests for the MPC-CBF kernel (pytest -m gpu).  Everything goes through the C-ABI.

The kernel mirrors oracle/mpc_cbf.py operation for operation in f64, so on identical inputs the
two follow the same iterates: same status, (almost always) the same iteration count, and
|u0 - u0_oracle| <= 1e-6, |z - z_oracle| <= 1e-5.  Independently of the oracle's solver, every
point reported optimal must be feasible to 1e-6 and must not be improvable by SLSQP by more than
1e-6 relative cost (local optimality of the restated NLP).  Parity with IPOPT itself is UNPINNED
(see oracle/mpc_cbf.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import mpc_cbf as M  # noqa: E402
import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402

DEV = "cuda:0"
SPEC = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
P = dict(M.DEFAULTS)


def run_gpu(X, up, goal, obs, io="f64", horizon=10):
    ctl = sca.BatchedMPCCBF(dict(SPEC), io_dtype=io, horizon=horizon)
    td = ctl.torch_dtype
    t = lambda a: torch.tensor(a, dtype=td, device=DEV)
    tX, tu, tg, to = t(X), t(up), t(goal), t(obs)
    u, st, it, z = ctl.solve(tX, tu, tg, to, want_z=True)
    torch.cuda.synchronize()
    seen = tuple(a.double().cpu().numpy() for a in (tX, tu, tg, to))
    return u.double().cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.double().cpu().numpy(), seen


@pytest.mark.parametrize("io", ["f64", "f32"])
def test_config3_sample_against_oracle(io):
    """BASELINE config 3 draws (DU, N = 10, K = 8, u_prev = 0); first 192 agents against the numpy oracle."""
    B = 192
    X, goal, ur, obs = W.du_cbfqp_batch(B, 8, seed=0)
    up = np.zeros((B, 2))
    u, st, it, z, (Xs, us, gs, os_) = run_gpu(X, up, goal, obs, io)
    n_same_it = 0
    for i in range(B):
        uo, so, ito, info = M.solve(Xs[i], us[i], gs[i], os_[i], return_info=True)
        assert st[i] == so, (i, st[i], so)
        if so == M.STATUS_OPTIMAL:
            tol = 1e-6 if io == "f64" else 5e-6
            assert np.abs(u[i] - uo).max() <= tol, (i, u[i], uo)
            assert np.abs(z[i] - info["z"]).max() <= 20 * tol
            assert abs(it[i] - ito) <= 2
        n_same_it += int(it[i] == ito)
    assert n_same_it >= 0.9 * B
    assert (st == 0).mean() > 0.8 and (st == 1).sum() > 0


def test_config3_full_batch_against_oracle():
    """ALL 4096 problems of BASELINE config 3 (not a sample): the numpy oracle runs on the host cores in child processes
    (tests/_oracle_pool.py); same status everywhere, |u0 - u0_oracle| <= 1e-6 and |z - z_oracle| <= 2e-5 on every optimal
    problem, iteration counts within 2."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from _oracle_pool import mpc_cbf_solve_many
    B = 4096
    X, goal, ur, obs = W.du_cbfqp_batch(B, 8, seed=0)
    up = np.zeros((B, 2))
    u, st, it, z, (Xs, us, gs, os_) = run_gpu(X, up, goal, obs, "f64")
    uo, so, ito, zo, fo = mpc_cbf_solve_many(Xs, us, gs, os_)
    assert np.array_equal(st, so), np.flatnonzero(st != so)[:10]
    ok = so == M.STATUS_OPTIMAL
    assert ok.mean() > 0.85
    assert np.abs(u[ok] - uo[ok]).max() <= 1e-6
    assert np.abs(z[ok] - zo[ok]).max() <= 2e-5
    assert np.abs(it[ok] - ito[ok]).max() <= 2
    assert (it == ito).mean() >= 0.95


def test_feasibility_and_local_optimality_independent_of_oracle_solver():
    from scipy.optimize import minimize
    B = 64
    X, goal, ur, obs = W.du_cbfqp_batch(B, 8, seed=3)
    rng = np.random.default_rng(3)
    up = rng.uniform(-1, 1, (B, 2)) * np.array([1.0, 0.5])
    u, st, it, z, _ = run_gpu(X, up, goal, obs, "f64")
    checked = 0
    for i in range(B):
        if st[i] != 0:
            continue
        ev = M.evaluate(X[i], z[i], up[i], goal[i], obs[i], P, level=0)
        assert ev["g"].min() >= -1e-6
        if checked < 10:
            fun = lambda zz: M.evaluate(X[i], zz, up[i], goal[i], obs[i], P, level=0)["f"]
            con = lambda zz: M.evaluate(X[i], zz, up[i], goal[i], obs[i], P, level=0)["g"]
            jac = lambda zz: M.evaluate(X[i], zz, up[i], goal[i], obs[i], P, level=1)["grad"]
            cjac = lambda zz: M.evaluate(X[i], zz, up[i], goal[i], obs[i], P, level=1)["J"]
            r = minimize(fun, z[i], jac=jac, constraints=[{"type": "ineq", "fun": con, "jac": cjac}],
                         method="SLSQP", options={"ftol": 1e-13, "maxiter": 100})
            assert r.fun >= ev["f"] * (1 - 1e-6) - 1e-6      # SLSQP cannot improve the reported optimum
            checked += 1
    assert checked == 10


@pytest.mark.parametrize("N,K", [(5, 3), (10, 1), (20, 8), (12, 16), (15, 8), (30, 4)])   # 30: order 60, partial last tile row of the LDS Cholesky
def test_other_horizons_and_obstacle_counts(N, K):
    B = 24
    X, goal, ur, obs = W.du_cbfqp_batch(B, K, seed=N * 100 + K)
    up = np.zeros((B, 2))
    u, st, it, z, _ = run_gpu(X, up, goal, obs, "f64", horizon=N)
    for i in range(0, B, 3):
        uo, so, ito = M.solve(X[i], up[i], goal[i], obs[i], params={"N": N})
        assert st[i] == so
        if so == 0:
            assert np.abs(u[i] - uo).max() <= 1e-6


def test_superellipsoid_and_dummy_obstacles():
    B = 32
    X, goal, ur, obs = W.du_cbfqp_batch(B, 6, seed=8)
    rng = np.random.default_rng(8)
    for i in range(B):
        obs[i, 4] = M.DUMMY_OBS                                     # update_tvp padding row
        a, b = rng.uniform(0.4, 1.0, 2)
        rho, phi = rng.uniform(1.8, 3.5), rng.uniform(-np.pi, np.pi)
        obs[i, 5] = [X[i, 0] + rho * np.cos(phi), X[i, 1] + rho * np.sin(phi), a, b, float(rng.choice([2, 4, 6])),
                     rng.uniform(-3, 3), 1.0]
    up = np.zeros((B, 2))
    u, st, it, z, _ = run_gpu(X, up, goal, obs, "f64")
    for i in range(B):
        uo, so, ito = M.solve(X[i], up[i], goal[i], obs[i])
        assert st[i] == so
        if so == 0:
            assert np.abs(u[i] - uo).max() <= 2e-6


def test_full_batch_4096_is_deterministic_and_position_independent():
    B = 4096
    X, goal, ur, obs = W.du_cbfqp_batch(B, 8, seed=0)
    ctl = sca.BatchedMPCCBF(dict(SPEC), io_dtype="f32")
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    tX, tu, tg, to = t(X), torch.zeros((B, 2), device=DEV), t(goal), t(obs)
    u1, s1, i1 = ctl.solve(tX, tu, tg, to)
    u2, s2, i2 = ctl.solve(tX, tu, tg, to)
    assert torch.equal(s1, s2) and torch.equal(i1, i2) and torch.equal(u1, u2)
    u3, s3, i3 = ctl.solve(tX[1000:1300].contiguous(), tu[1000:1300].contiguous(), tg[1000:1300].contiguous(),
                           to[1000:1300].contiguous())
    assert torch.equal(s3, s1[1000:1300]) and torch.equal(u3, u1[1000:1300])
    frac_opt = (s1 == 0).double().mean().item()
    assert 0.8 < frac_opt < 1.0
    assert (s1 == 2).double().mean().item() < 0.01                   # hardly any run into the iteration limit
    ok = s1 == 0
    assert (u1[ok, 0].abs() <= 1.0 + 1e-6).all() and (u1[ok, 1].abs() <= 0.5 + 1e-6).all()
    assert i1[ok].double().mean().item() < 30


def test_dropin_class_closed_loop_matches_oracle():
    """MPCCBF through the reference's plugin surface on the CONDENSED kernel (robot_spec['mpc_formulation'] = 'condensed'; the default
    formulation of this robot is the multiple-shooting kernel since round 6: tests/test_mpccbf_ms_gpu.py), 15 closed-loop steps with u_prev feedback."""
    from oracle import robots as R
    spec = dict(SPEC, mpc_formulation="condensed")
    robot = sca.RobotHandle(np.array([2.0, 2.0, np.pi / 2, 1.0]), spec, dt=0.05)
    ctl = sca.MPCCBF(robot, spec, num_obs=8)
    obs = [[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3]]
    goal = np.array([2.0, 12.0])
    Xo = robot.X.reshape(-1).copy(); up = np.zeros(2)
    ospec = R.default_spec(R.MODEL_DU); ospec.update(a_max=1.0, w_max=0.5)
    for k in range(15):
        u = ctl.solve_control_problem(robot.X, {"state_machine": "track", "u_ref": np.zeros((2, 1)), "goal": goal}, obs)
        uo, so, ito = M.solve(Xo, up, goal, M.pad_obstacles(obs, 8))
        assert u.shape == (2, 1) and ctl.status == "optimal" and ctl.solver_status == "optimal" and so == 0
        np.testing.assert_allclose(u.reshape(-1), uo, atol=2e-6)
        Xo = R.step(R.MODEL_DU, Xo, uo, 0.05, ospec); up = uo
        robot.X = R.step(R.MODEL_DU, robot.X.reshape(-1), u.reshape(-1), 0.05, ospec).reshape(-1, 1)
    # not tracking -> u_ref passes through untouched (mpc_cbf.py:379-381)
    ur = np.array([[0.3], [-0.1]])
    assert ctl.solve_control_problem(robot.X, {"state_machine": "stop", "u_ref": ur, "goal": goal}, obs) is ur
    with pytest.raises(ValueError):
        ctl.solve_control_problem(robot.X, {"state_machine": "track", "u_ref": ur, "goal": goal}, [[1.0, 2.0, 0.3, 0.0, 0.0]])


def test_non_finite_inputs_terminate_and_are_not_reported_optimal():
    """NaN / inf in the state, the goal or an obstacle row: every loop of the kernels is bounded (max_iter, 12 halvings,
    40 inertia retries), the launch returns and the affected problems are not 'optimal'; the others are untouched."""
    X, goal, ur, obs = W.du_cbfqp_batch(64, 8, seed=11)
    bad = {3: "X", 10: "goal", 17: "obs", 30: "Xinf"}
    X2, g2, o2 = X.copy(), goal.copy(), obs.copy()
    X2[3, 2] = np.nan; g2[10, 0] = np.nan; o2[17, 4, 1] = np.nan; X2[30, 0] = np.inf
    up = np.zeros((64, 2))
    u, st, it, z, _ = run_gpu(X2, up, g2, o2, "f64")
    u0, st0, it0, z0, _ = run_gpu(X, up, goal, obs, "f64")
    for i in bad:
        assert st[i] != 0, (i, st[i])
    good = [i for i in range(64) if i not in bad]
    assert np.array_equal(st[good], st0[good]) and np.array_equal(u[good], u0[good])
    import safe_control_amd as sca
    dev = torch.device("cuda:0")
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    od = sca.BatchedOptimalDecayMPCCBF(dict(SPEC), io_dtype="f64")
    uo, rho, sto, ito = od.solve(t(X2), t(up), t(g2), t(o2))
    torch.cuda.synchronize()
    sto = sto.cpu().numpy()
    for i in bad:
        assert sto[i] != 0, (i, sto[i])
