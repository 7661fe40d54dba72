"""CPU: the Manipulator2D oracle (oracle/manipulator.py, oracle/qp.py: solve_qpn) against vectors recorded from the
reference's own Manipulator2D class and CBFQP row loop (tests/golden/make_golden.py: gen_manipulator)."""
import os

import numpy as np
import pytest

from oracle import manipulator as M
from oracle import qp as Q

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "manipulator2d.npz"))
W_MAX, KP, RADIUS, BX, BY, NR, DT = G["meta"]
NR = int(NR)
BASE = (BX, BY)
SPEC = dict(w_max=W_MAX, Kp=KP, radius=RADIUS)


def cases(mode):
    n = G[f"{mode}/X"].shape[0]
    for i in range(n):
        k = int(G[f"{mode}/k"][i])
        yield i, G[f"{mode}/X"][i], G[f"{mode}/obs"][i][:k]


def test_kinematics_match_reference():
    for i, X, obs in cases("cbf"):
        np.testing.assert_allclose(M.end_effector(X, BASE), G["cbf/ee"][i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(M.jacobian(X), G["cbf/jac"][i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(M.step(X, G["cbf/U"][i], DT), G["cbf/step"][i], rtol=0, atol=1e-15)
        np.testing.assert_allclose(M.nominal_input(X, G["cbf/goal"][i], SPEC, BASE), G["cbf/nominal"][i], rtol=0, atol=1e-12)


def test_link_circles_match_reference():
    assert M.link_steps() == [8, 8, 6]                    # float64 ceil of 80/60 / (10/60) etc. (25 circles)
    for i, X, obs in cases("cbf"):
        c = M.link_circles(X, BASE)
        ref = G["cbf/circles"][i]
        assert len(c) == ref.shape[0] == 25
        np.testing.assert_allclose(np.array([p for p, _ in c]), ref[:, :2], rtol=0, atol=1e-12)
        assert [li for _, li in c] == list(ref[:, 2].astype(int))


def test_barrier_matches_reference():
    for i, X, obs in cases("cbf"):
        hs, dhs = M.agent_barrier(X, obs[0], RADIUS, base=BASE)
        np.testing.assert_allclose(hs, G["cbf/h0"][i], rtol=0, atol=1e-11)
        np.testing.assert_allclose(np.array(dhs), G["cbf/dh0"][i], rtol=0, atol=1e-11)


@pytest.mark.parametrize("mode", ["cbf", "hard"])
def test_rows_match_reference_row_loop(mode):
    for i, X, obs in cases(mode):
        A, b, hv = M.assemble_rows(X, list(obs), SPEC, alpha=1.0, num_rows=NR, dt=DT, cbf_mode=mode, base=BASE)
        np.testing.assert_allclose(A, G[f"{mode}/A"][i], rtol=0, atol=1e-10)
        np.testing.assert_allclose(b, G[f"{mode}/b"][i], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("mode", ["cbf", "hard"])
def test_solution_recorded_with_the_rows(mode):
    n_inf = 0
    for i, X, obs in cases(mode):
        r = M.solve(X, G[f"{mode}/u_ref"][i], list(obs), SPEC, 1.0, NR, DT, mode, BASE)
        assert r["status"] == G[f"{mode}/status_oracle"][i]
        if r["status"] == 0:
            np.testing.assert_allclose(r["u"], G[f"{mode}/u_star_oracle"][i], rtol=0, atol=1e-9)
        else:
            n_inf += 1
    assert n_inf > 0


def test_qpn_against_full_enumeration_and_slsqp():
    """solve_qpn (constraint generation) = brute-force enumeration over ALL rows on small problems, = SLSQP."""
    from scipy.optimize import minimize
    rng = np.random.default_rng(5)
    n_opt = 0
    for t in range(60):
        m = int(rng.integers(3, 12))
        Gm = rng.normal(size=(m, 3)); c = rng.normal(size=m) + 0.8
        Gm = np.vstack([Gm, np.eye(3), -np.eye(3)]); c = np.concatenate([c, np.full(6, 2.0)])
        ur = rng.uniform(-3, 3, 3)
        u, st = Q.solve_qpn(Gm, c, ur)
        ub = Q._enumerate_qpn(Gm, c, ur, Q.FEAS_TOL)
        assert (ub is None) == (st == Q.STATUS_INFEASIBLE)
        if st == 0:
            n_opt += 1
            np.testing.assert_allclose(u, ub, rtol=0, atol=1e-9)
            res = minimize(lambda x: ((x - ur) ** 2).sum(), np.zeros(3), jac=lambda x: 2 * (x - ur),
                           constraints=[{"type": "ineq", "fun": lambda x: Gm @ x + c, "jac": lambda x: Gm}],
                           method="SLSQP", options={"ftol": 1e-14, "maxiter": 300})
            if res.success:
                assert np.abs(res.x - u).max() < 1e-5
    assert n_opt > 20


def test_no_obstacle_returns_u_ref_unclipped():
    r = M.solve(np.zeros(3), [5.0, -7.0, 1.0], None, SPEC)
    assert r["status"] == 0 and np.all(r["u"] == [5.0, -7.0, 1.0])


@pytest.mark.parametrize("tag", ["example", "sweep", "behind"])
def test_arm_closed_loop_matches_reference(tag):
    """LocalTrackingController with Manipulator2D (tests/golden/make_golden.py: gen_closed_loop_manipulator): every joint
    state, return code and state-machine state of the reference's run to the last waypoint."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "closed_loop_manipulator.npz"))
    names = ["idle", "track", "stop", "rotate"]
    o = M.ArmTrackingOracle(g[f"{tag}/q0"], dict(w_max=2.0, Kp=5.0, radius=0.25, reached_threshold=0.5), base=g["base"], obs=g[f"{tag}/obs"])
    o.set_waypoints(g[f"{tag}/waypoints"])
    np.testing.assert_allclose(o.waypoints, g[f"{tag}/filtered_waypoints"][:, :2], rtol=0, atol=1e-12)
    assert names.index(o.state_machine) == g[f"{tag}/sm"][0]
    Xr, rets = g[f"{tag}/X"], g[f"{tag}/ret"]
    for i in range(len(rets)):
        assert o.control_step() == rets[i]
        assert names.index(o.state_machine) == g[f"{tag}/sm"][i + 1]
        np.testing.assert_allclose(o.X, Xr[i + 1], rtol=0, atol=1e-10)
    assert rets[-1] == -1
