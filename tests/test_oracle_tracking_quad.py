"""CPU: oracle/tracking_quad.py against tests/golden/closed_loop_quads.npz -- the reference's own Quad2D / Quad3D robot
functions and LocalTrackingController.control_step (tests/golden/make_golden_quads.py; the position controller's NLP solve
inside those runs is this repo's oracle, everything around it is the reference's code)."""
import os

import numpy as np
import pytest

from oracle import tracking_quad as T

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "closed_loop_quads.npz"))
Q2 = dict(T.default_spec("Quad2D"), f_min=3.0, f_max=10.0, radius=0.25)
Q3 = T.default_spec("Quad3D")
NAMES = ["idle", "track", "stop", "rotate"]


def test_quad2d_functions():
    X, goal, U = G["q2/X"], G["q2/goal"], G["q2/U"]
    for i in range(len(X)):
        assert np.abs(T.q2_nominal(X[i], goal[i], Q2) - G["q2/nominal"][i]).max() <= 1e-12
        assert np.abs(T.q2_stop(X[i], Q2) - G["q2/stop"][i]).max() <= 1e-12
        assert T.q2_has_stopped(X[i]) == bool(G["q2/has_stopped"][i])
        assert np.abs(T.q2_step(X[i], U[i], 0.05, Q2) - G["q2/step"][i]).max() <= 1e-12
    assert G["q2/has_stopped"].any() and not G["q2/has_stopped"].all()


def test_quad3d_functions():
    X, goal, U, ang = G["q3/X"], G["q3/goal"], G["q3/U"], G["q3/ang"]
    assert np.abs(np.linalg.pinv(T.q3_matrices(Q3)[2]) - G["q3/pinvB2"]).max() <= 1e-14
    for i in range(len(X)):
        assert np.abs(T.q3_nominal(X[i], goal[i], Q3) - G["q3/nominal"][i]).max() <= 1e-11
        assert np.abs(T.q3_stop(X[i], Q3) - G["q3/stop"][i]).max() <= 1e-11
        assert np.abs(T.q3_rotate_to(X[i], ang[i], Q3) - G["q3/rotate_to"][i]).max() <= 1e-11
        assert T.q3_has_stopped(X[i]) == bool(G["q3/has_stopped"][i])
        assert np.abs(T.q3_step(X[i], U[i], 0.05, Q3) - G["q3/step"][i]).max() <= 1e-12
    assert G["q3/has_stopped"].any() and not G["q3/has_stopped"].all()


@pytest.mark.parametrize("tag,model,steps", [("q2_example", "Quad2D", 60), ("q2_behind", "Quad2D", 60), ("q3_example", "Quad3D", 80),
                                             ("q3_behind", "Quad3D", 120)])
def test_closed_loop(tag, model, steps):
    spec = Q2 if model == "Quad2D" else Q3
    ctl = T.QuadTrackingOracle(model, G[f"{tag}/x0"], spec, obs=G[f"{tag}/obs"])
    ctl.set_waypoints(G[f"{tag}/waypoints"])
    assert np.abs(ctl.waypoints - G[f"{tag}/filtered_waypoints"]).max() <= 1e-15
    assert NAMES.index(ctl.state_machine) == int(G[f"{tag}/sm"][0])
    assert np.abs(ctl.X - G[f"{tag}/X"][0]).max() == 0.0
    # the NLP solves inside the golden run are replayed from the fixture where the inputs agree (keeps this test fast);
    # the oracle solver itself is covered by tests/test_oracle_mpc_gn.py / test_oracle_mpc_lin.py
    mx, mu = G[f"{tag}/mpc_x"], G[f"{tag}/mpc_u"]
    state = {"k": 0, "replayed": 0}

    def solve_fn(X, u_prev, goal, obs):
        k = state["k"]
        assert k < len(mx)
        assert np.abs(X - mx[k]).max() <= 1e-8 and np.abs(u_prev - G[f"{tag}/mpc_u_prev"][k]).max() <= 1e-8
        assert np.abs(np.asarray(goal) - G[f"{tag}/mpc_goal"][k]).max() <= 1e-12
        assert np.abs(obs - G[f"{tag}/mpc_obs"][k]).max() <= 1e-12
        state["k"] += 1
        return mu[k]

    ctl.solve_fn = solve_fn
    seen = {ctl.state_machine}
    for k in range(min(steps, len(G[f"{tag}/ret"]))):
        ret = ctl.control_step()
        assert ret == int(G[f"{tag}/ret"][k])
        if ret == -2:                                         # the reference returns before stepping (tracking.py:627-634)
            break
        assert NAMES.index(ctl.state_machine) == int(G[f"{tag}/sm"][k + 1])
        assert ctl.current_goal_index == int(G[f"{tag}/goal_index"][k + 1])
        assert np.abs(ctl.X - G[f"{tag}/X"][k + 1]).max() <= 1e-9
        assert np.abs(ctl.u_pos - G[f"{tag}/U"][k]).max() <= 1e-9
        seen.add(ctl.state_machine)
    if tag == "q3_behind":
        assert {"stop", "rotate", "track"} <= seen


def test_oracle_solver_reproduces_a_few_recorded_solves():
    from oracle import mpc_gn as OG, mpc_lin as OL
    for tag, model in (("q2_example", "Quad2D"), ("q3_example", "Quad3D")):
        for k in (0, 7):
            x, up, goal, obs = (G[f"{tag}/mpc_{n}"][k] for n in ("x", "u_prev", "goal", "obs"))
            if model == "Quad2D":
                u = OG.solve(OG.quad2d_model(dict(Q2), dt=0.05), x, up, goal[:2], obs, N=10)[0]
            else:
                u = OL.solve(OL.quad3d_model(dict(Q3), dt=0.05), x, up, goal[:3], obs, N=10)[0]
            assert np.abs(u - G[f"{tag}/mpc_u"][k]).max() <= 1e-9


def test_vtol_loop_restatement():
    """VTOL2D around a stub controller: X0 padding (5 m/s cruise), 'rotate' skipped, the 1.2 pi cone about the pitch angle with its
    nearest-of-all fallback, zero reference input outside 'track', the ground test."""
    from oracle.tracking_quad import QuadTrackingOracle
    obs = np.array([[30.0, 10.0, 1.0, 0, 0, 0, 0], [-3.0, 10.0, 0.5, 0, 0, 0, 0], [10.0, 40.0, 1.0, 0, 0, 0, 0], [50.0, 9.0, 1.0, 0, 0, 0, 0]])
    seen = []

    def stub(X, up, goal, ob):
        seen.append(ob.copy())
        return np.array([0.6, 0.6, 0.3, 0.0])
    o = QuadTrackingOracle("VTOL2D", [0.0, 10.0], obs=obs, num_constraints=3, solve_fn=stub)
    assert np.array_equal(o.X, [0.0, 10.0, 0.0, 5.0, 0.0, 0.0]) and o.N == 30
    o.set_waypoints(np.array([[0.0, 10.0], [100.0, 10.0]]))
    assert o.state_machine == "track"
    assert o.control_step() == 0
    # in the cone: (30, 10), (50, 9) and (10, 40) (72 degrees up); (-3, 10) is behind -> never handed over while something is ahead
    assert np.array_equal(seen[0][:, 0], [30.0, 10.0, 50.0])
    o2 = QuadTrackingOracle("VTOL2D", [0.0, 10.0], obs=obs[1:2], num_constraints=3, solve_fn=stub)
    o2.set_waypoints(np.array([[0.0, 10.0], [100.0, 10.0]]))
    o2.control_step()
    assert seen[-1][0, 0] == -3.0 and seen[-1][1, 0] == 1000.0            # empty cone: nearest of all, padded with the far dummy
    low = QuadTrackingOracle("VTOL2D", [0.0, 0.2, 0.0, 10.0, -5.0, 0.0], obs=obs, num_constraints=3, solve_fn=lambda *a: np.zeros(4))
    low.set_waypoints(np.array([[0.0, 0.2], [100.0, 0.2]]))
    assert [low.control_step() for _ in range(2)][-1] == -2 and low.X[1] < 0
