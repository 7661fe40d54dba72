"""GPU: closed loop of Quad2D / Quad3D (csrc/tracking_quad.hip around csrc/mpc_gn.hip / csrc/mpc_lin.hip) against the
reference-executed goldens tests/golden/closed_loop_quads.npz and the oracle loop (oracle/tracking_quad.py).  The position
controller's solve in the goldens is this repo's numpy oracle, which the MPC kernels follow iterate for iterate; the loop
around it is the reference's own control_step.  Tolerances: states 1e-6 over the compared window (MPC inputs agree to
1e-6 per step and the loop is stable), identical return codes / state machine / goal indices."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402

DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "closed_loop_quads.npz"))
Q2 = {"model": "Quad2D", "f_min": 3.0, "f_max": 10.0, "radius": 0.25}
Q3 = {"model": "Quad3D", "radius": 0.25}


@pytest.mark.parametrize("tag,spec,steps", [("q2_example", Q2, 100), ("q2_behind", Q2, 100), ("q3_example", Q3, 150), ("q3_behind", Q3, 150)])
def test_reference_closed_loops(tag, spec, steps):
    ctl = sca.BatchedTrackingController(G[f"{tag}/x0"][None, :], dict(spec), controller_type={"pos": "mpc_cbf"}, obs=G[f"{tag}/obs"],
                                        io_dtype="f64", device=DEV)
    assert type(ctl).__name__ == "BatchedQuadTrackingController"
    ctl.set_waypoints(G[f"{tag}/waypoints"])
    assert int(ctl.state_machine[0].item()) == int(G[f"{tag}/sm"][0])
    assert np.abs(ctl.X[0].cpu().numpy() - G[f"{tag}/X"][0]).max() == 0.0
    seen = {int(ctl.state_machine[0].item())}
    worst = 0.0
    for k in range(steps):
        ret = ctl.control_step(1)
        assert int(ret[0].item()) == int(G[f"{tag}/ret"][k]), k
        assert int(ctl.state_machine[0].item()) == int(G[f"{tag}/sm"][k + 1]), k
        assert int(ctl.current_goal_index[0].item()) == int(G[f"{tag}/goal_index"][k + 1]), k
        worst = max(worst, np.abs(ctl.X[0].cpu().numpy() - G[f"{tag}/X"][k + 1]).max())
        assert np.abs(ctl.u_pos[0].cpu().numpy() - G[f"{tag}/U"][k]).max() <= 1e-5, k
        seen.add(int(ctl.state_machine[0].item()))
    assert worst <= 1e-6, worst
    if tag == "q3_behind":
        assert {1, 2, 3} <= seen                      # stop -> rotate -> track


def test_quad3d_example_passes_its_first_obstacle():
    """The reference scene of --model quad3d.  With round 2's interior point (no restoration phase) the golden run ended with -2
    at step 73: the solver gave up at an infeasible iterate beside the first obstacle and the loop flew that iterate into it.
    With the feasibility restoration every one of the 260 recorded solves converges and the vehicle passes; the batched loop
    reproduces the whole recorded run, return codes included, and never touches an obstacle."""
    tag = "q3_example"
    n = len(G[f"{tag}/ret"])
    assert n == 260 and np.all(G[f"{tag}/ret"] == 0)
    ctl = sca.BatchedTrackingController(G[f"{tag}/x0"][None, :], dict(Q3), obs=G[f"{tag}/obs"], device=DEV)
    ctl.set_waypoints(G[f"{tag}/waypoints"])
    o = G[f"{tag}/obs"]
    clear = np.inf
    for k in range(n):
        ret = ctl.control_step(1)
        assert int(ret[0].item()) == 0, k
        x = ctl.X[0].cpu().numpy()
        clear = min(clear, (np.hypot(o[:, 0] - x[0], o[:, 1] - x[1]) - o[:, 2] - 0.25).min())
    assert clear > 0.0
    assert np.abs(ctl.X[0].cpu().numpy() - G[f"{tag}/X"][-1])[:2].max() < 1e-4


@pytest.mark.parametrize("spec", [Q2, Q3])
def test_batch_agrees_with_single_agents_and_f32_runs(spec):
    rng = np.random.default_rng(3)
    B = 96
    obs = G["q2_example/obs"]
    P = rng.uniform(0.5, 13.5, (8 * B, 2))
    clear = (np.hypot(P[:, None, 0] - obs[None, :, 0], P[:, None, 1] - obs[None, :, 1]) - obs[None, :, 2]).min(axis=1) > 0.8
    P = P[clear][:B]
    X0 = P if spec["model"] == "Quad2D" else np.column_stack([P, rng.uniform(0.5, 1.5, B), rng.uniform(-3, 3, B)])
    wps = [np.column_stack([rng.uniform(1, 13, (2, 2)), rng.uniform(0.5, 1.5, 2)]) for _ in range(B)]
    ctl = sca.BatchedTrackingController(X0, dict(spec), obs=obs, device=DEV)
    ctl.set_waypoints(wps)
    ctl.control_step(12)
    for i in (0, 17, 95):
        one = sca.BatchedTrackingController(X0[i][None, :], dict(spec), obs=obs, device=DEV)
        one.set_waypoints([wps[i]])
        one.control_step(12)
        assert torch.equal(one.X[0], ctl.X[i]) and int(one.ret[0]) == int(ctl.ret[i])
    c32 = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f32", device=DEV)
    c32.set_waypoints(wps)
    c32.control_step(12)
    same = (c32.ret == ctl.ret)
    assert same.double().mean() > 0.95
    # f32 arrays round the states fed to every solve; an agent whose solve sits at a branch of the solver (restoration entered or
    # not, a line search decided at the round-off level) can take another input there, so the bound is on all but a few agents
    dev = (c32.X.double() - ctl.X)[same][:, :2].abs().max(dim=1).values
    assert torch.quantile(dev, 0.95) < 5e-3
