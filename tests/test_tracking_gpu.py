"""GPU tests of the fused closed-loop rollout (csrc/tracking.hip) against oracle/tracking.py, which is
itself pinned to the reference's closed-loop trajectories (tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import robots as R, tracking  # noqa: E402
import safe_control_amd as sca  # noqa: E402

DEV = "cuda:0"
SM = {"idle": 0, "track": 1, "stop": 2, "rotate": 3}


def oracle_rollout(model, X0, spec, obs, wps, T, dyn=False, num_constraints=10):
    t = tracking.TrackingOracle(model, X0, spec, dt=0.05, obs=obs, num_constraints=num_constraints, dyn_obs=dyn)
    t.set_waypoints(wps)
    Xs, rets = [], []
    for k in range(T):
        ret = t.control_step()
        Xs.append(t.X.copy()); rets.append(ret)
        if ret != 0:
            break
    return np.array(Xs), rets, t


def test_reference_config1_scene_single_agent(golden_dir):
    """examples/test_tracking.py --model du --algo cbf_qp: the reference's own trajectory, 700 steps in 4 launches."""
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    wps = g["du14/waypoints"]
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    ctl = sca.BatchedTrackingController(np.append(wps[0], 1.0)[None, :], spec, obs=g["du14/obs"], io_dtype="f64")
    ctl.set_waypoints(wps)
    Xg, Ug, retg = g["du14/X"], g["du14/U"], g["du14/ret"]
    done = 0
    for n in (1, 199, 250, 250):
        ret, tX, tU = ctl.control_step(n, record=True)
        tX = tX.cpu().numpy()[:, 0]; tU = tU.cpu().numpy()[:, 0]
        np.testing.assert_allclose(tX, Xg[done + 1: done + n + 1], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tU, Ug[done: done + n], rtol=1e-6, atol=1e-6)
        done += n
    assert int(ret[0].item()) == 0 and np.all(retg[:700] == 0)


@pytest.mark.parametrize("model_name,tag", [("KinematicBicycle2D_C3BF", "c3bf_dyn"), ("KinematicBicycle2D_DPCBF", "dpcbf_dyn")])
def test_reference_moving_obstacle_runs(golden_dir, model_name, tag):
    """dynamic_env/main.py scenario: obstacles move after the selection of each step; run ends with -1."""
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    wps = g[f"{tag}/waypoints"]
    spec = {"model": model_name, "a_max": 5.0, "radius": 0.3}
    ctl = sca.BatchedTrackingController(np.append(wps[0], 1.0)[None, :], spec, obs=g[f"{tag}/obs0"], dyn_obs=True)
    ctl.set_waypoints(wps)
    Xg, retg = g[f"{tag}/X"], g[f"{tag}/ret"]
    T = len(retg)
    ret, tX, tU = ctl.control_step(T + 5, record=True)
    tX = tX.cpu().numpy()[:, 0]
    np.testing.assert_allclose(tX[:T], Xg[1: T + 1], rtol=1e-6, atol=1e-6)
    assert int(ret[0].item()) == -1 and int(ctl.ret_step[0].item()) == T - 1
    np.testing.assert_allclose(tX[T:], np.repeat(Xg[T][None], 5, 0), atol=1e-9)     # frozen after finishing
    np.testing.assert_allclose(ctl.obs.cpu().numpy(), g[f"{tag}/obs_final"] + 5 * 0.05 * np.pad(g[f"{tag}/obs0"][:, 3:5], ((0, 0), (0, 5)))[:, [0, 1, 2, 3, 4, 5, 6]] * 0
                               + np.hstack([5 * 0.05 * g[f"{tag}/obs0"][:, 3:5], np.zeros((8, 5))]), atol=1e-9)


@pytest.mark.parametrize("lane_per_agent", ["0", "1"])
def test_moving_obstacles_large_grid_every_block_sees_the_same_table(golden_dir, lane_per_agent, monkeypatch):
    """131072 copies of the reference's moving-obstacle agent: a grid many times larger than what the chip keeps resident, so
    late blocks start after early ones have finished.  Every copy must reproduce the reference trajectory bit for bit like
    copy 0 (the table a block reads is the table at launch; the advance is a stream-ordered follow-up kernel), across
    two launches, and the table must end where n_steps of obs += v dt put it."""
    monkeypatch.setenv("SC_TRACK_LANE_PER_AGENT", lane_per_agent)
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    tag = "c3bf_dyn"
    wps = g[f"{tag}/waypoints"]
    spec = {"model": "KinematicBicycle2D_C3BF", "a_max": 5.0, "radius": 0.3}
    B = 131072
    X0 = np.repeat(np.append(wps[0], 1.0)[None, :], B, 0)
    ctl = sca.BatchedTrackingController(X0, spec, obs=g[f"{tag}/obs0"], dyn_obs=True)
    ctl.set_waypoints(np.repeat(wps[None], B, 0))
    Xg = g[f"{tag}/X"]
    T1, T2 = 40, 35
    ctl.control_step(T1)
    X1 = ctl.X.cpu().numpy()
    assert np.array_equal(X1, np.repeat(X1[:1], B, 0))
    np.testing.assert_allclose(X1[0], Xg[T1], rtol=1e-6, atol=1e-6)
    ctl.control_step(T2)
    X2 = ctl.X.cpu().numpy()
    assert np.array_equal(X2, np.repeat(X2[:1], B, 0))
    np.testing.assert_allclose(X2[0], Xg[T1 + T2], rtol=1e-6, atol=1e-6)
    obs0 = g[f"{tag}/obs0"]
    want = obs0.copy()
    for _ in range(T1 + T2):
        want[:, 0:2] += want[:, 3:5] * 0.05
    np.testing.assert_allclose(ctl.obs.cpu().numpy(), want, rtol=0, atol=1e-12)


def test_finish_step_and_last_input_survive_later_launches(golden_dir):
    """run_all_steps works in chunks: an agent that finished (or failed) in an earlier launch keeps its absolute ret_step
    and its last applied input; the fused rollout and the select / apply split agree on both."""
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    obs = g["du14/obs"]
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    X0 = np.array([[2.0, 2.0, np.pi / 2, 1.0], [6.0, 1.0, 2.6, 0.3], [1.0, 6.0, -1.2, 0.0]])
    wl = [np.array([[2.0, 3.2]]), np.array([[5.2, 1.4]]), np.array([[1.0, 12.0]])]     # two short routes, one long
    one = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f64")
    one.set_waypoints(wl)
    ret1, tX1, tU1 = one.control_step(120, record=True)
    rs1 = one.ret_step.cpu().numpy().copy(); ul1 = one.u_pos.cpu().numpy().copy(); ret1 = ret1.cpu().numpy().copy()
    assert (ret1[:2] == -1).all() and ret1[2] == 0 and (rs1[:2] >= 0).all() and (rs1[:2] < 100).all()
    many = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f64")
    many.set_waypoints(wl)
    for n in (7, 33, 50, 30):
        many.control_step(n)
    assert np.array_equal(many.ret.cpu().numpy(), ret1)
    assert np.array_equal(many.ret_step.cpu().numpy(), rs1)
    np.testing.assert_array_equal(many.u_pos.cpu().numpy(), ul1)
    np.testing.assert_array_equal(many.X.cpu().numpy(), one.X.cpu().numpy())
    tU1 = tU1.cpu().numpy()
    for i in range(2):                                                           # the last input is the one applied at the finishing step
        np.testing.assert_array_equal(ul1[i], tU1[rs1[i], i])


def test_many_agents_static_scene_against_oracle(golden_dir):
    """48 agents scattered over the config-1 scene, each with its own start pose and waypoint list."""
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    obs = g["du14/obs"]
    rng = np.random.default_rng(42)
    B, T = 48, 260
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    X0, wlists = [], []
    while len(X0) < B:
        p = rng.uniform(0.5, 13.5, 2)
        if np.min(np.hypot(obs[:, 0] - p[0], obs[:, 1] - p[1]) - obs[:, 2]) < 0.6:
            continue
        X0.append([p[0], p[1], rng.uniform(-np.pi, np.pi), rng.uniform(0, 1)])
        wlists.append(rng.uniform(1, 13, (int(rng.integers(1, 4)), 2)))
    X0 = np.array(X0)
    ctl = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f64")
    ctl.set_waypoints(wlists)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); ret = ret.cpu().numpy(); rstep = ctl.ret_step.cpu().numpy()
    ospec = {k: v for k, v in spec.items() if k != "model"}
    n_finished = n_failed = 0
    for i in range(B):
        Xo, rets, t = oracle_rollout(R.MODEL_DU, X0[i], ospec, obs, wlists[i], T)
        n = len(rets)
        np.testing.assert_allclose(tX[:n, i], Xo, rtol=1e-6, atol=1e-6, err_msg=f"agent {i}")
        if rets[-1] != 0:
            assert ret[i] == rets[-1] and rstep[i] == n - 1, (i, ret[i], rets[-1], rstep[i], n)
            n_finished += rets[-1] == -1; n_failed += rets[-1] == -2
        else:
            assert ret[i] == 0
        assert int(ctl.state_machine[i].item()) == SM[t.state_machine]
        assert int(ctl.current_goal_index[i].item()) == t.current_goal_index
    assert n_finished > 0            # the scenario exercises the finished (-1) path; failures (-2) may or may not occur


def test_no_obstacles_passes_u_ref_through():
    """obs_list None -> u = u_ref, unclipped (cbf_qp.py:113-118): the agent just follows the nominal controller."""
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    X0 = np.array([[1.0, 1.0, 0.2, 0.0]])
    wps = np.array([[6.0, 2.0], [6.0, 6.0]])
    ctl = sca.BatchedTrackingController(X0, dict(spec), obs=None)
    ctl.set_waypoints(wps)
    ret, tX, tU = ctl.control_step(100, record=True)
    Xo, rets, t = oracle_rollout(R.MODEL_DU, X0[0], {k: v for k, v in spec.items() if k != "model"}, None, wps, 100)
    np.testing.assert_allclose(tX.cpu().numpy()[: len(rets), 0], Xo, rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("num_constraints,n_obs", [(10, 14), (8, 14), (4, 40)])
def test_cooperative_and_lane_per_agent_kernels_agree(golden_dir, num_constraints, n_obs):
    """The cooperative rollout (G lanes per agent: ranked selection, one row per lane, cooperative walk) and the
    lane-per-agent rollout (sorted insertion, sequential walk) are two implementations of the same step."""
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    rng = np.random.default_rng(7)
    obs = g["du14/obs"]
    if n_obs > len(obs):                                         # extra small circles, some at equal distances (ties)
        extra = np.zeros((n_obs - len(obs), 7))
        extra[:, 0:2] = rng.uniform(0.5, 13.5, (len(extra), 2))
        extra[:, 2] = 0.15
        extra[1] = extra[0]                                      # a duplicate: identical distance from every agent
        obs = np.vstack([obs, extra])
    B, T = 200, 150
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25, "num_constraints": num_constraints}
    X0 = np.column_stack([rng.uniform(0.5, 13.5, (B, 2)), rng.uniform(-np.pi, np.pi, B), rng.uniform(0, 1, B)])
    wl = [rng.uniform(1, 13, (3, 2)) for _ in range(B)]
    out = {}
    for mode in ("0", "1"):
        os.environ["SC_TRACK_LANE_PER_AGENT"] = mode
        try:
            ctl = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f64")
            ctl.set_waypoints(wl)
            ret, tX, tU = ctl.control_step(T, record=True)
            out[mode] = (ret.cpu().numpy(), tX.cpu().numpy(), tU.cpu().numpy(), ctl.ret_step.cpu().numpy())
        finally:
            os.environ.pop("SC_TRACK_LANE_PER_AGENT", None)
    assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][3], out["1"][3])
    np.testing.assert_allclose(out["0"][1], out["1"][1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["0"][2], out["1"][2], rtol=0, atol=1e-9)
    assert (out["0"][0] != 0).any() and (out["0"][0] == 0).any()


@pytest.mark.parametrize("pos", ["mpc_cbf", "mpc_cbf/condensed", "optimal_decay_mpc_cbf"])
def test_closed_loop_with_mpc_position_controller(golden_dir, pos):
    """examples/test_tracking.py's default --algo mpc_cbf, batched: select -> one MPC launch -> apply per step, against
    the oracle loop with the oracle MPC behind solve_fn (u_prev feedback, u_ref pass-through when not tracking).  'mpc_cbf': the
    default formulation since round 6 -- multiple shooting under IPOPT's algorithm (kernel 13) against oracle/ms_ipopt.py;
    'mpc_cbf/condensed': robot_spec['mpc_formulation'] = 'condensed' (kernel 3) against oracle/mpc_cbf.py."""
    from oracle import mpc_cbf as M, od_mpc_cbf as O, ms_ipopt as MS
    pos, _, form = pos.partition("/")
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    obs = g["du14/obs"]
    K = 8
    spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25, "num_constraints": K}
    if form:
        spec["mpc_formulation"] = form
    X0 = np.array([[2.0, 2.0, np.pi / 2, 1.0], [6.0, 1.0, 2.6, 0.3], [1.0, 6.0, -1.2, 0.0]])   # the last one starts in 'stop'
    wl = [np.array([[2.0, 12.0], [12.0, 12.0]]), np.array([[1.0, 4.0]]), np.array([[1.0, 12.0]])]
    T = 30
    ctl = sca.BatchedTrackingController(X0, dict(spec), controller_type={"pos": pos}, obs=obs, io_dtype="f64")
    ctl.set_waypoints(wl)
    assert (ctl.mpc_ms is not None) == (pos == "mpc_cbf" and not form)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); tU = tU.cpu().numpy(); ret = ret.cpu().numpy()
    ospec = {k: v for k, v in spec.items() if k not in ("model", "num_constraints", "mpc_formulation")}
    n_track = 0
    for i in range(len(X0)):
        state = {"up": np.zeros(2)}

        def solve_fn(X, cref, nobs, state=state):
            if cref["state_machine"] != "track":
                return np.asarray(cref["u_ref"], dtype=np.float64).reshape(-1), 0
            o = M.pad_obstacles(None if nobs is None else list(nobs), K if pos == "mpc_cbf" else 5)   # OD: five slots
            if pos == "mpc_cbf" and not form:
                u, st, it = MS.solve(MS.du_model(), X, state["up"], cref["goal"], o, opts=MS.KERNEL_PROFILE)
            elif pos == "mpc_cbf":
                u, st, it = M.solve(X, state["up"], cref["goal"], o)
            else:
                u, rho, st, it = O.solve(X, state["up"], cref["goal"], o)
            state["up"] = u
            state["n"] = state.get("n", 0) + 1
            return u, 0                                           # MPCCBF.status stays 'optimal' (mpc_cbf.py:10)

        t = tracking.TrackingOracle(R.MODEL_DU, X0[i], ospec, dt=0.05, obs=obs, num_constraints=K, solve_fn=solve_fn)
        t.set_waypoints(wl[i])
        for k in range(T):
            r = t.control_step()
            np.testing.assert_allclose(tX[k, i], t.X, rtol=0, atol=5e-6, err_msg=f"agent {i} step {k}")
            if r != 0:
                assert ret[i] == r
                break
        n_track += state.get("n", 0)
    assert n_track >= T                                           # the MPC really ran in the loop


def test_unicycle2d_closed_loop_against_oracle(golden_dir):
    """test_tracking.py --model un with the CBF-QP behind it: the reference's CBFQP cannot run this model as checked in
    (tests/golden/make_golden.py gen_unicycle2d), so the loop is checked against oracle/tracking.py over the robot
    functions that ARE pinned on the reference (tests/golden/unicycle2d.npz: step, nominal_input, stop, rotate_to,
    barrier).  3-wide start poses; has_stopped() is always true for this model."""
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    obs = g["du14/obs"]
    rng = np.random.default_rng(11)
    B, T = 40, 300
    spec = {"model": "Unicycle2D", "v_max": 1.0, "w_max": 0.5, "radius": 0.25}
    X0, wlists = [], []
    while len(X0) < B:
        p = rng.uniform(0.5, 13.5, 2)
        if np.min(np.hypot(obs[:, 0] - p[0], obs[:, 1] - p[1]) - obs[:, 2]) < 0.6:
            continue
        X0.append([p[0], p[1], rng.uniform(-np.pi, np.pi)])
        wlists.append(rng.uniform(1, 13, (int(rng.integers(1, 4)), 2)))
    X0 = np.array(X0)
    ospec = {k: v for k, v in spec.items() if k != "model"}
    for mode in ("0", "1"):                                       # cooperative and lane-per-agent kernels
        os.environ["SC_TRACK_LANE_PER_AGENT"] = mode
        try:
            ctl = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f64")
            ctl.set_waypoints(wlists)
            ret, tX, tU = ctl.control_step(T, record=True)
        finally:
            os.environ.pop("SC_TRACK_LANE_PER_AGENT", None)
        tX = tX.cpu().numpy(); ret = ret.cpu().numpy(); rstep = ctl.ret_step.cpu().numpy()
        n_done = 0
        for i in range(B):
            Xo, rets, t = oracle_rollout(R.MODEL_UNI, X0[i], ospec, obs, wlists[i], T)
            n = len(rets)
            np.testing.assert_allclose(tX[:n, i], Xo, rtol=1e-6, atol=1e-6, err_msg=f"agent {i}")
            if rets[-1] != 0:
                assert ret[i] == rets[-1] and rstep[i] == n - 1, (i, ret[i], rets[-1], rstep[i], n)
                n_done += 1
            else:
                assert ret[i] == 0
            assert int(ctl.state_machine[i].item()) == SM[t.state_machine]
        assert n_done > 0


@pytest.mark.parametrize("formulation", ["multiple_shooting", "condensed"])
def test_unicycle2d_closed_loop_with_mpc(golden_dir, formulation):
    """test_tracking.py --model un (default --algo mpc_cbf), batched, against the oracle loop with the Unicycle2D MPC
    oracle behind solve_fn: the multiple-shooting NLP under IPOPT's algorithm (the loop's default: kernel 13, oracle/ms_ipopt.py: uni_model) and
    the condensed solve (robot_spec['mpc_formulation'] = 'condensed', oracle/mpc_cbf_uni.py)."""
    from oracle import mpc_cbf as M, mpc_cbf_uni as U, ms_ipopt as MS
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    obs = g["du14/obs"]
    K = 8
    spec = {"model": "Unicycle2D", "v_max": 1.0, "w_max": 0.5, "radius": 0.25, "num_constraints": K, "mpc_formulation": formulation}
    X0 = np.array([[2.0, 2.0, np.pi / 2], [6.0, 1.0, 2.6], [1.0, 6.0, -1.2]])       # the last one starts in 'stop'
    wl = [np.array([[2.0, 12.0], [12.0, 12.0]]), np.array([[1.0, 4.0]]), np.array([[1.0, 12.0]])]
    T = 30
    ctl = sca.BatchedTrackingController(X0, dict(spec), controller_type={"pos": "mpc_cbf"}, obs=obs, io_dtype="f64")
    ctl.set_waypoints(wl)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); ret = ret.cpu().numpy()
    ospec = {k: v for k, v in spec.items() if k not in ("model", "num_constraints", "mpc_formulation")}
    assert (ctl.mpc_ms is not None) == (formulation == "multiple_shooting")
    ms_model = MS.uni_model(dict(v_max=1.0, w_max=0.5, radius=0.25))
    n_track = 0
    for i in range(len(X0)):
        state = {"up": np.zeros(2)}

        def solve_fn(X, cref, nobs, state=state):
            if cref["state_machine"] != "track":
                return np.asarray(cref["u_ref"], dtype=np.float64).reshape(-1), 0
            o = M.pad_obstacles(None if nobs is None else list(nobs), K)
            if formulation == "multiple_shooting":
                u, st, it = MS.solve(ms_model, X, state["up"], cref["goal"], o, opts=dict(MS.KERNEL_PROFILE))
            else:
                u, st, it = U.solve(X, state["up"], cref["goal"], o)
            state["up"] = u
            state["n"] = state.get("n", 0) + 1
            return u, 0

        t = tracking.TrackingOracle(R.MODEL_UNI, X0[i], ospec, dt=0.05, obs=obs, num_constraints=K, solve_fn=solve_fn)
        t.set_waypoints(wl[i])
        for k in range(T):
            r = t.control_step()
            np.testing.assert_allclose(tX[k, i], t.X, rtol=0, atol=5e-6, err_msg=f"agent {i} step {k}")
            if r != 0:
                assert ret[i] == r
                break
        n_track += state.get("n", 0)
    assert n_track >= T


@pytest.mark.parametrize("tag,model_name", [("si", "SingleIntegrator2D"), ("di", "DoubleIntegrator2D"), ("di_back", "DoubleIntegrator2D")])
def test_integrators_reproduce_the_reference_closed_loop(golden_dir, tag, model_name):
    """examples/test_tracking.py --model si / di --algo cbf_qp with enable_rotation=False: the reference's own trajectories
    (tests/golden/closed_loop_integrators.npz) through the 14-circle scene to the last waypoint, in three launches; the
    'di_back' start faces away from the first goal, so it begins in 'stop' and brakes first."""
    g = np.load(os.path.join(golden_dir, "closed_loop_integrators.npz"))
    spec = {"model": model_name, "v_max": 1.0, "radius": 0.25}
    if model_name == "DoubleIntegrator2D":
        spec["a_max"] = 1.0
    ctl = sca.BatchedTrackingController(g[f"{tag}/x0"][None, :], spec, obs=g[f"{tag}/obs"], io_dtype="f64", enable_rotation=False)
    ctl.set_waypoints(g[f"{tag}/waypoints"])
    assert int(ctl.state_machine[0].item()) == int(g[f"{tag}/sm"][0])
    Xg, Ug, retg = g[f"{tag}/X"], g[f"{tag}/U"], g[f"{tag}/ret"]
    T, nx = len(retg), Xg.shape[1]
    done = 0
    for n in (1, 399, T - 400 + 5):
        ret, tX, tU = ctl.control_step(n, record=True)
        tX = tX.cpu().numpy()[:, 0]; tU = tU.cpu().numpy()[:, 0]
        m = min(n, T - done)
        np.testing.assert_allclose(tX[:m, :nx], Xg[done + 1: done + m + 1], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(tU[:m], Ug[done: done + m], rtol=1e-7, atol=1e-7)
        done += n
    assert int(ret[0].item()) == -1 and int(retg[-1]) == -1 and ctl.steps_done == T + 5
    with pytest.raises(ValueError):
        sca.BatchedTrackingController(g[f"{tag}/x0"][None, :], dict(spec), obs=g[f"{tag}/obs"])    # enable_rotation defaults to True


def test_integrator_fleet_against_oracle(golden_dir):
    """32 DoubleIntegrator2D agents with their own start states / headings / waypoints against oracle/tracking.py."""
    g = np.load(os.path.join(golden_dir, "closed_loop_integrators.npz"))
    obs = g["di/obs"]
    rng = np.random.default_rng(17)
    B, T = 32, 220
    spec = {"model": "DoubleIntegrator2D", "v_max": 1.0, "a_max": 1.0, "radius": 0.25}
    X0, wl = [], []
    while len(X0) < B:
        p = rng.uniform(0.5, 13.5, 2)
        if np.min(np.linalg.norm(obs[:, :2] - p, axis=1) - obs[:, 2]) < 0.6:
            continue
        X0.append([p[0], p[1], *rng.uniform(-0.5, 0.5, 2), rng.uniform(-np.pi, np.pi)])
        wl.append(np.vstack([p, rng.uniform(1, 13, (2, 2))]))
    X0 = np.array(X0)
    ctl = sca.BatchedTrackingController(X0, dict(spec), obs=obs, io_dtype="f64", enable_rotation=False)
    ctl.set_waypoints(wl)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); ret = ret.cpu().numpy()
    n_fin = 0
    for i in range(B):
        t = tracking.TrackingOracle(R.MODEL_DI, X0[i, :4], spec, dt=0.05, obs=obs, num_constraints=10, enable_rotation=False,
                                    yaw0=X0[i, 4])
        t.set_waypoints(wl[i])
        last = 0
        for k in range(T):
            last = t.control_step()
            if last == -2:
                break
            np.testing.assert_allclose(tX[k, i], t.X, rtol=1e-7, atol=1e-7)
            if last != 0:
                break
        assert ret[i] == last
        n_fin += int(last != -2)
    assert n_fin > B // 2                                  # most agents are still under way (or done), not failed


def test_single_integrator_closed_loop_with_mpc(golden_dir):
    """examples/test_tracking.py --model si (default --algo mpc_cbf), batched: select -> linear-model MPC launch -> apply,
    against the oracle loop with oracle/mpc_lin.py behind solve_fn."""
    from oracle import mpc_cbf as M, mpc_lin as L
    g = np.load(os.path.join(golden_dir, "closed_loop_integrators.npz"))
    obs = g["si/obs"]
    K = 6
    spec = {"model": "SingleIntegrator2D", "v_max": 1.0, "radius": 0.25, "num_constraints": K}
    X0 = np.array([[2.0, 2.0, np.pi / 2], [6.0, 1.0, 2.6], [1.0, 6.0, -1.2]])                   # the last one starts in 'stop'
    wl = [np.array([[2.0, 12.0], [12.0, 12.0]]), np.array([[1.0, 4.0]]), np.array([[1.0, 12.0]])]
    T = 40
    ctl = sca.BatchedTrackingController(X0, dict(spec), controller_type={"pos": "mpc_cbf"}, obs=obs, io_dtype="f64",
                                        enable_rotation=False)
    ctl.set_waypoints(wl)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); ret = ret.cpu().numpy()
    mdl = L.si_model({"v_max": 1.0, "radius": 0.25})
    n_track = 0
    for i in range(len(X0)):
        state = {"up": np.zeros(2)}

        def solve_fn(X, cref, nobs, state=state):
            if cref["state_machine"] != "track":
                return np.asarray(cref["u_ref"], dtype=np.float64).reshape(-1), 0
            o = M.pad_obstacles(None if nobs is None else list(nobs), K)
            u, st, it = L.solve(mdl, X[:2], state["up"], cref["goal"], o)
            state["up"] = u
            state["n"] = state.get("n", 0) + 1
            return u, 0

        t = tracking.TrackingOracle(R.MODEL_SI, [X0[i, 0], X0[i, 1], 0.0, 0.0], {"v_max": 1.0, "radius": 0.25}, dt=0.05, obs=obs,
                                    num_constraints=K, solve_fn=solve_fn, enable_rotation=False, yaw0=X0[i, 2])
        t.set_waypoints(wl[i])
        for k in range(T):
            r = t.control_step()
            np.testing.assert_allclose(tX[k, i, :2], t.X[:2], rtol=0, atol=5e-6, err_msg=f"agent {i} step {k}")
            if r != 0:
                assert ret[i] == r
                break
        n_track += state.get("n", 0)
    assert n_track >= T


@pytest.mark.parametrize("formulation", ["multiple_shooting", "condensed"])
def test_double_integrator_closed_loop_with_mpc(golden_dir, formulation):
    """--model di with the default --algo mpc_cbf, batched: select -> MPC launch -> apply, against the oracle loop with the matching oracle
    behind solve_fn: the multiple-shooting NLP under IPOPT's algorithm (the loop's default: kernel 13, oracle/ms_ipopt.py: di_model) and the
    condensed Gauss-Newton solve (robot_spec['mpc_formulation'] = 'condensed': csrc/mpc_gn.hip, oracle/mpc_gn.py)."""
    from oracle import mpc_cbf as M, mpc_gn as G, ms_ipopt as MS
    g = np.load(os.path.join(golden_dir, "closed_loop_integrators.npz"))
    obs = g["di/obs"]
    K = 6
    spec = {"model": "DoubleIntegrator2D", "v_max": 1.0, "a_max": 1.0, "radius": 0.25, "num_constraints": K, "mpc_formulation": formulation}
    X0 = np.array([[2.0, 2.0, 0.0, 0.3, np.pi / 2], [6.0, 1.0, -0.2, 0.1, 2.6], [1.0, 6.0, 0.3, -0.3, -1.2]])   # the last starts in 'stop'
    wl = [np.array([[2.0, 12.0], [12.0, 12.0]]), np.array([[1.0, 4.0]]), np.array([[1.0, 12.0]])]
    T = 40
    ctl = sca.BatchedTrackingController(X0, dict(spec), controller_type={"pos": "mpc_cbf"}, obs=obs, io_dtype="f64",
                                        enable_rotation=False)
    ctl.set_waypoints(wl)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); ret = ret.cpu().numpy()
    mdl = G.di_model({"v_max": 1.0, "a_max": 1.0, "radius": 0.25})
    ms_model = MS.di_model({"v_max": 1.0, "a_max": 1.0, "radius": 0.25})
    assert (ctl.mpc_ms is not None) == (formulation == "multiple_shooting")
    n_track = 0
    for i in range(len(X0)):
        state = {"up": np.zeros(2)}

        def solve_fn(X, cref, nobs, state=state):
            if cref["state_machine"] != "track":
                return np.asarray(cref["u_ref"], dtype=np.float64).reshape(-1), 0
            o = M.pad_obstacles(None if nobs is None else list(nobs), K)
            if formulation == "multiple_shooting":
                u, st, it = MS.solve(ms_model, X[:4], state["up"], cref["goal"], o, opts=dict(MS.KERNEL_PROFILE))
            else:
                u, st, it = G.solve(mdl, X[:4], state["up"], cref["goal"], o)
            state["up"] = u
            state["n"] = state.get("n", 0) + 1
            return u, 0

        t = tracking.TrackingOracle(R.MODEL_DI, X0[i, :4], {"v_max": 1.0, "a_max": 1.0, "radius": 0.25}, dt=0.05, obs=obs,
                                    num_constraints=K, solve_fn=solve_fn, enable_rotation=False, yaw0=X0[i, 4])
        t.set_waypoints(wl[i])
        for k in range(T):
            r = t.control_step()
            np.testing.assert_allclose(tX[k, i], t.X, rtol=0, atol=2e-5, err_msg=f"agent {i} step {k}")
            if r != 0:
                assert ret[i] == r
                break
        n_track += state.get("n", 0)
    assert n_track >= T


def test_kinematic_bicycle_closed_loop_with_mpc():
    """examples/test_tracking.py --model kb (default --algo mpc_cbf) on the example's own scene (:44-52,104-110), batched: select ->
    MPC launch (csrc/mpc_gn.hip, model id 1) -> apply, against the oracle loop with oracle/mpc_gn.py (kb_model) behind solve_fn.
    Three agents: the example's start, and two that start elsewhere on the map."""
    from oracle import mpc_cbf as M, mpc_gn as G
    obs = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0], [4.0, 3.5, 1.5],
                    [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8], [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
    obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
    K = 5
    spec = {"model": "KinematicBicycle2D", "a_max": 0.5, "radius": 0.5, "num_constraints": K}
    X0 = np.array([[2.0, 2.0, np.pi / 2, 1.0], [12.0, 12.5, -1.4, 1.5], [8.0, 2.0, 2.8, 0.6]])
    wl = [np.array([[2.0, 12.0], [12.0, 12.0], [12.0, 2.0]]), np.array([[12.0, 2.0]]), np.array([[2.0, 2.0]])]
    T = 60
    ctl = sca.BatchedTrackingController(X0, dict(spec), controller_type={"pos": "mpc_cbf"}, obs=obs7, io_dtype="f64")
    ctl.set_waypoints(wl)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); ret = ret.cpu().numpy()
    mdl = G.kb_model({"a_max": 0.5, "radius": 0.5})
    n_track = 0
    for i in range(len(X0)):
        state = {"up": np.zeros(2)}

        def solve_fn(X, cref, nobs, state=state):
            if cref["state_machine"] != "track":
                return np.asarray(cref["u_ref"], dtype=np.float64).reshape(-1), 0
            o = M.pad_obstacles(None if nobs is None else list(nobs), K)
            u, st, it = G.solve(mdl, X[:4], state["up"], cref["goal"], o)
            state["up"] = u
            state["n"] = state.get("n", 0) + 1
            return u, 0

        t = tracking.TrackingOracle(R.MODEL_KB, X0[i], {"a_max": 0.5, "radius": 0.5}, dt=0.05, obs=obs7, num_constraints=K,
                                    solve_fn=solve_fn)
        t.set_waypoints(wl[i])
        for k in range(T):
            r = t.control_step()
            np.testing.assert_allclose(tX[k, i], t.X, rtol=0, atol=5e-5, err_msg=f"agent {i} step {k}")
            if r != 0:
                assert ret[i] == r
                break
        n_track += state.get("n", 0)
    assert n_track >= T
    with pytest.raises(ValueError):
        sca.BatchedTrackingController(X0, {"model": "KinematicBicycle2D_C3BF"}, controller_type={"pos": "optimal_decay_mpc_cbf"}, obs=obs7)


@pytest.mark.parametrize("model_name,model_id", [("KinematicBicycle2D_C3BF", R.MODEL_KB_C3BF), ("KinematicBicycle2D_DPCBF", R.MODEL_KB_DPCBF)])
def test_collision_cone_bicycles_closed_loop_with_mpc(model_name, model_id):
    """--model kb with the C3BF / DPCBF robots under the default --algo mpc_cbf: select -> MPC launch with the full-state discrete-time
    barrier (csrc/mpc_gn.hip, model ids 2 / 3) -> apply, against the oracle loop with oracle/mpc_kb_state.py behind solve_fn."""
    from oracle import mpc_cbf as M, mpc_kb_state as S
    obs = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0], [4.0, 3.5, 1.5],
                    [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8], [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
    obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
    K, T = 5, 25
    spec = {"model": model_name, "a_max": 0.5, "radius": 0.5, "num_constraints": K}
    X0 = np.array([[2.0, 2.0, np.pi / 2, 1.0], [12.0, 12.5, -1.4, 1.5]])
    wl = [np.array([[2.0, 12.0]]), np.array([[12.0, 2.0]])]
    ctl = sca.BatchedTrackingController(X0, dict(spec), controller_type={"pos": "mpc_cbf"}, obs=obs7, io_dtype="f64")
    ctl.set_waypoints(wl)
    ret, tX, tU = ctl.control_step(T, record=True)
    tX = tX.cpu().numpy(); ret = ret.cpu().numpy()
    mdl = (S.c3bf_model if model_id == R.MODEL_KB_C3BF else S.dpcbf_model)({"a_max": 0.5, "radius": 0.5})
    n_track = 0
    for i in range(len(X0)):
        state = {"up": np.zeros(2)}

        def solve_fn(X, cref, nobs, state=state):
            if cref["state_machine"] != "track":
                return np.asarray(cref["u_ref"], dtype=np.float64).reshape(-1), 0
            o = M.pad_obstacles(None if nobs is None else list(nobs), K)
            u, st, it = S.solve(mdl, X[:4], state["up"], cref["goal"], o)
            state["up"] = u
            state["n"] = state.get("n", 0) + 1
            return u, 0

        t = tracking.TrackingOracle(model_id, X0[i], {"a_max": 0.5, "radius": 0.5}, dt=0.05, obs=obs7, num_constraints=K, solve_fn=solve_fn)
        t.set_waypoints(wl[i])
        for k in range(T):
            r = t.control_step()
            np.testing.assert_allclose(tX[k, i], t.X, rtol=0, atol=5e-5, err_msg=f"agent {i} step {k}")
            if r != 0:
                assert ret[i] == r
                break
        n_track += state.get("n", 0)
    assert n_track >= T
