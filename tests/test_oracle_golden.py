"""CPU tests: the float64 oracle against the golden vectors captured from the reference.

These pin the oracle (tests/golden/README.md says which reference lines made
each file).  Tolerances are float64 round-off class: the oracle reassociates a
few products, nothing more.
"""
import os

import numpy as np
import pytest

from oracle import cbf_qp, qp, robots as R, tracking

RTOL = 1e-10
ATOL = 1e-10

MODELS = {
    "DynamicUnicycle2D": R.MODEL_DU,
    "KinematicBicycle2D": R.MODEL_KB,
    "KinematicBicycle2D_C3BF": R.MODEL_KB_C3BF,
    "KinematicBicycle2D_DPCBF": R.MODEL_KB_DPCBF,
}


def spec_for(model):
    s = R.default_spec(model)
    if model == R.MODEL_DU:
        s.update(a_max=1.0, w_max=0.5, radius=0.25)
    else:
        s.update(a_max=5.0, radius=0.3)
    return s


@pytest.fixture(scope="module")
def callbacks(golden_dir):
    return np.load(os.path.join(golden_dir, "callbacks.npz"))


@pytest.mark.parametrize("name", list(MODELS))
def test_dynamics_and_nominal_inputs(callbacks, name):
    m = MODELS[name]
    spec = spec_for(m)
    g = {k.split("/", 1)[1]: callbacks[k] for k in callbacks.files if k.startswith(name + "/")}
    for i in range(len(g["X"])):
        X = g["X"][i]
        np.testing.assert_allclose(R.f(m, X, spec), g["f"][i], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(R.g(m, X, spec), g["g"][i], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(R.step(m, X, g["U"][i], 0.05, spec), g["step"][i], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(R.nominal_input(m, X, g["goal"][i], spec), g["nominal"][i], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(R.stop(m, X, spec), g["stop"][i], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(R.rotate_to(m, X, g["theta_des"][i]), g["rotate"][i], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("name", list(MODELS))
def test_agent_barrier(callbacks, name):
    m = MODELS[name]
    spec = spec_for(m)
    g = {k.split("/", 1)[1]: callbacks[k] for k in callbacks.files if k.startswith(name + "/")}
    for i in range(len(g["X"])):
        bar = R.agent_barrier(m, g["X"][i], g["obs"][i], spec["radius"])
        np.testing.assert_allclose(bar[0], g["h"][i], rtol=1e-9, atol=1e-9)
        if len(bar) == 3:
            np.testing.assert_allclose(bar[1], g["g1"][i], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(bar[2], g["g2"][i], rtol=1e-9, atol=1e-8)
        else:
            np.testing.assert_allclose(bar[1], g["g2"][i], rtol=1e-9, atol=1e-8)


def test_du_bad_obstacle_flag_raises():
    with pytest.raises(ValueError):
        R.agent_barrier(R.MODEL_DU, np.array([0, 0, 0, 1.0]), np.array([2, 0, .5, 0, 0, 0, 2.0]), 0.25)


GROUPS = {
    "du_circle": (R.MODEL_DU, "cbf"), "du_circle_hard": (R.MODEL_DU, "hard"),
    "du_superellipsoid": (R.MODEL_DU, "cbf"), "du_mixed_trunc": (R.MODEL_DU, "cbf"),
    "du_overlap": (R.MODEL_DU, "cbf"), "kb_circle": (R.MODEL_KB, "cbf"),
    "c3bf": (R.MODEL_KB_C3BF, "cbf"), "c3bf_k16": (R.MODEL_KB_C3BF, "cbf"),
    "dpcbf": (R.MODEL_KB_DPCBF, "cbf"),
}


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(os.path.join(golden_dir, "cbfqp_cases.npz"))


@pytest.mark.parametrize("gname", list(GROUPS))
def test_cbfqp_rows_and_solution(cases, gname):
    """A/b rows equal what the reference's CBFQP.solve_control_problem wrote; u*, status reproduce."""
    model, mode = GROUPS[gname]
    g = {k.split("/", 1)[1]: cases[k] for k in cases.files if k.startswith(gname + "/")}
    num_obs = int(g["meta"][0])
    spec = spec_for(model)
    for i in range(len(g["X"])):
        K = int(g["k"][i])
        obs = list(g["obs"][i][:K])
        r = cbf_qp.solve(model, g["X"][i], g["u_ref"][i], obs, spec, num_obs=num_obs, dt=0.05, cbf_mode=mode)
        scale = 1.0 + np.abs(g["b"][i])
        np.testing.assert_allclose(r["A"], g["A"][i], rtol=1e-9, atol=1e-8)
        assert np.all(np.abs(r["b"] - g["b"][i]) <= 1e-9 * scale)
        assert r["status"] == int(g["status_oracle"][i])
        if r["status"] == 0:
            np.testing.assert_allclose(r["u"], g["u_star_oracle"][i], rtol=1e-8, atol=1e-8)
        # rows beyond the obstacles given stay zero (cbf_qp.py:110-111)
        used = min(K, num_obs)
        assert np.all(r["A"][used:] == 0) and np.all(r["b"][used:] == 0)


def test_obs_none_returns_uref_unclipped(cases):
    r = cbf_qp.solve(R.MODEL_DU, np.array([1.0, 2.0, 0.3, 0.5]), cases["none/u_ref"], None, spec_for(R.MODEL_DU))
    np.testing.assert_array_equal(r["u"], cases["none/u"])
    assert r["status"] == 0 and bool(cases["none/status_optimal"][0])


def test_qp_enumerator_against_slsqp():
    """Independent check of the exact enumerator with scipy SLSQP on random feasible QPs."""
    from scipy.optimize import minimize
    rng = np.random.default_rng(3)
    checked = 0
    for _ in range(300):
        m = int(rng.integers(1, 9))
        G = rng.normal(size=(m, 2)); c = rng.normal(size=m) + 0.8
        Gb, cb = qp.box_rows([-1.0, -0.5], [1.0, 0.5])
        Gf, cf = np.vstack([G, Gb]), np.concatenate([c, cb])
        u_ref = rng.normal(size=2) * 1.5
        u, st = qp.solve_qp2(Gf, cf, u_ref)
        margin = qp.feasibility_margin(Gf, cf)
        if abs(margin) < 1e-6:
            continue
        assert (st == 0) == (margin > 0)
        if st != 0:
            continue
        res = minimize(lambda x: np.sum((x - u_ref) ** 2), u, jac=lambda x: 2 * (x - u_ref),
                       constraints=[{"type": "ineq", "fun": lambda x: Gf @ x + cf, "jac": lambda x: Gf}],
                       method="SLSQP", options={"ftol": 1e-14, "maxiter": 200})
        assert res.success
        np.testing.assert_allclose(res.x, u, atol=2e-6)
        assert np.sum((u - u_ref) ** 2) <= res.fun + 1e-9
        checked += 1
    assert checked > 100


def test_qp_degenerate_rows():
    Gb, cb = qp.box_rows([-1.0, -1.0], [1.0, 1.0])
    # all-zero CBF rows (0 u + 0 >= 0) leave the box projection
    u, st = qp.solve_qp2(np.vstack([np.zeros((3, 2)), Gb]), np.concatenate([np.zeros(3), cb]), [2.0, -3.0])
    assert st == 0 and np.allclose(u, [1.0, -1.0])
    # zero row with negative offset is infeasible
    u, st = qp.solve_qp2(np.vstack([np.zeros((1, 2)), Gb]), np.concatenate([[-1.0], cb]), [0.0, 0.0])
    assert st == 1 and u is None
    # duplicate rows
    G = np.array([[1.0, 1.0], [1.0, 1.0], [2.0, 2.0]]); c = np.array([-0.5, -0.5, -1.0])
    u, st = qp.solve_qp2(np.vstack([G, Gb]), np.concatenate([c, cb]), [0.0, 0.0])
    assert st == 0 and np.allclose(u, [0.25, 0.25])
    # u_ref exactly on a bound, CBF row active at a vertex
    u, st = qp.solve_qp2(np.vstack([[[0.0, 1.0]], Gb]), np.concatenate([[-1.0], cb]), [1.0, 0.0])
    assert st == 0 and np.allclose(u, [1.0, 1.0])
    # NaN data
    u, st = qp.solve_qp2(np.vstack([[[np.nan, 1.0]], Gb]), np.concatenate([[0.0], cb]), [0.0, 0.0])
    assert st == 1


@pytest.mark.parametrize("name", ["DynamicUnicycle2D", "KinematicBicycle2D_C3BF"])
def test_nearest_unpassed_obs(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "nearest_obs.npz"))
    m = MODELS[name]
    for i in range(len(g[f"{name}/X"])):
        X = g[f"{name}/X"][i]
        M = int(g[f"{name}/m"][i])
        got = tracking.get_nearest_unpassed_obs(m, g[f"{name}/table"][i][:M], X[:2], X[2], 10)
        n = int(g[f"{name}/nsel"][i])
        assert got.shape[0] == n
        np.testing.assert_array_equal(got, g[f"{name}/sel"][i][:n])


def test_nearest_none_when_empty():
    assert tracking.get_nearest_unpassed_obs(R.MODEL_DU, np.zeros((0, 7)), [0, 0], 0.0, 10) is None


@pytest.mark.parametrize("tag", ["du14", "du3"])
def test_closed_loop_config1(golden_dir, tag):
    """BASELINE config 1 (examples/test_tracking.py --model du --algo cbf_qp), first 600 steps."""
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    spec = {"a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    wps = g[f"{tag}/waypoints"]
    t = tracking.TrackingOracle(R.MODEL_DU, np.append(wps[0], 1.0), spec, dt=0.05, obs=g[f"{tag}/obs"],
                                num_constraints=10)
    t.set_waypoints(wps)
    Xg, Ug, retg = g[f"{tag}/X"], g[f"{tag}/U"], g[f"{tag}/ret"]
    n = min(600, len(retg))
    for k in range(n):
        ret = t.control_step()
        assert ret == retg[k]
        if ret == -2:
            break
        np.testing.assert_allclose(t.u_pos, Ug[k], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(t.X, Xg[k + 1], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("tag,model", [("c3bf_dyn", R.MODEL_KB_C3BF), ("dpcbf_dyn", R.MODEL_KB_DPCBF)])
def test_closed_loop_moving_obstacles(golden_dir, tag, model):
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    spec = {"a_max": 5.0, "radius": 0.3}
    wps = g[f"{tag}/waypoints"]
    t = tracking.TrackingOracle(model, np.append(wps[0], 1.0), spec, dt=0.05, obs=g[f"{tag}/obs0"],
                                num_constraints=10, dyn_obs=True)
    t.set_waypoints(wps)
    Xg, Ug, retg = g[f"{tag}/X"], g[f"{tag}/U"], g[f"{tag}/ret"]
    for k in range(len(retg)):
        ret = t.control_step()
        assert ret == retg[k]
        if ret != 0:
            break
        np.testing.assert_allclose(t.u_pos, Ug[k], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t.X, Xg[k + 1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(t.obs, g[f"{tag}/obs_final"], rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ integrator models
@pytest.mark.parametrize("name,model", [("SingleIntegrator2D", R.MODEL_SI), ("DoubleIntegrator2D", R.MODEL_DI)])
def test_integrator_models_against_reference(golden_dir, name, model):
    """tests/golden/integrators.npz: f, g, step, nominal_input, stop, CBF rows (reference code verbatim), u*."""
    g = np.load(os.path.join(golden_dir, "integrators.npz"))
    G = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(name + "/")}
    spec = R.default_spec(model)
    if model == R.MODEL_DI:
        spec.update(a_max=1.5)
    nx = 2 if model == R.MODEL_SI else 4
    for i in range(len(G["X"])):
        X, K = G["X"][i], int(G["k"][i])
        np.testing.assert_allclose(R.f(model, X, spec)[:nx], G["f"][i], atol=1e-12)
        np.testing.assert_allclose(R.g(model, X, spec)[:nx], G["g"][i], atol=1e-12)
        np.testing.assert_allclose(R.nominal_input(model, X, G["goal"][i], spec), G["nominal"][i], atol=1e-12)
        np.testing.assert_allclose(R.stop(model, X, spec), G["stop"][i], atol=1e-12)
        np.testing.assert_allclose(R.step(model, X, G["U"][i], 0.05, spec), G["step"][i], atol=1e-12)
        r = cbf_qp.solve(model, X, G["u_ref"][i], list(G["obs"][i][:K]), spec, num_obs=6)
        np.testing.assert_allclose(r["A"], G["A"][i], rtol=1e-9, atol=1e-8)
        assert np.all(np.abs(r["b"] - G["b"][i]) <= 1e-9 * (1 + np.abs(G["b"][i])))
        assert r["status"] == int(G["status_oracle"][i])
        if r["status"] == 0:
            np.testing.assert_allclose(r["u"], G["u_star_oracle"][i], rtol=1e-8, atol=1e-8)


def test_quad2d_against_reference(golden_dir):
    """tests/golden/quad2d.npz: f, g, step, agent_barrier, CBF rows (reference code verbatim), u*."""
    g = np.load(os.path.join(golden_dir, "quad2d.npz"))
    G = {k.split("/", 1)[1]: g[k] for k in g.files}
    m = R.MODEL_QUAD2D
    spec = R.default_spec(m)
    spec.update(f_min=3.0, f_max=10.0)
    for i in range(len(G["X"])):
        X, K = G["X"][i], int(G["k"][i])
        np.testing.assert_allclose(R.f(m, X, spec), G["f"][i], atol=1e-12)
        np.testing.assert_allclose(R.g(m, X, spec), G["g"][i], atol=1e-12)
        np.testing.assert_allclose(R.step(m, X, G["U"][i], 0.05, spec), G["step"][i], atol=1e-12)
        h, hd, dhd = R.agent_barrier(m, X, G["obs"][i][0], spec["radius"])
        np.testing.assert_allclose([h, hd], [G["h"][i], G["hdot"][i]], rtol=1e-10)
        np.testing.assert_allclose(dhd, G["dhd"][i], rtol=1e-10, atol=1e-12)
        r = cbf_qp.solve(m, X, G["u_ref"][i], list(G["obs"][i][:K]), spec, num_obs=6)
        np.testing.assert_allclose(r["A"], G["A"][i], rtol=1e-9, atol=1e-8)
        assert np.all(np.abs(r["b"] - G["b"][i]) <= 1e-9 * (1 + np.abs(G["b"][i])))
        assert r["status"] == int(G["status_oracle"][i])
        if r["status"] == 0:
            np.testing.assert_allclose(r["u"], G["u_star_oracle"][i], rtol=1e-8, atol=1e-8)


def test_unicycle2d_against_reference(golden_dir):
    """tests/golden/unicycle2d.npz: f, g, step, nominal_input, stop, rotate_to and agent_barrier of the reference's
    Unicycle2D (barrier called with a column obstacle, see make_golden.gen_unicycle2d); rows per cbf_qp.py:155-165."""
    g = np.load(os.path.join(golden_dir, "unicycle2d.npz"))
    G = {k.split("/", 1)[1]: g[k] for k in g.files}
    m = R.MODEL_UNI
    spec = R.default_spec(m)
    for i in range(len(G["X"])):
        X, K = G["X"][i], int(G["k"][i])
        np.testing.assert_allclose(R.f(m, X, spec), G["f"][i], atol=1e-12)
        np.testing.assert_allclose(R.g(m, X, spec), G["g"][i], atol=1e-12)
        np.testing.assert_allclose(R.step(m, X, G["U"][i], 0.05, spec), G["step"][i], atol=1e-12)
        np.testing.assert_allclose(R.nominal_input(m, X, G["goal"][i], spec), G["nominal"][i], atol=1e-12)
        np.testing.assert_allclose(R.stop(m, X, spec), G["stop"][i], atol=1e-12)
        np.testing.assert_allclose(R.rotate_to(m, X, 0.7), G["rotate"][i], atol=1e-12)
        for r in range(K):
            h, dh = R.agent_barrier(m, X, G["obs"][i][r], spec["radius"])
            np.testing.assert_allclose(h, G["h"][i][r], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(dh[:3], G["dh"][i][r], rtol=1e-11, atol=1e-12)
        r_ = cbf_qp.solve(m, X, G["u_ref"][i], list(G["obs"][i][:K]), spec, num_obs=6)
        np.testing.assert_allclose(r_["A"], G["A"][i], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(r_["b"], G["b"][i], rtol=1e-10, atol=1e-10)
        assert r_["status"] == int(G["status_oracle"][i])
        if r_["status"] == 0:
            np.testing.assert_allclose(r_["u"], G["u_star_oracle"][i], rtol=1e-9, atol=1e-9)


def test_unicycle2d_sigma_has_no_overflow_far_behind_the_robot():
    """The reference's sigma(s) = k2 (e^(k1-s) - 1) / (e^(k1-s) + 1) is NaN once e^(k1-s) overflows (an obstacle more
    than ~709 m behind the heading, e.g. the [1000, 1000] padding rows); the restated tanh form is finite there and
    equal elsewhere."""
    m = R.MODEL_UNI
    X = np.array([0.0, 0.0, 0.0, 0.0])
    h, dh = R.agent_barrier(m, X, np.array([1000.0, 0.0, 0.5, 0, 0, 0, 0]), 0.25)
    assert np.isfinite(h) and np.all(np.isfinite(dh))
    for s_ in (-3.0, 0.0, 2.5):
        k1, k2 = R.UNI_K1, R.UNI_K2
        ref = k2 * (np.exp(k1 - s_) - 1) / (np.exp(k1 - s_) + 1)
        assert abs(ref - k2 * np.tanh(0.5 * (k1 - s_))) < 1e-14


@pytest.mark.parametrize("tag,model", [("si", R.MODEL_SI), ("di", R.MODEL_DI), ("di_back", R.MODEL_DI)])
def test_integrator_closed_loops_match_reference(golden_dir, tag, model):
    """LocalTrackingController with SingleIntegrator2D / DoubleIntegrator2D, enable_rotation=False
    (tests/golden/make_golden.py: gen_closed_loop_integrators): every state of the reference's run to the last waypoint."""
    from oracle import tracking as T
    g = np.load(os.path.join(golden_dir, "closed_loop_integrators.npz"))
    x0 = g[f"{tag}/x0"]
    if model == R.MODEL_SI:
        X0, yaw, spec = np.array([x0[0], x0[1], 0.0, 0.0]), x0[2], dict(v_max=1.0, radius=0.25)
    else:
        X0, yaw, spec = x0[:4], x0[4], dict(v_max=1.0, a_max=1.0, radius=0.25)
    o = T.TrackingOracle(model, X0, spec, obs=g[f"{tag}/obs"], num_constraints=10, enable_rotation=False, yaw0=yaw)
    o.set_waypoints(g[f"{tag}/waypoints"])
    assert ["idle", "track", "stop", "rotate"].index(o.state_machine) == g[f"{tag}/sm"][0]
    Xr, rets = g[f"{tag}/X"], g[f"{tag}/ret"]
    for i in range(len(rets)):
        assert o.control_step() == rets[i]
        np.testing.assert_allclose(o.X[: Xr.shape[1]], Xr[i + 1], rtol=0, atol=1e-9)
    assert rets[-1] == -1
