"""GPU: MPC-CBF for KinematicBicycle2D (csrc/mpc_gn.hip, model id 1: exact Hessian, speed-bound rows, step() with the speed clip
inside the barrier) through the C-ABI against the numpy oracle (oracle/mpc_gn.py: kb_model; its F / S / barrier are pinned on the
reference's own functions, tests/test_oracle_mpc_pinned.py).  Same bar as tests/test_mpcgn_gpu.py: same status, |u0 - u0_oracle|
<= 1e-6, |z - z_oracle| <= 2e-5 where the oracle converged to its tolerance.  The scene of the closed-loop test is the reference's
own (examples/test_tracking.py:44-52,104-110: four waypoints, fourteen static obstacles, a_max 0.5, radius 0.5, v0 = 1)."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402
from oracle import mpc_cbf as M  # noqa: E402
from oracle import mpc_gn as G  # noqa: E402

DEV = "cuda:0"
NAME = "KinematicBicycle2D"

SCENE_OBS = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0],
                      [4.0, 3.5, 1.5], [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8], [13.5, 11.0, 0.6],
                      [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
SCENE_WPS = np.array([[2.0, 2.0], [2.0, 12.0], [12.0, 12.0], [12.0, 2.0]])


def t(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def draw(mdl, rng, K, clear=1.0):
    """A bicycle driving roughly towards its goal, K - 1 circles around it, one far dummy row."""
    x0 = np.zeros(4); x0[:2] = rng.uniform(0, 14, 2)
    goal = rng.uniform(0, 14, 2)
    x0[2] = math.atan2(goal[1] - x0[1], goal[0] - x0[0]) + rng.uniform(-0.6, 0.6)
    x0[3] = rng.uniform(0.5, 3.0)
    obs = np.tile(M.DUMMY_OBS, (K, 1))
    for j in range(max(1, K - 1)):
        r = rng.uniform(0.2, 1.0); rho = rng.uniform(r + mdl["radius"] + clear, 5.0); ph = rng.uniform(-np.pi, np.pi)
        obs[j] = [x0[0] + rho * np.cos(ph), x0[1] + rho * np.sin(ph), r, 0, 0, 0, 0]
    return x0, goal, obs


def compare(u, st, it, z, X, up, Gl, O, solve, N, K):
    """The bar of tests/test_mpccbf_gpu.py: same status on EVERY problem, |u0 - u0_oracle| <= 1e-6 and |z - z_oracle| <= 2e-5 on every
    optimal one (1e-4 on optima that stopped on the acceptable rule: both solvers are then within acceptable_tol of the same point).
    Iteration counts within 2, except on at most ONE problem of the batch (the bicycle's line search decides at the speed-clip kink on
    differences at the round-off level; such a problem reaches the same optimum a few iterations apart).
    Returns the number of optimal problems, of those with an active CBF row, and of certified infeasible ones."""
    B = X.shape[0]
    n_opt = n_act = n_inf = n_path = 0
    for i in range(B):
        uo, so, ito, info = solve(X[i], up[i], Gl[i], O[i])
        assert st[i] == so, f"status differs at problem {i}: {st[i]} vs {so}"
        if so == 0:
            tol_u, tol_z = (1e-6, 2e-5) if info["err"] <= 1e-6 else (1e-4, 1e-3)
            assert np.abs(u[i] - uo).max() <= tol_u * max(1.0, np.abs(uo).max()), i
            assert np.abs(z[i] - info["z"]).max() <= tol_z * max(1.0, np.abs(info["z"]).max()), i
            n_path += int(abs(int(it[i]) - ito) > 2)
            n_opt += 1
            n_act += int(np.min(info["g"][: N * K]) < 1e-4)
        elif so == 1:
            assert info["theta"] > 1e-6 and np.abs(u[i] - uo).max() <= 1e-5 * max(1.0, np.abs(uo).max()), i
            n_inf += 1
    assert n_path <= 1
    return n_opt, n_act, n_inf


@pytest.mark.parametrize("N,K", [(10, 5), (10, 8), (6, 3), (14, 4)])
def test_batch_matches_oracle(N, K):
    B = 24
    mdl = G.kb_model()
    rng = np.random.default_rng(100 * N + K)
    X = np.zeros((B, 4)); Gl = np.zeros((B, 2)); O = np.zeros((B, K, 7))
    for i in range(B):
        X[i], Gl[i], O[i] = draw(mdl, rng, K)
    up = np.zeros((B, 2)); up[B // 2:, 1] = rng.uniform(-0.2, 0.2, B - B // 2)
    ctl = sca.BatchedGnMPCCBF({"model": NAME}, io_dtype="f64", horizon=N)
    u, st, it, z = ctl.solve(t(X), t(up), t(Gl), t(O), want_z=True)
    torch.cuda.synchronize()
    solve = lambda x, u_, g_, o_: G.solve(mdl, x, u_, g_, o_, N=N, return_info=True)
    n_opt, n_act, n_inf = compare(u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.cpu().numpy(), X, up, Gl, O, solve, N, K)
    assert n_opt >= B // 3 and n_act >= 1                                  # cold starts among seven circles: the rest has no plan or jams


def test_reference_scene_closed_loop_drop_in():
    """The reference's KinematicBicycle2D scene, 120 control steps: the drop-in class against the oracle on the oracle's own closed
    loop (state and previous input taken from the oracle's run, so one differing step cannot hide the rest)."""
    spec = {"model": NAME, "a_max": 0.5, "radius": 0.5}
    mdl = G.kb_model({"a_max": 0.5, "radius": 0.5})
    K, N = 5, 10
    x = np.array([2.0, 2.0, math.pi / 2, 1.0]); up = np.zeros(2); wi = 1
    robot = sca.RobotHandle(x, dict(spec))
    ctl = sca.MPCCBF(robot, robot.robot_spec, num_obs=K)
    assert type(ctl).__name__ == "GnMPCCBF" and ctl.Q.shape == (4, 4) and ctl.R.tolist() == [0.5, 5000.0]
    assert ctl.cbf_param == {"alpha1": 0.1, "alpha2": 0.1}
    n_opt = 0
    for step in range(120):
        if np.linalg.norm(x[:2] - SCENE_WPS[wi]) < 0.3:
            wi += 1
        goal = SCENE_WPS[wi]
        near = np.argsort(np.linalg.norm(SCENE_OBS[:, :2] - x[:2], axis=1) - SCENE_OBS[:, 2])[:K]
        uo, so, ito, info = G.solve(mdl, x, up, goal, M.pad_obstacles(SCENE_OBS[near], K), N=N, return_info=True)
        ctl.u_prev = up.copy()
        ref = {"state_machine": "track", "u_ref": np.zeros((2, 1)), "goal": goal}
        u = ctl.solve_control_problem(x.reshape(-1, 1), ref, SCENE_OBS[near]).reshape(-1)
        names = {0: "optimal", 1: "infeasible", 2: "optimal_inaccurate"}
        assert ctl.solver_status == names[so], step
        if so == 0:
            assert np.abs(u - uo).max() <= (1e-6 if info["err"] <= 1e-6 else 1e-4) * max(1.0, np.abs(uo).max()), step
            n_opt += 1
        x = G.kb_S(x, uo, mdl["spec"], mdl["dt"]); up = uo
    assert n_opt >= 110 and x[1] > 6.0                                   # the bicycle made its way north past the first obstacles


def test_f32_arrays_shared_table_and_guards():
    B, K, N = 16, 4, 10
    mdl = G.kb_model()
    rng = np.random.default_rng(5)
    X = np.zeros((B, 4)); Gl = np.zeros((B, 2))
    shared = np.tile(M.DUMMY_OBS, (K, 1)); shared[0] = [7.0, 7.0, 1.0, 0, 0, 0, 0]; shared[1] = [4.0, 9.0, 0.5, 0, 0, 0, 0]
    for i in range(B):
        X[i], Gl[i], _ = draw(mdl, rng, K)
        while min(np.hypot(*(X[i, :2] - shared[j, :2])) - shared[j, 2] for j in range(2)) < 1.5:
            X[i], Gl[i], _ = draw(mdl, rng, K)
    X32, G32, S32 = X.astype(np.float32), Gl.astype(np.float32), shared.astype(np.float32)
    ctl = sca.BatchedGnMPCCBF({"model": NAME}, io_dtype="f32", horizon=N)
    u, st, it = ctl.solve(t(X32, torch.float32), t(np.zeros((B, 2)), torch.float32), t(G32, torch.float32), t(S32, torch.float32))
    u, st = u.double().cpu().numpy(), st.cpu().numpy()
    n = 0
    for i in range(0, B, 2):
        uo, so, _, info = G.solve(mdl, X32[i].astype(np.float64), np.zeros(2), G32[i].astype(np.float64), S32.astype(np.float64), N=N,
                                  return_info=True)
        assert st[i] == so
        if so == 0:
            assert np.abs(u[i] - uo).max() <= (2e-6 if info["err"] <= 1e-6 else 1e-4) * max(1.0, np.abs(uo).max())
            n += 1
    assert n >= 4
    with pytest.raises(NotImplementedError):
        sca.BatchedGnMPCCBF({"model": "Unicycle2D"})
    with pytest.raises(RuntimeError):
        sca.BatchedGnMPCCBF({"model": NAME, "rear_ax_dist": 0.0}).solve(t(X), t(np.zeros((B, 2))), t(Gl), t(shared))


def test_full_batch_properties():
    """4096 problems: deterministic launches, termination within the iteration limit, reported optima feasible (oracle's constraint
    functions on a strided sample: CBF rows, speed bounds, input box)."""
    B, K, N = 4096, 8, 10
    mdl = G.kb_model()
    rng = np.random.default_rng(21)
    X = np.zeros((B, 4)); Gl = np.zeros((B, 2)); O = np.zeros((B, K, 7))
    for i in range(B):
        X[i], Gl[i], O[i] = draw(mdl, rng, K)
    up = np.zeros((B, 2))
    ctl = sca.BatchedGnMPCCBF({"model": NAME}, io_dtype="f64", horizon=N)
    args = (t(X), t(up), t(Gl), t(O))
    u1, s1, i1, z1 = ctl.solve(*args, want_z=True)
    u2, s2, i2, z2 = ctl.solve(*args, want_z=True)
    torch.cuda.synchronize()
    assert torch.equal(u1, u2) and torch.equal(s1, s2) and torch.equal(i1, i2) and torch.equal(z1, z2)
    st, it, z = s1.cpu().numpy(), i1.cpu().numpy(), z1.cpu().numpy()
    assert it.max() <= 3000 and set(np.unique(st)) <= {0, 1, 2} and (st == 0).mean() > 0.5
    lo, hi = np.tile(mdl["u_lo"], N), np.tile(mdl["u_hi"], N)
    ok = st == 0
    assert np.all(z[ok] >= lo - 1e-9) and np.all(z[ok] <= hi + 1e-9)
    P = G.params(mdl, N)
    for i in np.flatnonzero(ok)[::97]:
        g = G.evaluate(X[i], z[i], up[i], Gl[i], O[i], P, level=0)["g"]
        assert g.min() >= -1e-6


# ---- KinematicBicycle2D_C3BF / _DPCBF: full-state discrete-time barriers (oracle/mpc_kb_state.py) ---------------------------------
from oracle import mpc_kb_state as S  # noqa: E402

STATE_MODELS = {"KinematicBicycle2D_C3BF": S.c3bf_model, "KinematicBicycle2D_DPCBF": S.dpcbf_model}


@pytest.mark.parametrize("name,N,K", [("KinematicBicycle2D_C3BF", 10, 5), ("KinematicBicycle2D_DPCBF", 10, 5), ("KinematicBicycle2D_C3BF", 6, 3)])
def test_state_barrier_batch_matches_oracle(name, N, K):
    B = 12
    mdl = STATE_MODELS[name]()
    rng = np.random.default_rng(7 * N + K + len(name))
    X = np.zeros((B, 4)); Gl = np.zeros((B, 2)); O = np.zeros((B, K, 7))
    for i in range(B):
        X[i], Gl[i], O[i] = draw(mdl, rng, K, clear=1.5)
        O[i, :, 3:5] = rng.uniform(-1, 1, (K, 2))                       # velocity columns: the MPC's barrier never reads them
    up = np.zeros((B, 2))
    ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    u, st, it, z = ctl.solve(t(X), t(up), t(Gl), t(O), want_z=True)
    torch.cuda.synchronize()
    solve = lambda x, u_, g_, o_: S.solve(mdl, x, u_, g_, o_, N=N, return_info=True)
    n_opt, n_act, n_inf = compare(u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.cpu().numpy(), X, up, Gl, O, solve, N, K)
    assert n_opt >= 1                       # cold starts in a field of cones: most of these draws start inside one


@pytest.mark.parametrize("name", list(STATE_MODELS))
def test_state_barrier_reference_scene_closed_loop_drop_in(name):
    """The reference's bicycle scene, 40 control steps, MPCCBF drop-in against the oracle on the oracle's own closed loop."""
    spec = {"model": name, "a_max": 0.5, "radius": 0.5}
    mdl = STATE_MODELS[name]({"a_max": 0.5, "radius": 0.5})
    K, N = 5, 10
    x = np.array([2.0, 2.0, math.pi / 2, 1.0]); up = np.zeros(2)
    robot = sca.RobotHandle(x, dict(spec))
    ctl = sca.MPCCBF(robot, robot.robot_spec, num_obs=K)
    assert type(ctl).__name__ == "GnMPCCBF" and ctl.cbf_param == {"alpha": 0.15} and ctl.R.tolist() == [0.5, 5000.0]
    goal = SCENE_WPS[1]
    n_opt = 0
    names = {0: "optimal", 1: "infeasible", 2: "optimal_inaccurate"}
    for step in range(40):
        near = np.argsort(np.linalg.norm(SCENE_OBS[:, :2] - x[:2], axis=1) - SCENE_OBS[:, 2])[:K]
        uo, so, ito, info = S.solve(mdl, x, up, goal, M.pad_obstacles(SCENE_OBS[near], K), N=N, return_info=True)
        ctl.u_prev = up.copy()
        ref = {"state_machine": "track", "u_ref": np.zeros((2, 1)), "goal": goal}
        u = ctl.solve_control_problem(x.reshape(-1, 1), ref, SCENE_OBS[near]).reshape(-1)
        assert ctl.solver_status == names[so], step
        if so == 0:
            assert np.abs(u - uo).max() <= (1e-6 if info["err"] <= 1e-6 else 1e-4) * max(1.0, np.abs(uo).max()), step
            n_opt += 1
        x = G.kb_S(x, uo, mdl["spec"], mdl["dt"]); up = uo
    assert n_opt >= 20


@pytest.mark.parametrize("name", list(STATE_MODELS))
def test_state_barrier_full_batch_properties(name):
    B, K, N = 2048, 8, 10
    mdl = STATE_MODELS[name]()
    rng = np.random.default_rng(33)
    X = np.zeros((B, 4)); Gl = np.zeros((B, 2)); O = np.zeros((B, K, 7))
    for i in range(B):
        X[i], Gl[i], O[i] = draw(mdl, rng, K, clear=1.5)
    up = np.zeros((B, 2))
    ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    args = (t(X), t(up), t(Gl), t(O))
    u1, s1, i1, z1 = ctl.solve(*args, want_z=True)
    u2, s2, i2, z2 = ctl.solve(*args, want_z=True)
    torch.cuda.synchronize()
    assert torch.equal(u1, u2) and torch.equal(s1, s2) and torch.equal(i1, i2) and torch.equal(z1, z2)
    st, it, z = s1.cpu().numpy(), i1.cpu().numpy(), z1.cpu().numpy()
    # seven cones within five metres of a cold start: about half of these draws have a feasible plan (47 % / 60 % measured)
    assert it.max() <= 3000 and set(np.unique(st)) <= {0, 1, 2} and (st == 0).mean() > 0.35
    lo, hi = np.tile(mdl["u_lo"], N), np.tile(mdl["u_hi"], N)
    ok = st == 0
    assert np.all(z[ok] >= lo - 1e-9) and np.all(z[ok] <= hi + 1e-9)
    P = S.params(mdl, N)
    for i in np.flatnonzero(ok)[::211]:
        g = S.evaluate(X[i], z[i], up[i], Gl[i], O[i], P, level=0)["g"]
        assert g.min() >= -1e-6
