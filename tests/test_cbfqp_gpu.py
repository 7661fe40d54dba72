"""GPU parity tests for the fused CBF-QP kernels (run on the MI355X box: pytest -m gpu).

Everything goes through the C-ABI (ctypes -> libsafe_control_hip.so); the
oracle (oracle/) is only the checker.

Stated tolerances (DESIGN.md "Parity"):
  * f64 arithmetic (f64 or f32 storage): |u - u_oracle| <= 1e-7 * max(1, |u|_inf) on f64
    storage, one f32 ulp (2e-6 relative) on f32 storage; h likewise; status equal except for
    problems whose feasibility margin is below 1e-6 (set aside and counted).
  * f32 arithmetic: |u - u_oracle| <= 1e-4 * max(1, bound) per SURVEY 8c on >= 99.5 % of the
    optimal cases, h within 1e-5 * max(1, |h|); status equal except margin < 1e-4.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle, cbf_qp as ocbf, qp as oqp, robots as R  # noqa: E402
import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402

DEV = "cuda:0"

MODEL_NAME = {R.MODEL_DU: "DynamicUnicycle2D", R.MODEL_KB: "KinematicBicycle2D",
              R.MODEL_KB_C3BF: "KinematicBicycle2D_C3BF", R.MODEL_KB_DPCBF: "KinematicBicycle2D_DPCBF",
              R.MODEL_SI: "SingleIntegrator2D", R.MODEL_DI: "DoubleIntegrator2D"}


def du_spec():
    return {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}


def kb_spec(name):
    return {"model": name, "a_max": 5.0, "radius": 0.3}


def oracle_spec(model, spec):
    s = R.default_spec(model)
    s.update({k: v for k, v in spec.items() if k != "model"})
    return s


def margins(model, X, obs, spec, cbf_param, n_obs=None, cbf_mode="cbf"):
    """Feasibility margin of every problem (oracle rows + Chebyshev LP)."""
    out = np.empty(X.shape[0])
    lo, hi = ocbf.input_bounds(model, spec)
    Gb, cb = oqp.box_rows(lo, hi)
    K = obs.shape[-2]
    for i in range(X.shape[0]):
        o = obs if obs.ndim == 2 else obs[i]
        k = K if n_obs is None else int(n_obs[i])
        A, b, _ = ocbf.assemble_rows(model, X[i], list(o[:k]), spec, cbf_param, K, 0.05, cbf_mode)
        out[i] = oqp.feasibility_margin(np.vstack([A, Gb]), np.concatenate([b, cb]))
    return out


def run_gpu(spec, X, u_ref, obs, io="f32", comp="f64", n_obs=None):
    ctl = sca.BatchedCBFQP(dict(spec), dt=0.05, io_dtype=io, compute_dtype=comp)
    td = ctl.torch_dtype
    tX = torch.tensor(X, dtype=td, device=DEV)
    tu = torch.tensor(u_ref, dtype=td, device=DEV)
    to = torch.tensor(obs, dtype=td, device=DEV)
    tn = None if n_obs is None else torch.tensor(n_obs, dtype=torch.int32, device=DEV)
    u, st, h = ctl.solve(tX, tu, to, tn)
    torch.cuda.synchronize()
    # what the kernel actually saw (storage-rounded inputs), for the oracle
    seen = (tX.double().cpu().numpy(), tu.double().cpu().numpy(), to.double().cpu().numpy())
    return u.double().cpu().numpy(), st.cpu().numpy(), h.double().cpu().numpy(), seen


def compare(model, spec, X, u_ref, obs, io, comp, n_obs=None, cbf_mode="cbf", frac_ok=1.0):
    ug, sg, hg, (Xs, us, os_) = run_gpu(spec, X, u_ref, obs, io, comp, n_obs)
    ospec = oracle_spec(model, spec)
    cp = ocbf.default_cbf_param(model)
    uo, so, ho = c_oracle.cbfqp_batch(model, Xs, us, os_, ospec, cp, 0.05, cbf_mode, n_obs)
    mtol = 1e-6 if comp == "f64" else 1e-4
    diff = np.nonzero(sg != so)[0]
    if len(diff):
        mg = margins(model, Xs[diff], os_ if os_.ndim == 2 else os_[diff], ospec, cp,
                     None if n_obs is None else np.asarray(n_obs)[diff], cbf_mode)
        assert np.all(np.abs(mg) < mtol), f"status mismatch away from the margin: {diff[:8]}, margins {mg[:8]}"
    assert len(diff) <= max(2, 0.01 * len(sg))
    ok = (sg == 0) & (so == 0)
    assert np.all(np.isnan(ug[sg != 0]))
    bound = max(1.0, float(np.max(np.abs(ocbf.input_bounds(model, ospec)[1]))))
    err = np.max(np.abs(ug[ok] - uo[ok]), axis=1)
    if comp == "f64":
        tol_u = (1e-7 if io == "f64" else 2e-6) * bound
        tol_h = (1e-9 if io == "f64" else 2e-6)
    else:
        tol_u, tol_h = 1e-4 * bound, 1e-5
    good = err <= tol_u
    assert good.mean() >= frac_ok, f"u*: {100 * good.mean():.3f}% within {tol_u}, worst {err.max()}"
    herr = np.abs(hg - ho) / np.maximum(1.0, np.abs(ho))
    assert np.all(herr <= tol_h), f"h: worst {herr.max()}"
    return dict(n=len(sg), infeasible=int((so != 0).sum()), excluded=len(diff), worst_u=float(err.max()))


# ------------------------------------------------------------------ config 2
@pytest.mark.parametrize("io,comp,frac", [("f64", "f64", 1.0), ("f32", "f64", 1.0), ("f32", "f32", 0.995)])
def test_config2_du_4096x8(io, comp, frac):
    """BASELINE config 2: 4096 DynamicUnicycle2D agents, 8 circles each, seed 0."""
    X, goal, u_ref, obs = W.du_cbfqp_batch(4096, 8, seed=0)
    r = compare(R.MODEL_DU, du_spec(), X, u_ref, obs, io, comp, frac_ok=frac)
    assert r["infeasible"] > 0          # the seeded batch contains infeasible agents: they must be reported


@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 127, 1000])
def test_ragged_batch_sizes(B):
    X, goal, u_ref, obs = W.du_cbfqp_batch(B, 8, seed=B)
    compare(R.MODEL_DU, du_spec(), X, u_ref, obs, "f32", "f64")


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 7, 9, 10, 12, 13, 16, 17, 24, 25, 32])
def test_every_row_count(K):
    """K hits every kernel instantiation, both the K == KMAX wide-read path and the generic one."""
    X, goal, u_ref, obs = W.du_cbfqp_batch(333, K, seed=100 + K)
    compare(R.MODEL_DU, du_spec(), X, u_ref, obs, "f32", "f64")
    compare(R.MODEL_DU, du_spec(), X, u_ref, obs, "f64", "f64")


def test_ragged_obstacle_counts():
    """n_obs[i] < K: unused rows are the reference's zero rows (cbf_qp.py:110-111)."""
    X, goal, u_ref, obs = W.du_cbfqp_batch(500, 10, seed=5)
    rng = np.random.default_rng(5)
    n_obs = rng.integers(0, 11, 500).astype(np.int32)
    obs2 = obs.copy()
    for i in range(500):
        obs2[i, n_obs[i]:] = 1e30        # garbage beyond n_obs must be ignored
    compare(R.MODEL_DU, du_spec(), X, u_ref, obs2, "f32", "f64", n_obs=n_obs)
    ug, sg, hg, _ = run_gpu(du_spec(), X, u_ref, obs2, "f32", "f64", n_obs)
    for i in range(500):
        assert np.all(hg[i, n_obs[i]:] == 0)


def test_shared_obstacle_table():
    X, goal, u_ref, _ = W.du_cbfqp_batch(777, 8, seed=9)
    rng = np.random.default_rng(9)
    table = np.zeros((6, 7))
    table[:, 0:2] = rng.uniform(0, 14, (6, 2)); table[:, 2] = rng.uniform(0.2, 0.6, 6)
    compare(R.MODEL_DU, du_spec(), X, u_ref, table, "f32", "f64")


def test_superellipsoid_and_mixed_obstacles():
    rng = np.random.default_rng(11)
    B, K = 600, 6
    X, goal, u_ref, obs = W.du_cbfqp_batch(B, K, seed=11)
    for i in range(B):
        for k in range(K):
            if rng.random() < 0.5:
                a, b = rng.uniform(0.3, 1.2, 2)
                rho = rng.uniform(max(a, b) + 0.55, 4.5); phi = rng.uniform(-np.pi, np.pi)
                obs[i, k] = [X[i, 0] + rho * np.cos(phi), X[i, 1] + rho * np.sin(phi), a, b,
                             float(rng.choice([4, 6, 10])), rng.uniform(-np.pi, np.pi), 1.0]
    # superellipsoid barrier values reach 1e4..1e8: compare u with f64 arithmetic only
    compare(R.MODEL_DU, du_spec(), X, u_ref, obs, "f64", "f64")


def test_hard_mode():
    X, goal, u_ref, obs = W.du_cbfqp_batch(400, 8, seed=21)
    spec = du_spec(); spec["cbf_mode"] = "hard"
    compare(R.MODEL_DU, spec, X, u_ref, obs, "f64", "f64", cbf_mode="hard")


@pytest.mark.parametrize("model", [R.MODEL_KB, R.MODEL_KB_C3BF, R.MODEL_KB_DPCBF])
@pytest.mark.parametrize("K", [10, 16])
def test_kinematic_bicycle_family(model, K):
    """Config-4 family: moving circles, state-dependent g, rel-deg-1 C3BF / DPCBF rows."""
    spec = kb_spec(MODEL_NAME[model])
    X, goal, u_ref, obs = W.kb_c3bf_batch(1024, K, seed=31, spec=spec)
    compare(model, spec, X, u_ref, obs, "f64", "f64")
    compare(model, spec, X, u_ref, obs, "f32", "f64")


@pytest.mark.parametrize("model,K", [(R.MODEL_DU, 8), (R.MODEL_DU, 5), (R.MODEL_DU, 3), (R.MODEL_KB, 8),
                                     (R.MODEL_KB_C3BF, 8), (R.MODEL_KB_DPCBF, 6)])
@pytest.mark.parametrize("io,comp", [("f32", "f64"), ("f64", "f64"), ("f32", "f32")])
def test_lane_per_qp_kernel_above_coop_threshold(model, K, io, comp):
    """Batches above 32768 agents run the lane-per-QP register kernel: same oracle, same tolerances."""
    B = 36864 + 37
    if model == R.MODEL_DU:
        spec = du_spec()
        X, goal, u_ref, obs = W.du_cbfqp_batch(B, K, seed=77 + K)
        if K == 5:                                   # mix in superellipsoids and ragged counts
            rng = np.random.default_rng(5)
            idx = rng.choice(B, 3000, replace=False)
            for i in idx:
                a, b = rng.uniform(0.3, 1.2, 2)
                rho, phi = rng.uniform(max(a, b) + 0.55, 4.5), rng.uniform(-np.pi, np.pi)
                obs[i, 1] = [X[i, 0] + rho * np.cos(phi), X[i, 1] + rho * np.sin(phi), a, b,
                             float(rng.choice([4, 6])), rng.uniform(-np.pi, np.pi), 1.0]
    else:
        spec = kb_spec(MODEL_NAME[model])
        X, goal, u_ref, obs = W.kb_c3bf_batch(B, K, seed=78 + K, spec=spec)
    n_obs = None
    if K == 3:
        n_obs = np.random.default_rng(3).integers(0, 4, B).astype(np.int32)
    if comp == "f32" and model == R.MODEL_DU and K == 5:
        pytest.skip("superellipsoid rows reach 1e4..1e8: f64 arithmetic only (SURVEY 7)")
    compare(model, spec, X, u_ref, obs, io, comp, n_obs=n_obs, frac_ok=1.0 if comp == "f64" else 0.99)


def test_bad_obstacle_flag_and_nan_inputs():
    _bad_flag_case(130)
    _bad_flag_case(33000)                  # same through the lane-per-QP kernel


def _bad_flag_case(B):
    X, goal, u_ref, obs = W.du_cbfqp_batch(B, 8, seed=3)
    obs[5, 2, 6] = 2.0                     # invalid flag
    X[7, 0] = np.nan                       # NaN state -> non-finite rows -> not optimal
    u_ref[9, 1] = np.inf
    ug, sg, hg, _ = run_gpu(du_spec(), X, u_ref, obs, "f32", "f64")
    assert sg[5] == 3 and np.all(np.isnan(ug[5]))
    assert sg[7] == 1 and sg[9] == 1
    other = np.ones(B, bool); other[[5, 7, 9]] = False
    uo, so, ho = c_oracle.cbfqp_batch(R.MODEL_DU, X[other].astype(np.float32).astype(np.float64),
                                      u_ref[other].astype(np.float32).astype(np.float64),
                                      obs[other].astype(np.float32).astype(np.float64),
                                      oracle_spec(R.MODEL_DU, du_spec()), ocbf.default_cbf_param(R.MODEL_DU))
    assert np.array_equal(sg[other], so)


def test_degenerate_geometry():
    """Duplicate obstacles (parallel duplicate rows), an agent standing still (v = 0: second column of every row
    vanishes), obstacle rows collinear with the agent, an agent exactly on an obstacle centre, huge coordinates."""
    rng = np.random.default_rng(99)
    for B in (512, 33000):                                  # cooperative kernel and lane-per-QP kernel
        X, goal, u_ref, obs = W.du_cbfqp_batch(B, 8, seed=B + 1)
        obs[0::7, 3] = obs[0::7, 1]                          # exact duplicates
        obs[1::7, 5] = obs[1::7, 4]; obs[1::7, 6] = obs[1::7, 4]
        X[2::7, 3] = 0.0                                     # standing still
        th = X[3::7, 2]
        for k in range(4):                                   # collinear obstacles straight ahead
            obs[3::7, k, 0] = X[3::7, 0] + (1.5 + k) * np.cos(th)
            obs[3::7, k, 1] = X[3::7, 1] + (1.5 + k) * np.sin(th)
            obs[3::7, k, 2] = 0.3
        obs[4::7, 2, 0:2] = X[4::7, 0:2]                     # agent on the obstacle centre (h < 0)
        X[5::7, 0:2] += 1e6; obs[5::7, :, 0:2] += 1e6        # far from the origin
        u_ref = np.where(rng.random((B, 1)) < 0.3, u_ref * 5.0, u_ref)
        for io, comp in (("f64", "f64"), ("f32", "f64")):
            if io == "f32":
                keep = np.ones(B, bool); keep[5::7] = False   # 1e6 + O(1) is not representable in f32 storage
                compare(R.MODEL_DU, du_spec(), X[keep], u_ref[keep], obs[keep], io, comp)
            else:
                compare(R.MODEL_DU, du_spec(), X, u_ref, obs, io, comp)


@pytest.mark.parametrize("K", [8, 13, 16])
def test_crowded_scenes_cooperative_kernels(K):
    """Many rows violated at once (obstacles 0.35 - 1.5 m ahead of a moving agent, large reference inputs): optima at vertices of two
    rows, rows that only bind after another row moved the point, and infeasible crowds.  The 8- and 16-lanes-per-agent kernels take
    every candidate line at once (sc_group.hpp: coop_solve_all8 / 16); every agent of the batch is compared with the enumerating C
    oracle."""
    rng = np.random.default_rng(1000 + K)
    B = 4096
    X, goal, u_ref, obs = W.du_cbfqp_batch(B, K, seed=K)
    X[:, 3] = rng.uniform(0.3, 1.0, B)
    for j in range(K):
        rho = rng.uniform(0.35, 1.5, B) + 0.25; ang = X[:, 2] + rng.uniform(-1.2, 1.2, B)
        obs[:, j, 0] = X[:, 0] + rho * np.cos(ang); obs[:, j, 1] = X[:, 1] + rho * np.sin(ang)
        obs[:, j, 2] = rng.uniform(0.05, 0.3, B); obs[:, j, 3:] = 0.0
    u_ref = rng.uniform(-1.0, 1.0, (B, 2)) * np.array([1.0, 0.5]) * rng.choice([0.3, 1.0, 4.0], (B, 1))
    for io, comp in (("f64", "f64"), ("f32", "f64")):
        r = compare(R.MODEL_DU, du_spec(), X, u_ref, obs, io, comp)
        assert 0.02 * B < r["infeasible"] < 0.9 * B                       # both kinds of outcome are in the batch


# ------------------------------------------------------------------ golden fixtures
def test_golden_cases_through_dropin_class(golden_dir):
    """tests/golden/cbfqp_cases.npz through the reference-shaped CBFQP class (host-pointer C-ABI)."""
    g = np.load(os.path.join(golden_dir, "cbfqp_cases.npz"))
    groups = {"du_circle": ("DynamicUnicycle2D", None), "du_circle_hard": ("DynamicUnicycle2D", "hard"),
              "du_superellipsoid": ("DynamicUnicycle2D", None), "du_mixed_trunc": ("DynamicUnicycle2D", None),
              "du_overlap": ("DynamicUnicycle2D", None), "kb_circle": ("KinematicBicycle2D", None),
              "c3bf": ("KinematicBicycle2D_C3BF", None), "c3bf_k16": ("KinematicBicycle2D_C3BF", None),
              "dpcbf": ("KinematicBicycle2D_DPCBF", None)}
    for gname, (model, mode) in groups.items():
        num_obs = int(g[f"{gname}/meta"][0])
        spec = du_spec() if model == "DynamicUnicycle2D" else kb_spec(model)
        if mode:
            spec["cbf_mode"] = mode
        robot = sca.RobotHandle(np.zeros(4), spec, dt=0.05)
        ctl = sca.CBFQP(robot, spec, num_obs=num_obs)
        Xs, us, obs, ks = g[f"{gname}/X"], g[f"{gname}/u_ref"], g[f"{gname}/obs"], g[f"{gname}/k"]
        ustar, st = g[f"{gname}/u_star_oracle"], g[f"{gname}/status_oracle"]
        for i in range(len(Xs)):
            robot.X = Xs[i].reshape(-1, 1)
            u = ctl.solve_control_problem(robot.X, {"u_ref": us[i].reshape(2, 1)}, list(obs[i][: int(ks[i])]))
            if st[i] == 0:
                assert ctl.status == "optimal", (gname, i)
                np.testing.assert_allclose(u.reshape(-1), ustar[i], rtol=1e-7, atol=1e-7)
                assert u.shape == (2, 1)
            else:
                assert ctl.status != "optimal" and u is None, (gname, i)
    # obs_list None -> u_ref unclipped
    robot = sca.RobotHandle(np.zeros(4), du_spec())
    ctl = sca.CBFQP(robot, du_spec(), num_obs=8)
    u = ctl.solve_control_problem(robot.X, {"u_ref": g["none/u_ref"].reshape(2, 1)}, None)
    np.testing.assert_array_equal(u.reshape(-1), g["none/u"])
    assert ctl.status == "optimal"


# ------------------------------------------------------------------ integrator models
@pytest.mark.parametrize("name,model", [("SingleIntegrator2D", R.MODEL_SI), ("DoubleIntegrator2D", R.MODEL_DI)])
def test_integrator_models(golden_dir, name, model):
    """Reference-generated cases through the drop-in class, then a 40k-agent batch (both kernels) vs the C oracle."""
    g = np.load(os.path.join(golden_dir, "integrators.npz"))
    G = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(name + "/")}
    spec = {"model": name, "radius": 0.25, "v_max": 1.0}
    if model == R.MODEL_DI:
        spec["a_max"] = 1.5
    robot = sca.RobotHandle(np.zeros(4), dict(spec), dt=0.05)
    ctl = sca.CBFQP(robot, dict(spec), num_obs=6)
    nx = 2 if model == R.MODEL_SI else 4
    for i in range(len(G["X"])):
        robot.X = G["X"][i][:nx].reshape(-1, 1)
        u = ctl.solve_control_problem(robot.X, {"u_ref": G["u_ref"][i].reshape(2, 1)}, list(G["obs"][i][: int(G["k"][i])]))
        if G["status_oracle"][i] == 0:
            assert ctl.status == "optimal"
            np.testing.assert_allclose(u.reshape(-1), G["u_star_oracle"][i], rtol=1e-7, atol=1e-7)
        else:
            assert u is None and ctl.status != "optimal"
    # batches: superellipsoids mixed in, speeds in the (theta, v) slots for the double integrator
    for B in (3000, 40000):
        X, goal, u_ref, obs = W.du_cbfqp_batch(B, 6, seed=B)
        rng = np.random.default_rng(B)
        X[:, 2:4] = rng.uniform(-0.8, 0.8, (B, 2)) if model == R.MODEL_DI else 0.0
        u_ref = rng.uniform(-2, 2, (B, 2))
        for i in rng.choice(B, B // 10, replace=False):
            a, b = rng.uniform(0.3, 1.2, 2)
            rho, phi = rng.uniform(max(a, b) + 0.55, 4.5), rng.uniform(-np.pi, np.pi)
            obs[i, 2] = [X[i, 0] + rho * np.cos(phi), X[i, 1] + rho * np.sin(phi), a, b, float(rng.choice([4, 6])),
                         rng.uniform(-np.pi, np.pi), 1.0]
        compare(model, spec, X, u_ref, obs, "f64", "f64")


def test_unicycle2d_model(golden_dir):
    """Unicycle2D (3 states, inputs v / omega, rel-deg-1 barrier with the sigma(s) term): reference-generated cases
    through the drop-in class, then 3000 / 40000-agent batches (both kernels, far padding rows included) vs the C oracle."""
    g = np.load(os.path.join(golden_dir, "unicycle2d.npz"))
    G = {k.split("/", 1)[1]: g[k] for k in g.files}
    spec = {"model": "Unicycle2D", "radius": 0.25, "v_max": 1.0, "w_max": 0.5}
    robot = sca.RobotHandle(np.zeros(4), dict(spec), dt=0.05)
    ctl = sca.CBFQP(robot, dict(spec), num_obs=6)
    assert ctl.cbf_param == {"alpha": 1.0}
    for i in range(len(G["X"])):
        robot.X = G["X"][i][:3].reshape(-1, 1)
        u = ctl.solve_control_problem(robot.X, {"u_ref": G["u_ref"][i].reshape(2, 1)}, list(G["obs"][i][: int(G["k"][i])]))
        if G["status_oracle"][i] == 0:
            assert ctl.status == "optimal"
            np.testing.assert_allclose(u.reshape(-1), G["u_star_oracle"][i], rtol=1e-7, atol=1e-7)
            np.testing.assert_allclose(ctl.h, G["h"][i][: int(G["k"][i])], rtol=1e-9, atol=1e-9)
        else:
            assert u is None and ctl.status != "optimal"
    for B in (3000, 40000):
        X, goal, u_ref, obs = W.du_cbfqp_batch(B, 6, seed=B + 1)
        rng = np.random.default_rng(B)
        X[:, 3] = 0.0
        u_ref = np.column_stack([rng.uniform(-1.5, 1.5, B), rng.uniform(-1, 1, B)])
        obs[rng.choice(B, B // 8, replace=False), 5] = [1000.0, 1000.0, 0, 0, 0, 0, 0]      # update_tvp-style padding rows
        compare(R.MODEL_UNI, spec, X, u_ref, obs, "f64", "f64")
        compare(R.MODEL_UNI, spec, X, u_ref, obs, "f32", "f64")


def test_quad2d_model(golden_dir):
    """Quad2D (6 states, thrust box [f_min, f_max], both inputs enter identically => all rows parallel):
    reference-generated cases through the drop-in class, then 3000 / 40000-agent batches vs the C oracle."""
    g = np.load(os.path.join(golden_dir, "quad2d.npz"))
    G = {k.split("/", 1)[1]: g[k] for k in g.files}
    spec = {"model": "Quad2D", "f_min": 3.0, "f_max": 10.0, "radius": 0.25}
    robot = sca.RobotHandle(np.zeros(6), dict(spec), dt=0.05)
    ctl = sca.CBFQP(robot, dict(spec), num_obs=6)
    for i in range(len(G["X"])):
        robot.X = G["X"][i].reshape(-1, 1)
        u = ctl.solve_control_problem(robot.X, {"u_ref": G["u_ref"][i].reshape(2, 1)}, list(G["obs"][i][: int(G["k"][i])]))
        if G["status_oracle"][i] == 0:
            assert ctl.status == "optimal", i
            np.testing.assert_allclose(u.reshape(-1), G["u_star_oracle"][i], rtol=1e-7, atol=1e-7)
        else:
            assert u is None and ctl.status != "optimal", i
    ospec = R.default_spec(R.MODEL_QUAD2D); ospec.update(f_min=3.0, f_max=10.0)
    cp = ocbf.default_cbf_param(R.MODEL_QUAD2D)
    for B in (3000, 40000):
        rng = np.random.default_rng(B)
        Xd, goal, _, obs = W.du_cbfqp_batch(B, 6, seed=B + 5)
        X = np.column_stack([Xd[:, 0], Xd[:, 1], rng.uniform(-0.6, 0.6, B), rng.uniform(-1.5, 1.5, B),
                             rng.uniform(-1.5, 1.5, B), rng.uniform(-1, 1, B)])
        u_ref = rng.uniform(2.0, 11.0, (B, 2))
        for io in ("f64", "f32"):
            bc = sca.BatchedCBFQP(dict(spec), io_dtype=io, compute_dtype="f64")
            td = bc.torch_dtype
            tX, tu, to = (torch.tensor(a, dtype=td, device=DEV) for a in (X, u_ref, obs))
            u, st, h = bc.solve(tX, tu, to)
            uo, so, ho = c_oracle.cbfqp_batch(R.MODEL_QUAD2D, tX.double().cpu().numpy(), tu.double().cpu().numpy(),
                                              to.double().cpu().numpy(), ospec, cp, n_threads=4)
            sg = st.cpu().numpy()
            assert (sg == so).mean() >= 0.999
            ok = (sg == 0) & (so == 0)
            assert 0.3 < ok.mean() < 1.0
            err = np.abs(u.double().cpu().numpy()[ok] - uo[ok]).max()
            assert err <= (1e-6 if io == "f64" else 5e-5), err
            assert np.abs(h.double().cpu().numpy() - ho).max() <= (1e-9 if io == "f64" else 1e-4)


# ------------------------------------------------------------------ closed loop through the plugin surface
@pytest.mark.parametrize("tag,model_name,steps", [("du14", "DynamicUnicycle2D", 700), ("du3", "DynamicUnicycle2D", 400),
                                                  ("c3bf_dyn", "KinematicBicycle2D_C3BF", 160),
                                                  ("dpcbf_dyn", "KinematicBicycle2D_DPCBF", 198)])
def test_closed_loop_control_step_with_dropin_controller(golden_dir, tag, model_name, steps):
    """BASELINE config 1 (examples/test_tracking.py --model du --algo cbf_qp) and the dynamic_env runs:
    the control_step data flow (oracle/tracking.py, pinned to the reference's trajectories) with the HIP-backed
    CBFQP class behind the boundary must reproduce the reference's closed-loop X and U."""
    from oracle import tracking
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    model = {v: k for k, v in MODEL_NAME.items()}[model_name]
    spec = du_spec() if model == R.MODEL_DU else kb_spec(model_name)
    wps = g[f"{tag}/waypoints"]
    robot = sca.RobotHandle(np.append(wps[0], 1.0), dict(spec), dt=0.05)
    ctl = sca.CBFQP(robot, dict(spec), num_obs=10)

    def solve_fn(X, control_ref, obs):
        robot.X = np.asarray(X, dtype=np.float64).reshape(-1, 1)
        u = ctl.solve_control_problem(robot.X, {"u_ref": np.asarray(control_ref["u_ref"]).reshape(2, 1)}, obs)
        return (None if u is None else u.reshape(-1)), (0 if ctl.status == "optimal" else 1)

    dyn = tag.endswith("_dyn")
    obs0 = g[f"{tag}/obs0"] if dyn else g[f"{tag}/obs"]
    t = tracking.TrackingOracle(model, np.append(wps[0], 1.0), {k: v for k, v in spec.items() if k != "model"}, dt=0.05,
                                obs=obs0, num_constraints=10, dyn_obs=dyn, solve_fn=solve_fn)
    t.set_waypoints(wps)
    Xg, Ug, retg = g[f"{tag}/X"], g[f"{tag}/U"], g[f"{tag}/ret"]
    for k in range(min(steps, len(retg))):
        ret = t.control_step()
        assert ret == retg[k], (k, ret, retg[k])
        if ret != 0:
            break
        np.testing.assert_allclose(t.u_pos, Ug[k], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t.X, Xg[k + 1], rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ full-size properties
def _rows_du_circle(X, obs, R_, a1=1.5, a2=1.5):
    """fp64 torch restatement of the DU circle rows, only to verify constraint satisfaction at scale."""
    x, y, th, v = X[:, 0:1], X[:, 1:2], X[:, 2:3], X[:, 3:4]
    c, s = torch.cos(th), torch.sin(th)
    ex, ey = x - obs[:, :, 0], y - obs[:, :, 1]
    d = obs[:, :, 2] + R_
    h = ex * ex + ey * ey - 1.01 * d * d
    hdot = 2 * (ex * v * c + ey * v * s)
    n0 = 2 * (ex * c + ey * s)
    n1 = 2 * v * (-ex * s + ey * c)
    b = 2 * v * v * (c * c + s * s) + (a1 + a2) * hdot + a1 * a2 * h
    return n0, n1, b


@pytest.mark.parametrize("comp", ["f64", "f32"])
def test_full_size_properties(comp):
    """B = 2^20 agents x 8 obstacles: feasibility of every reported optimum, idempotence,
    obstacle-order invariance, batch-position independence."""
    B, K = 1 << 20, 8
    X, goal, u_ref, obs = W.du_cbfqp_batch(B, K, seed=1)
    ctl = sca.BatchedCBFQP(du_spec(), io_dtype="f32", compute_dtype=comp)
    tX = torch.tensor(X, dtype=torch.float32, device=DEV)
    tu = torch.tensor(u_ref, dtype=torch.float32, device=DEV)
    to = torch.tensor(obs, dtype=torch.float32, device=DEV)
    u, st, h = ctl.solve(tX, tu, to)
    ok = st == 0
    assert 0.90 < ok.double().mean().item() < 1.0
    # (1) every optimum satisfies box and CBF rows
    n0, n1, b = _rows_du_circle(tX.double(), to.double(), 0.25)
    ud = u.double()
    slack = n0 * ud[:, 0:1] + n1 * ud[:, 1:2] + b
    scale = 1 + n0.abs() * ud[:, 0:1].abs() + n1.abs() * ud[:, 1:2].abs() + b.abs()
    tol = 5e-6 if comp == "f64" else 2e-4
    assert (slack[ok] >= -tol * scale[ok]).all()
    assert (ud[ok, 0].abs() <= 1.0 + 1e-6).all() and (ud[ok, 1].abs() <= 0.5 + 1e-6).all()
    # h output equals the barrier value
    hd = (tX[:, 0:1].double() - to[:, :, 0].double()) ** 2 + (tX[:, 1:2].double() - to[:, :, 1].double()) ** 2 \
        - 1.01 * (to[:, :, 2].double() + 0.25) ** 2
    assert ((h.double() - hd).abs() <= 2e-6 * (1 + hd.abs())).all()
    # (2) projection is idempotent: u_ref := u* gives u* back
    u2, st2, _ = ctl.solve(tX, torch.where(ok[:, None], u, tu), to)
    assert (st2[ok] == 0).all()
    assert ((u2[ok] - u[ok]).abs().max().item()) <= (1e-6 if comp == "f64" else 5e-4)
    # (3) the answer does not depend on the obstacle order (the kernel walks rows incrementally)
    perm = torch.randperm(K, device=DEV)
    u3, st3, _ = ctl.solve(tX, tu, to[:, perm].contiguous())
    same = (st3 == st)
    assert same.double().mean().item() > 0.9999
    both = ok & (st3 == 0)
    d3 = (u3[both] - u[both]).abs().max(dim=1).values
    assert (d3 <= (1e-6 if comp == "f64" else 1e-3)).double().mean().item() >= (1.0 if comp == "f64" else 0.999)
    # (4) an agent's result does not depend on where it sits in the batch (same kernel: bitwise)
    lo, n4 = 12345, 40003
    u4, st4, _ = ctl.solve(tX[lo:lo + n4].contiguous(), tu[lo:lo + n4].contiguous(), to[lo:lo + n4].contiguous())
    assert torch.equal(st4, st[lo:lo + n4])
    assert torch.equal(torch.nan_to_num(u4, nan=7.0), torch.nan_to_num(u[lo:lo + n4], nan=7.0))
    # ... nor on which kernel serves the batch size: <= 32768 agents go to the 8-lanes-per-agent kernel
    u5, st5, h5 = ctl.solve(tX[lo:lo + 4099].contiguous(), tu[lo:lo + 4099].contiguous(), to[lo:lo + 4099].contiguous())
    assert (st5 == st[lo:lo + 4099]).double().mean().item() >= (1.0 if comp == "f64" else 0.999)
    b5 = (st5 == 0) & (st[lo:lo + 4099] == 0)
    d5 = (u5[b5] - u[lo:lo + 4099][b5]).abs().max(dim=1).values
    assert (d5 <= (1e-6 if comp == "f64" else 1e-4)).double().mean().item() >= (1.0 if comp == "f64" else 0.999)
    assert torch.equal(h5, h[lo:lo + 4099])
    # oracle on a strided sample of the big batch
    idx = np.arange(0, B, 509)
    uo, so, ho = c_oracle.cbfqp_batch(R.MODEL_DU, tX[idx].double().cpu().numpy(), tu[idx].double().cpu().numpy(),
                                      to[idx].double().cpu().numpy(), oracle_spec(R.MODEL_DU, du_spec()),
                                      ocbf.default_cbf_param(R.MODEL_DU))
    sg = st[idx].cpu().numpy(); ug = u[idx].double().cpu().numpy()
    agree = sg == so
    assert agree.mean() > 0.998
    okk = agree & (so == 0)
    e = np.abs(ug[okk] - uo[okk]).max(axis=1)
    assert np.mean(e <= (2e-6 if comp == "f64" else 1e-4)) >= (1.0 if comp == "f64" else 0.995)


@pytest.mark.parametrize("ragged", [False, True])
def test_two_launch_path_with_superellipsoids_mixed_into_waves(ragged, tmp_path):
    """DynamicUnicycle2D, f64 arithmetic, B >= 2^18: a circles-only fast launch, then the generic body for the waves that saw an obstacle
    flag != 0 (csrc/cbf_qp_kernel.hpp: cbfqp_reg_kernel<.., PASS>; between the two launches those agents carry the internal status
    SC_STATUS_PENDING = -1, which must never reach the caller).  Superellipsoids sit in a third of the waves -- some alone in their
    wave, some beside circles -- with and without ragged obstacle counts.  Held to (i) the single-launch kernel (SC_CBFQP_TWO_PASS=0,
    run in a child process: the switch is read once per process) bit for bit and (ii) the C oracle on a sample."""
    import os
    import subprocess
    import sys
    B, K = (1 << 18) + 77, 8
    rng = np.random.default_rng(5)
    X, goal, u_ref, obs = W.du_cbfqp_batch(B, K, seed=5)
    wave = np.arange(B) // 64
    pick = (rng.random(B) < 0.35) & (wave % 3 == 0)                      # a third of the waves hold superellipsoids
    for i in np.flatnonzero(pick):
        k = int(rng.integers(K))
        a, b = rng.uniform(0.3, 1.2, 2)
        rho = rng.uniform(max(a, b) + 0.55, 4.5); phi = rng.uniform(-np.pi, np.pi)
        obs[i, k] = [X[i, 0] + rho * np.cos(phi), X[i, 1] + rho * np.sin(phi), a, b, float(rng.choice([4, 6, 10])), rng.uniform(-np.pi, np.pi), 1.0]
    n_obs = rng.integers(0, K + 1, B).astype(np.int32) if ragged else None
    ug, sg, hg, (Xs, us, os_) = run_gpu(du_spec(), X, u_ref, obs, "f32", "f64", n_obs)
    assert set(np.unique(sg)) <= {0, 1}                                  # no pending agent left behind
    inp = tmp_path / "in.npz"; outp = tmp_path / "out.npz"
    np.savez(inp, X=Xs.astype(np.float32), u=us.astype(np.float32), o=os_.astype(np.float32), n=(n_obs if ragged else np.zeros(0, np.int32)))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import safe_control_amd as sca\n"
            "d = np.load(%r); ctl = sca.BatchedCBFQP(%r, dt=0.05, io_dtype='f32', compute_dtype='f64')\n"
            "t = lambda a: torch.tensor(a, device='cuda:0')\n"
            "n = t(d['n']) if d['n'].size else None\n"
            "u, st, h = ctl.solve(t(d['X']), t(d['u']), t(d['o']), n)\n"
            "np.savez(%r, u=u.cpu().numpy(), st=st.cpu().numpy(), h=h.cpu().numpy())\n") % (os.path.dirname(os.path.dirname(__file__)), str(inp), du_spec(), str(outp))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SC_CBFQP_TWO_PASS="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    one = np.load(outp)
    assert np.array_equal(one["st"], sg)
    assert np.array_equal(np.nan_to_num(one["u"].astype(np.float64), nan=7.0), np.nan_to_num(ug, nan=7.0))
    assert np.array_equal(one["h"].astype(np.float64), hg)
    idx = np.concatenate([np.flatnonzero(pick)[::97], np.arange(0, B, 1019)])
    uo, so, ho = c_oracle.cbfqp_batch(R.MODEL_DU, Xs[idx], us[idx], os_[idx], oracle_spec(R.MODEL_DU, du_spec()), ocbf.default_cbf_param(R.MODEL_DU),
                                      0.05, "cbf", None if n_obs is None else n_obs[idx])
    agree = sg[idx] == so
    assert agree.mean() > 0.995
    ok = agree & (so == 0)
    assert np.abs(ug[idx][ok] - uo[ok]).max() <= 2e-6


def test_argument_errors():
    ctl = sca.BatchedCBFQP(du_spec())
    tX = torch.zeros((4, 4), device=DEV); tu = torch.zeros((4, 2), device=DEV)
    with pytest.raises(sca.HipLibraryError):
        ctl.solve(tX, tu, torch.zeros((4, 40, 7), device=DEV))     # K above SC_CBFQP_MAX_OBS
    with pytest.raises(ValueError):
        ctl.solve(tX.cpu(), tu, torch.zeros((4, 8, 7), device=DEV))
    with pytest.raises(ValueError):
        sca.BatchedCBFQP({"model": "Quad3D"})
