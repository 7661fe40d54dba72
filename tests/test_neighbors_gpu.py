"""GPU tests: neighbour agents as moving obstacles (extension for BASELINE config 4) and the config-4 batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import c_oracle, cbf_qp as ocbf, robots as R  # noqa: E402
import safe_control_amd as sca  # noqa: E402
from safe_control_amd import sharding, workloads as W  # noqa: E402

DEV = "cuda:0"


def brute_force(X, lo, hi, K, radius):
    out = np.zeros((hi - lo, K, 7))
    out[:, :, 0:2] = 1000.0
    for i in range(lo, hi):
        d = (X[:, 0] - X[i, 0]) ** 2 + (X[:, 1] - X[i, 1]) ** 2
        d[i] = np.inf
        idx = np.argsort(d, kind="stable")[:K]
        for j, n in enumerate(idx):
            if np.isfinite(d[n]):
                out[i - lo, j] = [X[n, 0], X[n, 1], radius, X[n, 3] * np.cos(X[n, 2]), X[n, 3] * np.sin(X[n, 2]), 0, 0]
    return out


@pytest.mark.parametrize("B,K,dtype", [(3000, 16, torch.float64), (700, 8, torch.float32), (5, 8, torch.float64), (1000, 32, torch.float64)])
def test_neighbor_obstacles_match_brute_force(B, K, dtype):
    rng = np.random.default_rng(B)
    X = np.column_stack([rng.uniform(0, 40, B), rng.uniform(0, 40, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(0.2, 3.5, B)])
    tX = torch.tensor(X, dtype=dtype, device=DEV)
    obs = sharding.neighbor_obstacles(tX, B, K, 0.3).double().cpu().numpy()
    want = brute_force(tX.double().cpu().numpy(), 0, B, K, 0.3)
    np.testing.assert_allclose(obs, want, rtol=1e-6 if dtype == torch.float32 else 1e-12, atol=1e-6)
    # a sub-range (what one rank of a sharded run computes)
    import ctypes as C
    from safe_control_amd import _lib
    lo, hi = B // 3, B // 3 + min(300, B // 2)
    sub = torch.empty((hi - lo, K, 7), dtype=dtype, device=DEV)
    rc = _lib.load().sc_neighbor_obstacles_batch(0 if dtype == torch.float32 else 1, B, lo, hi - lo, K, 0.3, tX.data_ptr(),
                                                 sub.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    np.testing.assert_allclose(sub.double().cpu().numpy(), want[lo:hi], rtol=1e-6, atol=1e-6)


def test_config4_c3bf_16384_agents_16_moving_obstacles():
    """BASELINE config 4 batch (one GPU's view of it): 16384 KinematicBicycle2D C3BF agents, 16 moving circles each."""
    spec = {"model": "KinematicBicycle2D_C3BF", "a_max": 5.0, "radius": 0.3}
    B, K = 16384, 16
    X, goal, u_ref, obs = W.kb_c3bf_batch(B, K, seed=4, spec=spec)
    for io in ("f64", "f32"):
        ctl = sca.BatchedCBFQP(dict(spec), io_dtype=io, compute_dtype="f64")
        td = ctl.torch_dtype
        tX, tu, to = (torch.tensor(a, dtype=td, device=DEV) for a in (X, u_ref, obs))
        u, st, h = ctl.solve(tX, tu, to)
        ospec = R.default_spec(R.MODEL_KB_C3BF); ospec.update(a_max=5.0, radius=0.3)
        uo, so, ho = c_oracle.cbfqp_batch(R.MODEL_KB_C3BF, tX.double().cpu().numpy(), tu.double().cpu().numpy(),
                                          to.double().cpu().numpy(), ospec, ocbf.default_cbf_param(R.MODEL_KB_C3BF), n_threads=8)
        sg = st.cpu().numpy(); ug = u.double().cpu().numpy()
        assert (sg == so).mean() >= 0.9995
        ok = (sg == 0) & (so == 0)
        err = np.abs(ug[ok] - uo[ok]).max(axis=1)
        assert np.all(err <= (1e-7 if io == "f64" else 1e-5) * 5.0)
        assert 0.2 < (so == 0).mean() < 1.0


def test_agents_as_obstacles_pipeline_one_rank():
    """all-gather (no-op on one rank) -> neighbour rows -> C3BF CBF-QP: solutions equal the oracle on the same rows."""
    spec = {"model": "KinematicBicycle2D_C3BF", "a_max": 5.0, "radius": 0.3}
    B, K = 2048, 16
    rng = np.random.default_rng(1)
    X = np.column_stack([rng.uniform(0, 60, B), rng.uniform(0, 60, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(0.2, 3.5, B)])
    goal = rng.uniform(0, 60, (B, 2))
    from safe_control_amd.robots.spec import complete_robot_spec
    u_ref = W.nominal_input_kb(X, goal, complete_robot_spec(dict(spec)))
    tX = torch.tensor(X, dtype=torch.float64, device=DEV); tu = torch.tensor(u_ref, dtype=torch.float64, device=DEV)
    obs = sharding.neighbor_obstacles(tX, B, K, 0.3)
    ctl = sca.BatchedCBFQP(dict(spec), io_dtype="f64", compute_dtype="f64")
    u, st, h = ctl.solve(tX, tu, obs)
    ospec = R.default_spec(R.MODEL_KB_C3BF); ospec.update(a_max=5.0, radius=0.3)
    uo, so, ho = c_oracle.cbfqp_batch(R.MODEL_KB_C3BF, X, u_ref, obs.cpu().numpy(), ospec, ocbf.default_cbf_param(R.MODEL_KB_C3BF))
    assert np.array_equal(st.cpu().numpy(), so) or (st.cpu().numpy() == so).mean() > 0.999
    ok = (so == 0) & (st.cpu().numpy() == 0)
    assert np.abs(u.cpu().numpy()[ok] - uo[ok]).max() < 1e-6


@pytest.mark.parametrize("B,lo,n_local,K,dtype", [(20000, 0, 20000, 16, torch.float32), (20000, 5000, 2500, 16, torch.float32),
                                                   (3000, 100, 777, 8, torch.float64), (900, 0, 900, 32, torch.float64),
                                                   (300, 0, 300, 16, torch.float32)])
def test_candidate_split_search_equals_the_single_scan(B, lo, n_local, K, dtype):
    """sc_neighbor_obstacles_batch_ws (the uniform-grid cell list since round 6; rounds 3 - 5: candidate slices + merge) against
    sc_neighbor_obstacles_batch (one scan over every agent): bit for bit, with exact ties (agents on a grid) and duplicate positions."""
    from safe_control_amd import _lib
    rng = np.random.default_rng(B + K)
    X = np.column_stack([rng.integers(0, 60, B).astype(np.float64), rng.integers(0, 60, B).astype(np.float64),
                         rng.uniform(-np.pi, np.pi, B), rng.uniform(0.2, 3.5, B)])      # integer grid: many equal distances
    tX = torch.tensor(X, dtype=dtype, device=DEV)
    io = 0 if dtype == torch.float32 else 1
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.empty((n_local, K, 7), dtype=dtype, device=DEV)
    b = torch.empty((n_local, K, 7), dtype=dtype, device=DEV)
    assert lib.sc_neighbor_obstacles_batch(io, B, lo, n_local, K, 0.3, tX.data_ptr(), a.data_ptr(), stream) == 0
    nbytes = int(lib.sc_neighbor_workspace_bytes(io, B, n_local, K))
    assert nbytes > 0
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=DEV)
    assert lib.sc_neighbor_obstacles_batch_ws(io, B, lo, n_local, K, 0.3, tX.data_ptr(), b.data_ptr(), ws.data_ptr(), nbytes, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    if B <= 3000:                                                 # ties resolve by index, like a stable sort
        want = brute_force(tX.double().cpu().numpy(), lo, lo + n_local, K, 0.3)
        np.testing.assert_allclose(b.double().cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    assert lib.sc_neighbor_obstacles_batch_ws(io, B, lo, n_local, K, 0.3, tX.data_ptr(), b.data_ptr(), ws.data_ptr(), nbytes - 1, stream) != 0


@pytest.mark.parametrize("case", ["clusters", "line", "far_outlier", "coincident", "fewer_than_K", "nan_and_inf", "config4"])
def test_cell_list_on_awkward_fleets_equals_the_single_scan(case):
    """The cell list's stop test (K-th best inside the covered block) and its binning on fleets a uniform grid does not like: dense clusters
    far apart, every agent on one line, one agent 10 km away (the grid hits its 256 x 256 limit and the cells are large), everybody on the
    same spot, fewer agents than K, non-finite positions (never anybody's neighbour, as in the scan), and BASELINE configs[3]'s 16384 agents."""
    from safe_control_amd import _lib
    rng = np.random.default_rng(7)
    K, dtype = 16, torch.float32
    if case == "clusters":
        B = 6000
        c = rng.uniform(0, 500, (12, 2))
        P = c[rng.integers(0, 12, B)] + rng.normal(0, 0.3, (B, 2))
    elif case == "line":
        B = 4000
        P = np.column_stack([rng.uniform(0, 100, B), np.full(B, 3.0)])
    elif case == "far_outlier":
        B = 5000
        P = rng.uniform(0, 14, (B, 2)); P[17] = [1.0e4, -1.0e4]
    elif case == "coincident":
        B = 1500
        P = np.tile([[2.5, -1.0]], (B, 1)); P[::7] += rng.uniform(-1, 1, (len(P[::7]), 2))
    elif case == "fewer_than_K":
        B = 9
        P = rng.uniform(0, 5, (B, 2))
    elif case == "nan_and_inf":
        B = 2000
        P = rng.uniform(0, 30, (B, 2)); P[5, 0] = np.nan; P[99] = [np.inf, 3.0]; P[1000, 1] = -np.inf
    else:
        B = 16384
        P = W.kb_c3bf_batch(B, K, seed=0)[0][:, :2]
    X = np.column_stack([P, rng.uniform(-np.pi, np.pi, B), rng.uniform(0.2, 3.5, B)])
    tX = torch.tensor(X, dtype=dtype, device=DEV)
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.empty((B, K, 7), dtype=dtype, device=DEV)
    b = torch.empty((B, K, 7), dtype=dtype, device=DEV)
    assert lib.sc_neighbor_obstacles_batch(0, B, 0, B, K, 0.3, tX.data_ptr(), a.data_ptr(), stream) == 0
    nbytes = int(lib.sc_neighbor_workspace_bytes(0, B, B, K))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=DEV)
    for _ in range(2):                                              # (the workspace is reused call after call: no state may survive in it)
        assert lib.sc_neighbor_obstacles_batch_ws(0, B, 0, B, K, 0.3, tX.data_ptr(), b.data_ptr(), ws.data_ptr(), nbytes, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(a.nan_to_num(nan=-7.0, posinf=-8.0, neginf=-9.0), b.nan_to_num(nan=-7.0, posinf=-8.0, neginf=-9.0))
