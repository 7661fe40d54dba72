"""GPU tests of the optimal-decay CBF-QP kernel against oracle/od_cbf_qp.py (pytest -m gpu)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import od_cbf_qp as OD, robots as R  # noqa: E402
import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402

DEV = "cuda:0"
NAMES = {R.MODEL_DU: "DynamicUnicycle2D", R.MODEL_KB: "KinematicBicycle2D", R.MODEL_KB_C3BF: "KinematicBicycle2D_C3BF",
         R.MODEL_KB_DPCBF: "KinematicBicycle2D_DPCBF", R.MODEL_QUAD2D: "Quad2D"}


def setup(model, B, seed):
    if model == R.MODEL_DU:
        X, goal, ur, obs = W.du_cbfqp_batch(B, 1, seed=seed)
        spec = {"model": NAMES[model], "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
    elif model == R.MODEL_QUAD2D:
        # optimal_decay_cbf_qp.py:38-45,105-115: six states, thrusts in [f_min, f_max], references around hover
        Xd, goal, _, obs = W.du_cbfqp_batch(B, 1, seed=seed)
        rng = np.random.default_rng(seed + 5)
        X = np.zeros((B, 6)); X[:, :2] = Xd[:, :2]; X[:, 2] = rng.uniform(-0.4, 0.4, B); X[:, 3:5] = rng.uniform(-1.5, 1.5, (B, 2))
        spec = {"model": NAMES[model], "f_min": 3.0, "f_max": 10.0, "radius": 0.25}
        ur = rng.uniform(2.0, 11.0, (B, 2))
    else:
        spec = {"model": NAMES[model], "a_max": 5.0, "radius": 0.3}
        X, goal, ur, obs = W.kb_c3bf_batch(B, 1, seed=seed, spec=spec)
    ospec = R.default_spec(model); ospec.update({k: v for k, v in spec.items() if k != "model"})
    rng = np.random.default_rng(seed)
    ur = ur * rng.choice([1.0, 1.0, 4.0], (B, 1))            # some references far outside the box
    return X, ur, obs[:, 0], spec, ospec


@pytest.mark.parametrize("model", [R.MODEL_DU, R.MODEL_KB, R.MODEL_KB_C3BF, R.MODEL_KB_DPCBF, R.MODEL_QUAD2D])
@pytest.mark.parametrize("io,comp", [("f64", "f64"), ("f32", "f64")])
def test_against_oracle(model, io, comp):
    B = 700
    X, ur, obs, spec, ospec = setup(model, B, seed=11)
    has = np.ones(B, dtype=np.int32); has[::17] = 0
    ctl = sca.BatchedOptimalDecayCBFQP(dict(spec), io_dtype=io, compute_dtype=comp)
    td = ctl.torch_dtype
    tX, tu, to = (torch.tensor(a, dtype=td, device=DEV) for a in (X, ur, obs))
    u, w, st, h = ctl.solve(tX, tu, to, torch.tensor(has, device=DEV))
    u, w, st, h = u.double().cpu().numpy(), w.double().cpu().numpy(), st.cpu().numpy(), h.double().cpu().numpy()
    Xs, us, os_ = tX.double().cpu().numpy(), tu.double().cpu().numpy(), to.double().cpu().numpy()
    tol = 1e-7 if io == "f64" else 3e-6
    for i in range(B):
        r = OD.solve(model, Xs[i], us[i], os_[i] if has[i] else None, ospec)
        assert st[i] == r["status"], i
        if r["status"] == 0:
            scale = max(1.0, float(np.abs(r["u"]).max()))
            assert np.abs(u[i] - r["u"]).max() <= tol * scale, (i, u[i], r["u"])
            assert np.abs(w[i] - r["omega"]).max() <= tol * max(1.0, float(np.abs(r["omega"]).max())), (i, w[i], r["omega"])
            assert abs(h[i] - (r["h"] if has[i] else 0.0)) <= tol * max(1.0, abs(r["h"]))


def test_dropin_class_and_stale_reference_input_shape():
    X, ur, obs, spec, ospec = setup(R.MODEL_DU, 20, seed=3)
    robot = sca.RobotHandle(X[0], dict(spec), dt=0.05)
    ctl = sca.OptimalDecayCBFQP(robot, dict(spec))
    assert ctl.cbf_param["p_sb1"] == 10 ** 4 and ctl.cbf_param["alpha1"] == 0.5
    for i in range(20):
        robot.X = X[i].reshape(-1, 1)
        many = np.vstack([obs[i], obs[(i + 1) % 20]])            # what control_step passes: (k,7) -> nearest = row 0
        u = ctl.solve_control_problem(robot.X, {"u_ref": ur[i].reshape(2, 1)}, many)
        r = OD.solve(R.MODEL_DU, X[i], ur[i], obs[i], ospec)
        assert ctl.status == "optimal" and u.shape == (2, 1)
        np.testing.assert_allclose(u.reshape(-1), r["u"], atol=1e-7)
        np.testing.assert_allclose(ctl.omega, r["omega"], atol=1e-7)
    u = ctl.solve_control_problem(robot.X, {"u_ref": np.array([[3.0], [-2.0]])}, None)
    np.testing.assert_allclose(u.reshape(-1), [1.0, -0.5])
    with pytest.raises(sca.position_control.optimal_decay_cbf_qp.NotCompatibleError):
        sca.position_control.optimal_decay_cbf_qp.default_od_param("SingleIntegrator2D")


def test_quad2d_dropin_class():
    """optimal_decay_cbf_qp.py:38-45 accepts Quad2D; its row is the rel-deg-2 one of :105-115 with the thrust box."""
    X, ur, obs, spec, ospec = setup(R.MODEL_QUAD2D, 12, seed=4)
    robot = sca.RobotHandle(X[0], dict(spec), dt=0.05)
    ctl = sca.OptimalDecayCBFQP(robot, dict(spec))
    assert ctl.cbf_param == dict(alpha1=0.5, alpha2=0.5, omega1=1.0, p_sb1=10 ** 4, omega2=1.0, p_sb2=10 ** 4)
    n_act = 0
    for i in range(12):
        robot.X = X[i].reshape(-1, 1)
        u = ctl.solve_control_problem(robot.X, {"u_ref": ur[i].reshape(2, 1)}, obs[i])
        r = OD.solve(R.MODEL_QUAD2D, X[i], ur[i], obs[i], ospec)
        assert ctl.status == "optimal"
        np.testing.assert_allclose(u.reshape(-1), r["u"], atol=1e-7)
        np.testing.assert_allclose(ctl.omega, r["omega"], atol=1e-7)
        assert np.all(u >= 3.0 - 1e-9) and np.all(u <= 10.0 + 1e-9)
        n_act += int(np.abs(ctl.omega - 1.0).max() > 1e-9)
    assert n_act >= 1
