"""GPU: Backup-CBF QP kernel (csrc/backup_cbf.hip) through the C-ABI against the reference-executed goldens
(tests/golden/backup_cbf.npz) and the numpy oracle (oracle/backup_cbf.py).  Tolerances (float64 storage): kept rows
1e-7 relative (forward differences amplify rounding by 1 / eps = 1e5; the kernel evaluates the differenced functions
without FMA contraction, so in practice they agree far closer), same row count, same QP status and fallback, |u - u_ref|
<= 1e-6, |h_min| <= 1e-9."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402
from oracle import backup_cbf as O  # noqa: E402

DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "backup_cbf.npz"))


def t(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_golden_cases(tag):
    dt, hor = float(G[f"{tag}_dt"]), float(G[f"{tag}_horizon"])
    ctl = sca.BatchedBackupCBF(dt=dt, backup_horizon=hor)
    X, BX = G[f"{tag}_X"], G[f"{tag}_bullet_x"]
    u, st, using, hmin, n_rows, rows = ctl.solve(t(X), t(G[f"{tag}_u_nom"]), t(BX), want_rows=True)
    torch.cuda.synchronize()
    u, st, using, hmin, n_rows, rows = (a.cpu().numpy() for a in (u, st, using, hmin, n_rows, rows))
    assert np.array_equal(n_rows, G[f"{tag}_n_rows"])
    assert np.array_equal(st, G[f"{tag}_qp_status"])
    assert np.array_equal(using.astype(bool), G[f"{tag}_using_backup"])
    assert np.abs(hmin - G[f"{tag}_h_min"]).max() <= 1e-9
    for i in range(len(X)):
        ref = G[f"{tag}_rows"][i][: n_rows[i]]
        assert np.abs(rows[i][: n_rows[i]] - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), i
    assert np.abs(u - G[f"{tag}_u"]).max() <= 1e-6


def draw_states(B, seed):
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.uniform(2, 58, B), rng.uniform(-1.4, 1.4, B), rng.uniform(-0.5, 1.5, B), rng.uniform(-0.5, 0.5, B)])
    pocket = rng.uniform(size=B) < 0.3                                  # a third of the agents somewhere over / in the pocket
    X[pocket, 0] = rng.uniform(25.7, 34.3, pocket.sum())
    X[pocket, 1] = rng.uniform(-1.0, 5.3, pocket.sum())
    bx = X[:, 0] - rng.uniform(-10, 25, B)
    return X, bx


@pytest.mark.parametrize("dt,hor", [(0.1, 12.0), (0.05, 2.0)])
def test_batch_against_oracle(dt, hor):
    B = 192
    X, bx = draw_states(B, seed=int(hor * 10))
    ctl = sca.BatchedBackupCBF(dt=dt, backup_horizon=hor)
    u, st, using, hmin = ctl.solve(t(X), None, t(bx))                   # built-in nominal controller of the example
    torch.cuda.synchronize()
    u, st, using, hmin = (a.cpu().numpy() for a in (u, st, using, hmin))
    env, spec = O.default_env(), O.default_spec()
    kinds = set()
    n_bad = 0
    for i in range(B):
        uo, info = O.solve(X[i], O.nominal_control(X[i], spec), bx[i], dt, hor, env, spec, return_info=True)
        assert abs(hmin[i] - info["h_min"]) <= 1e-9
        if st[i] != info["qp_status"]:                                  # a QP whose feasibility margin is within the tolerance
            n_bad += 1
            continue
        assert bool(using[i]) == info["using_backup"], i
        assert np.abs(u[i] - uo).max() <= 1e-6, (i, u[i], uo)
        kinds.add((int(st[i]), bool(using[i])))
    assert n_bad <= 2
    assert len(kinds) >= 3


def test_f32_storage_and_shared_bullet():
    B = 256
    X, _ = draw_states(B, seed=3)
    ctl = sca.BatchedBackupCBF(io_dtype="f32")
    X32 = t(X, torch.float32)
    bx = t([12.5], torch.float32)
    u, st, using, hmin = ctl.solve(X32, None, bx)
    u2, st2, using2, hmin2 = ctl.solve(X32, None, bx.expand(B).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(u, u2) and torch.equal(st, st2) and torch.equal(hmin, hmin2)
    ctl64 = sca.BatchedBackupCBF()
    u64, st64, _, hmin64 = ctl64.solve(X32.double(), None, bx.double())
    same = st == st64
    assert same.double().mean() > 0.97
    assert (u.double() - u64)[same].abs().max() <= 1e-5 and (hmin.double() - hmin64).abs().max() <= 1e-5


def test_closed_loop_reproduces_the_example():
    """examples/evade/test_evade.py --algo backupcbf, first 260 steps (golden loop), as agent 0 of a batch whose other
    agents start elsewhere; the fused rollout and per-step launches give the same trajectory."""
    T = 260
    B = 64
    X0, bx0 = draw_states(B, seed=11)
    X0[0], bx0[0] = G["loop_X"][0], G["loop_bullet_x"][0]
    ctl = sca.BatchedBackupCBF()
    X, bx = t(X0), t(bx0)
    ret = torch.zeros(B, dtype=torch.int32, device=DEV); rs = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    traj = []
    for k in range(0, T, 20):
        for j in range(20):                                             # one step per launch: records the trajectory
            traj.append(X[0].cpu().numpy().copy())
            ctl.rollout(X, bx, ret, rs, 1, step_offset=k + j)
    traj = np.array(traj)
    assert np.abs(traj - G["loop_X"][:T]).max() <= 1e-6
    assert int(ret[0].item()) == 0
    Xf, bxf = t(X0), t(bx0)
    retf = torch.zeros(B, dtype=torch.int32, device=DEV); rsf = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    ctl.rollout(Xf, bxf, retf, rsf, T)
    torch.cuda.synchronize()
    assert torch.equal(Xf, X) and torch.equal(retf, ret) and torch.equal(rsf, rs) and torch.equal(bxf, bx)
    assert abs(float(bx[0].item()) - (G["loop_bullet_x"][T] if T < len(G["loop_bullet_x"]) else 0)) <= 1e-9


def test_dropin_class_on_the_evade_scenario():
    class Env:                                                          # the attributes of envs/evade_env.py the controller reads
        pass
    e = O.default_env()
    env = Env()
    for k, v in e.items():
        setattr(env, k, v)
    env.get_pocket_bounds = lambda: {"x_min": e["pocket_x_min"], "x_max": e["pocket_x_max"], "y_min": e["pocket_y_min"], "y_max": e["pocket_y_max"]}

    class Backup:
        safe_center, safe_bounds, Kp, Kd = (30.0, 4.0), env.get_pocket_bounds(), 2.0, 2.0

    spec = {"model": "DoubleIntegrator2D", "radius": 0.5, "a_max": 2.0, "v_max": 1.5, "safety_margin": 0.5}
    sh = sca.BackupCBF(None, spec, dt=0.1, backup_horizon=12.0)
    sh.set_backup_controller(Backup())
    sh.set_environment(env)
    for i in (0, 2, 3, 7):
        bx = float(G["a_bullet_x"][i])
        sh.set_moving_obstacles(lambda tt=0.0, bx=bx: {"x": bx + e["bullet_length"] / 6 + e["bullet_speed"] * tt, "y": 0.0, "vx": 3.0, "vy": 0.0,
                                                         "length": 4.0, "width": 4.0, "active": True})
        sh.set_nominal_trajectory(None, np.tile(G["a_u_nom"][i], (3, 1)))
        u = sh.solve_control_problem(G["a_X"][i].reshape(-1, 1))
        assert u.shape == (2, 1) and np.abs(u.flatten() - G["a_u"][i]).max() <= 1e-6
        assert sh.is_using_backup() == bool(G["a_using_backup"][i])
        assert abs(sh.get_status()["h_min"] - float(G["a_h_min"][i])) <= 1e-9


def test_argument_validation():
    import ctypes as C
    from safe_control_amd import _lib
    from safe_control_amd.position_control import backup_cbf_qp as BK
    lib = _lib.load()
    p = BK.make_params(BK.default_evade_env(), {"radius": 0.5}, 0.1, 12.0, _lib.DTYPE_F64)
    X = torch.zeros((1, 4), dtype=torch.float64, device=DEV)
    a = lambda x: x.data_ptr()
    u = torch.zeros((1, 2), dtype=torch.float64, device=DEV); s = torch.zeros(1, dtype=torch.int32, device=DEV)
    b = torch.zeros(1, dtype=torch.float64, device=DEV)
    assert lib.sc_backupcbf_solve_batch(C.byref(p), 1, a(X), None, None, a(u), a(s), None, None, None, None, None) == 1
    p.n_steps = 200
    assert lib.sc_backupcbf_solve_batch(C.byref(p), 1, a(X), None, a(b), a(u), a(s), None, None, None, None, None) == 2
    p.n_steps = 120
    assert lib.sc_backupcbf_rollout_batch(C.byref(p), 1, 1, 0, a(X), a(b), a(u), a(s), None, None, None, None, None) == 1
