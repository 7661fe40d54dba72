"""GPU: kernel 13 (csrc/mpc_du_ms.hip) instantiated for DoubleIntegrator2D -- the reference's MPC-CBF NLP for that robot as do-mpc poses it
(position_control/mpc_cbf.py:28-30,56-59,135-141,196-200; robots/double_integrator2D.py:79-107,222-226: the DT barrier goes through robot.step,
which rescales the velocity to norm v_max) under IPOPT's algorithm -- against oracle/ms_ipopt.py with di_model() in the kernel's profile:
same status and same iteration count problem by problem, |u0 - u0_oracle| <= 1e-8, on draws with and without a feasible point and with the
velocity rescaling active inside the rows."""
import os
from multiprocessing import Pool

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402
from safe_control_amd.robots.spec import complete_robot_spec  # noqa: E402
from oracle import ms_ipopt as MS  # noqa: E402

DEV = "cuda:0"
SPEC = {"model": "DoubleIntegrator2D"}


def t(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def di_oracle_model():
    sp = complete_robot_spec(dict(SPEC))
    keys = MS.di_model()["spec"].keys()
    return MS.di_model({k: v for k, v in sp.items() if k in keys})


def _one(args):
    x, up, g, ob, N = args
    os.environ["OMP_NUM_THREADS"] = "1"
    u, st, it, info = MS.solve(di_oracle_model(), x, up, g, ob, N=N, return_info=True, opts=dict(MS.KERNEL_PROFILE))
    return u, st, it, np.concatenate([info["X"].reshape(-1), info["U"].reshape(-1)])


def oracle_many(X, up, goal, obs, N=None):
    with Pool(min(32, os.cpu_count() or 4)) as p:
        return p.map(_one, [(X[i], up[i], goal[i], obs[i], N) for i in range(len(X))], chunksize=2)


def compare(u, st, it, plan, res, n_off):
    so, ito = np.array([r[1] for r in res]), np.array([r[2] for r in res])
    assert np.array_equal(st, so), np.flatnonzero(st != so)[:10]
    off = it != ito
    assert off.sum() <= n_off and np.abs(it - ito).max() <= 2, (int(off.sum()), int(np.abs(it - ito).max()))
    du = np.array([np.abs(u[i] - r[0]).max() for i, r in enumerate(res)])
    assert du[~off].max() <= 1e-8 and du.max() <= 1e-6, (du[~off].max(), du.max())
    if plan is not None:
        dp = np.array([np.abs(plan[i] - r[3]).max() for i, r in enumerate(res)])
        assert dp[(so == 0) & ~off].max() <= 1e-6, dp[(so == 0) & ~off].max()
    return so, ito


def test_bench_draws_against_the_oracle():
    n = 384
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("di", 4096, 8, seed=0))
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f64")
    u, st, it, plan = (a.cpu().numpy() for a in ctl.solve(t(X), t(up), t(goal), t(obs), want_plan=True))
    so, ito = compare(u, st, it, plan, oracle_many(X, up, goal, obs), n_off=6)
    assert 0.02 <= (so == 1).mean() <= 0.2 and (so == 2).mean() <= 0.01
    print(f"di ms kernel: optimal {np.mean(so == 0):.4f}, infeasible {np.mean(so == 1):.4f}, iterations mean {ito.mean():.1f} max {ito.max()}")


def test_fast_starts_rescaled_velocity_inside_the_rows():
    """Speeds up to 1.8 x v_max and a random last input: robot.step's rescaling (and its curvature) is active on most stages."""
    n = 192
    X, up, goal, obs = (a[:n].copy() for a in W.mpc_family_batch("di", 4096, 8, seed=1))
    rng = np.random.default_rng(5)
    X[:, 2:4] = rng.uniform(-1.3, 1.3, (n, 2)); up = rng.uniform(-1.0, 1.0, (n, 2))
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f64")
    u, st, it = (a.cpu().numpy() for a in ctl.solve(t(X), t(up), t(goal), t(obs)))
    so, ito = compare(u, st, it, None, oracle_many(X, up, goal, obs), n_off=4)
    assert (np.hypot(X[:, 2], X[:, 3]) > 1.0).mean() >= 0.4 and (so == 1).sum() >= 8
    print(f"di ms kernel, fast starts: optimal {np.mean(so == 0):.4f}, infeasible {np.mean(so == 1):.4f}, iterations max {ito.max()}")


def test_full_batch_f32_storage_box_and_the_condensed_kernel_where_both_are_optimal():
    X, up, goal, obs = W.mpc_family_batch("di", 4096, 8, seed=0)
    f = lambda a: t(a.astype(np.float32), torch.float32)                          # noqa: E731
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f32")
    u1, s1, i1 = ctl.solve(f(X), f(up), f(goal), f(obs))
    u2, s2, i2 = ctl.solve(f(X), f(up), f(goal), f(obs))
    assert torch.equal(u1, u2) and torch.equal(s1, s2) and torch.equal(i1, i2)
    assert (u1.abs() <= 1.0 + 1e-6).all()
    opt = (s1 == 0).double().mean().item()
    assert 0.88 <= opt <= 0.97 and (s1 == 2).double().mean().item() <= 0.005, opt
    uc, sc, ic = sca.BatchedGnMPCCBF(SPEC, io_dtype="f32").solve(f(X), f(up), f(goal), f(obs))[:3]
    both = (s1 == 0) & (sc == 0)
    same = ((u1 - uc).abs().amax(dim=1) <= 1e-4)[both].double().mean().item()
    assert both.double().mean().item() >= 0.85 and same >= 0.99, (both.double().mean().item(), same)
    print(f"di ms kernel, 4096: optimal {opt:.4f}, iterations mean {i1.double().mean().item():.1f} max {int(i1.max())}; same optimum as the condensed kernel on {same:.4f}")


def test_drop_in_class_and_batched_loop_use_the_kernel():
    from safe_control_amd.position_control.mpc_cbf import MPCCBF

    class Robot:
        dt, robot_radius = 0.05, 0.25
    ctl = MPCCBF(Robot(), {"model": "DoubleIntegrator2D"}, num_obs=8)
    assert ctl._ms is not None
    X, up, goal, obs = (a[:4] for a in W.mpc_family_batch("di", 64, 8, seed=2))
    ref = dict(goal=goal[0], state_machine="track", u_ref=np.zeros((2, 1)))
    u = ctl.solve_control_problem(X[0].reshape(-1, 1), ref, obs[0])
    uo = MS.solve(di_oracle_model(), X[0], np.zeros(2), goal[0], obs[0], opts=dict(MS.KERNEL_PROFILE))[0]
    assert np.abs(u.reshape(-1) - uo).max() <= 1e-8
    loop = sca.BatchedTrackingController(np.hstack([X, np.zeros((4, 1))]), {"model": "DoubleIntegrator2D"}, controller_type={"pos": "mpc_cbf"}, enable_rotation=False,
                                         obs=obs[0], device=DEV)
    assert loop.mpc_ms is not None and loop.mpc_ms.model == "DoubleIntegrator2D"
