"""GPU: kernel 13 (csrc/mpc_du_ms.hip) instantiated for SingleIntegrator2D -- the reference's MPC-CBF NLP for that robot as do-mpc poses it
(position_control/mpc_cbf.py:19-21,49-51,135-141,183-187,312-315; robots/single_integrator2D.py:45-66,148-190: two states, inputs (vx, vy), one-step
rows) under IPOPT's algorithm -- against oracle/ms_ipopt.py with si_model() in the kernel's profile: same status, same iteration count,
|u0 - u0_oracle| <= 1e-8.  The kernel holds the robot as four states, two of them idle; circles only."""
import os
from multiprocessing import Pool

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402
from oracle import ms_ipopt as MS  # noqa: E402

DEV = "cuda:0"
SPEC = {"model": "SingleIntegrator2D"}


def t(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def _one(args):
    x, up, g, ob = args
    os.environ["OMP_NUM_THREADS"] = "1"
    return MS.solve(MS.si_model(), x, up, g, ob, opts=dict(MS.KERNEL_PROFILE))


def oracle_many(X, up, goal, obs):
    with Pool(min(32, os.cpu_count() or 4)) as p:
        return p.map(_one, [(X[i], up[i], goal[i], obs[i]) for i in range(len(X))], chunksize=2)


def compare(u, st, it, res, n_off):
    so, ito = np.array([r[1] for r in res]), np.array([r[2] for r in res])
    assert np.array_equal(st, so), np.flatnonzero(st != so)[:10]
    off = it != ito
    assert off.sum() <= n_off and np.abs(it - ito).max() <= 2, (int(off.sum()), int(np.abs(it - ito).max()))
    du = np.array([np.abs(u[i] - r[0]).max() for i, r in enumerate(res)])
    assert du[~off].max() <= 1e-8 and du.max() <= 1e-6, (du[~off].max(), du.max())
    return so, ito


def test_bench_draws_and_a_crowded_scene_against_the_oracle():
    n = 256
    X, up, goal, obs = (a[:n].copy() for a in W.mpc_family_batch("si", 4096, 8, seed=0))
    assert X.shape[1] == 2 and not (obs[..., 6] >= 0.5).any()
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f64")
    u, st, it = (a.cpu().numpy() for a in ctl.solve(t(X), t(up), t(goal), t(obs)))           # two-column state rows, as the reference's
    so, ito = compare(u, st, it, oracle_many(X, up, goal, obs), n_off=4)
    rng = np.random.default_rng(7)
    m = 128
    ob2 = obs[:m].copy()
    ang = rng.uniform(0, 2 * np.pi, (m, 3)); rad = rng.uniform(0.45, 0.9, (m, 3))
    ob2[:, :3, 0] = X[:m, None, 0] + rad * np.cos(ang); ob2[:, :3, 1] = X[:m, None, 1] + rad * np.sin(ang); ob2[:, :3, 2] = 0.2
    up2 = rng.uniform(-1.0, 1.0, (m, 2))
    u, st, it = (a.cpu().numpy() for a in ctl.solve(t(X[:m]), t(up2), t(goal[:m]), t(ob2)))
    s2, i2 = compare(u, st, it, oracle_many(X[:m], up2, goal[:m], ob2), n_off=4)
    print(f"si ms kernel: bench draws optimal {np.mean(so == 0):.4f}, iterations mean {ito.mean():.1f} max {ito.max()}; crowded: optimal {np.mean(s2 == 0):.4f} "
          f"infeasible {np.mean(s2 == 1):.4f}, iterations max {i2.max()}")


def test_full_batch_f32_storage_and_the_condensed_kernel():
    X, up, goal, obs = W.mpc_family_batch("si", 4096, 8, seed=0)
    f = lambda a: t(a.astype(np.float32), torch.float32)                          # noqa: E731
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f32")
    u1, s1, i1 = ctl.solve(f(X), f(up), f(goal), f(obs))
    u2, s2, i2 = ctl.solve(f(X), f(up), f(goal), f(obs))
    assert torch.equal(u1, u2) and torch.equal(s1, s2) and torch.equal(i1, i2)
    assert (u1.abs() <= 1.0 + 1e-6).all()
    uc, sc, ic = sca.BatchedLinearMPCCBF(SPEC, io_dtype="f32").solve(f(X), f(up), f(goal), f(obs))[:3]
    both = (s1 == 0) & (sc == 0)
    same = ((u1 - uc).abs().amax(dim=1) <= 1e-4)[both].double().mean().item()
    assert both.double().mean().item() >= 0.97 and same >= 0.99, (both.double().mean().item(), same)
    print(f"si ms kernel, 4096: optimal {(s1 == 0).double().mean().item():.4f}, iterations mean {i1.double().mean().item():.1f} max {int(i1.max())}; same optimum as the condensed kernel on {same:.4f}")


def test_drop_in_class_on_request():
    from safe_control_amd.position_control.mpc_cbf import MPCCBF

    class Robot:
        dt, robot_radius = 0.05, 0.25
    assert MPCCBF(Robot(), {"model": "SingleIntegrator2D"}, num_obs=8)._ms is None
    ctl = MPCCBF(Robot(), {"model": "SingleIntegrator2D", "mpc_formulation": "multiple_shooting"}, num_obs=8)
    assert ctl._ms is not None
    X, up, goal, obs = (a[:2] for a in W.mpc_family_batch("si", 64, 8, seed=2))
    u = ctl.solve_control_problem(X[0].reshape(-1, 1), dict(goal=goal[0], state_machine="track", u_ref=np.zeros((2, 1))), obs[0])
    uo = MS.solve(MS.si_model(), X[0], np.zeros(2), goal[0], obs[0], opts=dict(MS.KERNEL_PROFILE))[0]
    assert np.abs(u.reshape(-1) - uo).max() <= 1e-8
