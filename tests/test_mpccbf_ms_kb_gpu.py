"""GPU: kernel 13 (csrc/mpc_du_ms.hip) instantiated for KinematicBicycle2D -- the reference's MPC-CBF NLP for that robot as do-mpc poses it
(position_control/mpc_cbf.py:31-33,64-66,135-141,202-208; robots/kinematic_bicycle2D.py:67-123,175-199: the DT barrier goes through robot.step,
which clips the speed to [v_min, v_max]) under IPOPT's algorithm -- against oracle/ms_ipopt.py with kb_model() in the kernel's profile.  The
inputs enter the positions directly (general stage layout: two columns of A and all of B per stage).  Where the plan slows down to v_min the
clip's kink sits on the solution and Newton's method cycles around it (IPOPT would, too: casadi differentiates fmin / fmax piecewise): those
solves run to the iteration limit on both sides, along paths that rounding separates after ~50 iterations -- they are compared by status only."""
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
SPEC = {"model": "KinematicBicycle2D"}
LIMIT = 150


def t(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def kb_oracle_model():
    sp = complete_robot_spec(dict(SPEC))
    return MS.kb_model({k: v for k, v in sp.items() if k in MS.kb_model()["spec"]})


def _one(args):
    x, up, g, ob = args
    os.environ["OMP_NUM_THREADS"] = "1"
    return MS.solve(kb_oracle_model(), x, up, g, ob, opts=dict(MS.KERNEL_PROFILE, max_iter=LIMIT))


def test_bench_draws_against_the_oracle():
    n = 320
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("kb", 4096, 8, seed=0))
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f64", max_iter=LIMIT)
    u, st, it = (a.cpu().numpy() for a in ctl.solve(t(X), t(up), t(goal), t(obs)))
    with Pool(min(32, os.cpu_count() or 4)) as p:
        res = p.map(_one, [(X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=2)
    uo, so, ito = np.array([r[0] for r in res]), np.array([r[1] for r in res]), np.array([r[2] for r in res])
    assert (st == so).mean() >= 0.99, np.flatnonzero(st != so)[:10]
    short = ito < 60                                                        # (the solves that end before rounding can separate two paths)
    assert short.mean() >= 0.9 and np.array_equal(st[short], so[short])
    off = short & (it != ito)
    assert off.sum() <= 4 and np.abs(it - ito)[short].max() <= 2, (int(off.sum()), int(np.abs(it - ito)[short].max()))
    du = np.abs(u - uo).max(axis=1)
    assert du[short & ~off].max() <= 1e-8, du[short & ~off].max()
    assert (so == 0).mean() >= 0.9 and (so == 1).sum() >= 3
    print(f"kb ms kernel: optimal {np.mean(so == 0):.4f}, infeasible {np.mean(so == 1):.4f}, at the limit of {LIMIT} {np.mean(ito >= LIMIT):.4f}; "
          f"iterations median {np.median(ito):.0f}; same status {np.mean(st == so):.4f}, same count {np.mean(it == ito):.4f}")


def test_full_batch_f32_storage_and_the_condensed_kernel_where_both_are_optimal():
    X, up, goal, obs = W.mpc_family_batch("kb", 4096, 8, seed=0)
    f = lambda a: t(a.astype(np.float32), torch.float32)                          # noqa: E731
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f32", max_iter=LIMIT)
    u1, s1, i1 = ctl.solve(f(X), f(up), f(goal), f(obs))
    u2, s2, i2 = ctl.solve(f(X), f(up), f(goal), f(obs))
    assert torch.equal(u1, u2) and torch.equal(s1, s2) and torch.equal(i1, i2)
    sp = complete_robot_spec(dict(SPEC))
    assert (u1[:, 0].abs() <= sp["a_max"] + 1e-5).all() and (u1[:, 1].abs() <= sp["beta_max"] + 1e-6).all()
    opt = (s1 == 0).double().mean().item()
    assert opt >= 0.9, opt
    uc, sc, ic = sca.BatchedGnMPCCBF(SPEC, io_dtype="f32").solve(f(X), f(up), f(goal), f(obs))[:3]
    both = (s1 == 0) & (sc == 0)
    same = ((u1 - uc).abs().amax(dim=1) <= 1e-3)[both].double().mean().item()
    assert both.double().mean().item() >= 0.85 and same >= 0.97, (both.double().mean().item(), same)
    print(f"kb ms kernel, 4096 at a limit of {LIMIT}: optimal {opt:.4f}, at the limit {(i1 >= LIMIT).double().mean().item():.4f}, iterations mean {i1.double().mean().item():.1f}; "
          f"same optimum as the condensed kernel on {same:.4f}")


def test_drop_in_class_serves_it_on_request_only():
    from safe_control_amd.position_control.mpc_cbf import MPCCBF

    class Robot:
        dt, robot_radius = 0.05, 0.3
    assert MPCCBF(Robot(), {"model": "KinematicBicycle2D"}, num_obs=8)._ms is None
    ctl = MPCCBF(Robot(), {"model": "KinematicBicycle2D", "mpc_formulation": "multiple_shooting"}, num_obs=8)
    assert ctl._ms is not None and ctl._ms.model == "KinematicBicycle2D"
    X, up, goal, obs = (a[:2] for a in W.mpc_family_batch("kb", 64, 8, seed=2))
    u = ctl.solve_control_problem(X[0].reshape(-1, 1), dict(goal=goal[0], state_machine="track", u_ref=np.zeros((2, 1))), obs[0])
    uo = MS.solve(kb_oracle_model(), X[0], np.zeros(2), goal[0], obs[0], opts=dict(MS.KERNEL_PROFILE))[0]
    assert np.abs(u.reshape(-1) - uo).max() <= 1e-8
