"""GPU: every MPC-CBF kernel against the numpy oracle on the WHOLE 4096-problem batch bench.py times for its model family
(workloads.mpc_family_batch, seed 0) -- the bar of tests/test_mpccbf_gpu.py::test_config3_full_batch_against_oracle for all of
them: the oracle runs on the host cores in child processes (tests/_oracle_pool.py); SAME STATUS on every problem, the
restoration phase included, and |u0 - u0_oracle| <= 1e-6, |z - z_oracle| <= 2e-5 on every problem both call optimal.

What is allowed to differ, and counted: two solvers that follow each other to rounding can part at a kink of the problem
functions (the speed clip / rescaling inside step(), the sqrt(max(., 0)) of the collision cone) or where a line search decides
on a difference of 1e-13 |phi|; such a problem ends with different statuses or iterates.  The test bounds their number by
`max_part` per family (measured: see the table in the test) instead of excusing them one by one."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402

sys.path.insert(0, os.path.dirname(__file__))
from _oracle_pool import family_solve_many  # noqa: E402

DEV = "cuda:0"


def t(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)


def gpu_solve(family, X, up, goal, obs, N=10):
    name = W.MPC_FAMILIES[family]
    if family == "du":
        ctl = sca.BatchedMPCCBF({"model": name, "a_max": 1.0, "w_max": 0.5, "radius": 0.25}, io_dtype="f64", horizon=N)   # oracle/mpc_cbf.py: DEFAULTS
    elif family in ("si", "quad3d"):
        ctl = sca.BatchedLinearMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    else:
        ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    u, st, it, z = ctl.solve(t(X), t(up), t(goal), t(obs), want_z=True)
    torch.cuda.synchronize()
    return u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.cpu().numpy()


# family: (problems, at most this many may part ways, at least this fraction optimal)
# measured (MI355X, round 3, bicycles with the slack reset): parted kb 0, c3bf 1, dpcbf 1, di 0, quad2d 1, si 0, quad3d 0 of 4096 each
CASES = {"kb": (4096, 4, 0.95), "c3bf": (4096, 8, 0.51), "dpcbf": (4096, 8, 0.69), "di": (4096, 2, 0.90), "quad2d": (4096, 2, 0.95),
         "si": (4096, 0, 0.99), "quad3d": (4096, 2, 0.88)}


@pytest.mark.parametrize("family", list(CASES))
def test_full_bench_batch_against_oracle(family):
    B, max_part, min_opt = CASES[family]
    X, up, goal, obs = W.mpc_family_batch(family, B, 8, seed=0)
    u, st, it, z = gpu_solve(family, X, up, goal, obs)
    o = family_solve_many(family, X, up, goal, obs)
    same = st == o["st"]
    ok = same & (o["st"] == 0)
    du = np.abs(u - o["u"]).max(axis=1); dz = np.abs(z - o["z"]).max(axis=1)
    parted = ~same | (ok & ((du > 1e-6) | (dz > 2e-5)))
    print(f"{family}: optimal {np.mean(o['st'] == 0):.4f} infeasible {np.mean(o['st'] == 1):.4f} inaccurate {np.mean(o['st'] == 2):.4f}; "
          f"parted {int(parted.sum())} (status {int((~same).sum())}); restoration entered on {np.mean(o['n_resto'] > 0):.4f}; "
          f"iterations equal on {np.mean(it == o['it']):.4f}")
    assert parted.sum() <= max_part, np.flatnonzero(parted)[:20]
    assert (o["st"] == 0).mean() >= min_opt
    # a certified infeasible problem keeps a violation; an optimal one has none
    assert np.all(o["theta"][o["st"] == 1] > 1e-6) and np.all(o["theta"][o["st"] == 0] <= 1e-6)
    # the restoration's minimiser is a defined point: where both certify infeasibility the returned inputs agree too
    inf = same & (o["st"] == 1)
    if inf.any():
        assert np.median(du[inf]) <= 1e-6


@pytest.mark.parametrize("family", ["c3bf", "dpcbf", "kb", "du"])
def test_no_feasible_plan_for_what_the_kernel_labels_infeasible(family):
    """SC_STATUS_INFEASIBLE as the KERNEL reports it on the first 256 problems of the family's bench batch, against an independent
    phase-1 (scipy L-BFGS-B on sum min(g, 0)^2 over the input box from the kernel's plan, the initial guess and random starts, with
    nothing but the oracle's problem functions; tests/test_oracle_mpc_resto.py).
    * STALL certificates (round 4, sc_resto_params.stall_iter; which ones: the oracle's `stalled` flag, statuses being equal): the
      search finds no feasible plan for ANY of them.
    * Certificates of converged restorations are LOCAL (a stationary point of the violation): the search may find a feasible plan
      from another start for a few of them -- measured: C3BF draws 62 (a plan ON the boundary, min g = 1e-14) and 204 (theta = 6.7e-4,
      a plan with min g = 1e-6) of 256, none in the other families; bounded here, listed in the output.
    * `optimal_inaccurate` is rare."""
    from _oracle_pool import phase_one_many
    n = 256
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch(family, 4096, 8, seed=0))
    u, st, it, z = gpu_solve(family, X, up, goal, obs)
    o = family_solve_many(family, X, up, goal, obs)
    assert np.mean(st == o["st"]) >= 0.99
    inf = np.flatnonzero(st == 1)
    best = phase_one_many(family, X[inf], up[inf], goal[inf], obs[inf], z[inf], starts=6)
    found = best >= -1e-7
    stall = (o["stalled"][inf] == 1) & (o["st"][inf] == 1)
    print(f"{family}: {len(inf)} labelled infeasible of {n} ({int(stall.sum())} by the stall certificate); a feasible plan found for draws "
          f"{inf[found]} (min g {best[found]}); inaccurate {np.mean(st == 2):.4f}")
    assert not (found & stall).any(), f"{family}: stall-certified draws {inf[found & stall]} have feasible plans"
    assert found.sum() <= 2
    assert np.mean(st == 2) <= 0.03


@pytest.mark.parametrize("family", ["du", "quad3d", "kb"])
def test_stall_window_of_two_iterations_follows_the_oracle(family):
    """The stall rule of the restoration (sc_resto_params.stall_iter / stall_theta) with a window of two iterations and no violation
    threshold, so that it fires in families whose restorations never stall by themselves (DynamicUnicycle2D: `mpc_cbf.hip`, Quad3D:
    `mpc_lin.hip`): the solves it stops end at the oracle's iteration with the oracle's status and input -- the counter and the
    reference violation are followed step for step."""
    from safe_control_amd import _lib
    n = 256
    X, up, goal, obs = (a[:n] for a in W.mpc_family_batch(family, 4096, 8, seed=0))
    name = W.MPC_FAMILIES[family]
    if family == "du":
        ctl = sca.BatchedMPCCBF({"model": name, "a_max": 1.0, "w_max": 0.5, "radius": 0.25}, io_dtype="f64", horizon=10)
    elif family == "quad3d":
        ctl = sca.BatchedLinearMPCCBF({"model": name}, io_dtype="f64", horizon=10)
    else:
        ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=10)
    ctl.resto = _lib.default_resto(stall_iter=2, stall_theta=1e-9)
    u, st, it, z = ctl.solve(t(X), t(up), t(goal), t(obs), want_z=True)
    torch.cuda.synchronize()
    u, st, it = u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    o = family_solve_many(family, X, up, goal, obs, params={"resto_stall_iter": 2, "resto_stall_theta": 1e-9})
    stalled = o["stalled"] == 1
    assert stalled.sum() >= 4, "the window must fire"
    assert np.mean(st == o["st"]) >= 0.99 and np.array_equal(st[stalled], o["st"][stalled])
    assert np.all(np.abs(it[stalled] - o["it"][stalled]) <= 1)
    same_it = stalled & (it == o["it"])                            # (a solve stopped one iteration apart returns another unfinished iterate)
    du = np.abs(u[same_it] - o["u"][same_it]).max(axis=1)
    # (unfinished iterates of an ill-conditioned restoration: measured 11 of 12 bicycles to 1e-9, one to 7e-5; the other families to 1e-9)
    assert same_it.sum() >= 0.8 * stalled.sum() and (du <= 1e-5).sum() >= len(du) - 1 and du.max() <= 1e-3
    base = family_solve_many(family, X, up, goal, obs)
    assert (base["it"][stalled] > o["it"][stalled]).all()          # without the window the same solves run on
