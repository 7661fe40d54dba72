"""GPU: kernel 13's instantiations away from the bench shape: horizons 5 / 20 / 40 (four, two and one lane per stage: the stage sums on
quad_perm DPP, a pair, or not at all) with 1 / 3 / 16 obstacle slots, f64 storage, against oracle/ms_ipopt.py in the kernel's profile -- same
status, same iteration count and u0 to 1e-8 on the solves that end within 60 iterations (the bicycle's cycling solves are compared by status)."""
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
DU = {"a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25}
MK = {"du": MS.du_model, "di": MS.di_model, "kb": MS.kb_model, "uni": MS.uni_model, "si": MS.si_model}


def _one(a):
    os.environ["OMP_NUM_THREADS"] = "1"
    fam, N, x, up, g, ob = a
    sp = complete_robot_spec(dict({"model": W.MPC_FAMILIES[fam]}, **(DU if fam == "du" else {})))
    return MS.solve(MK[fam]({k: v for k, v in sp.items() if k in MK[fam]()["spec"]}), x, up, g, ob, N=N, opts=dict(MS.KERNEL_PROFILE, max_iter=150))


@pytest.mark.parametrize("N,K", [(5, 1), (20, 3), (40, 16)])
def test_horizons_and_obstacle_counts(N, K):
    n = 16
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)     # noqa: E731
    for fam in ("du", "di", "uni", "si", "kb"):
        X, up, goal, obs = (a[:n].copy() for a in W.mpc_family_batch(fam, 64, K, seed=N))
        ctl = sca.BatchedMSMPCCBF(dict({"model": W.MPC_FAMILIES[fam]}, **(DU if fam == "du" else {})), io_dtype="f64", horizon=N, max_iter=150)
        u, st, it = (a.cpu().numpy() for a in ctl.solve(t(X), t(up), t(goal), t(obs)))
        with Pool(min(32, os.cpu_count() or 4)) as p:
            res = p.map(_one, [(fam, N, X[i], up[i], goal[i], obs[i]) for i in range(n)], chunksize=2)
        so = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res]); uo = np.array([r[0] for r in res])
        assert np.array_equal(st, so), (fam, N, K)
        short = ito < 60
        assert (it != ito)[short].sum() <= 1 and np.abs(u - uo)[short & (it == ito)].max() <= 1e-8, (fam, N, K)
    # one obstacle table shared by the batch (obs_shared) gives what per-agent copies of it give
    X, up, goal, obs = (a[:n].copy() for a in W.mpc_family_batch("uni", 64, K, seed=N))
    ctl = sca.BatchedMSMPCCBF({"model": "Unicycle2D"}, io_dtype="f64", horizon=N, max_iter=150)
    a = ctl.solve(t(X), t(up), t(goal), t(obs[0]))
    b = ctl.solve(t(X), t(up), t(goal), t(np.repeat(obs[:1], n, 0)))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
