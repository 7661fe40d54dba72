"""GPU parity of the optimal-decay MPC-CBF kernels for the relative-degree-1 models -- BASELINE config 5's EXTENSION
(Unicycle2D + Quad3D, superellipsoid obstacles, N = 20) -- against oracle/od_mpc_rd1.py.  No reference counterpart
(parity unpinned: see the oracle header).  Tolerances: same status, |u0 - u0_oracle| <= 1e-6, |z - z_oracle| <= 2e-5,
|rho - rho_oracle| <= 1e-4, iterations within 2, every row of a reported optimum >= -1e-6."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
from _oracle_pool import od_rd1_solve_many                # noqa: E402
from oracle import od_mpc_rd1 as O                        # noqa: E402
from safe_control_amd import workloads as W               # noqa: E402

DEV = "cuda:0"
UNI_SPEC = {"model": "Unicycle2D", "v_max": 1.0, "w_max": 0.5, "radius": 0.25}


def uni_scene(B, K, seed, superell=True):
    X, goal, ur, obs = W.du_cbfqp_batch(B, K, seed=seed)
    X[:, 3] = 0.0
    if superell:
        obs = W.superellipsoid_obstacles(X[:, :2], K, seed=seed + 1000)
        obs[::5, 0] = W.du_cbfqp_batch(B, K, seed=seed)[3][::5, 0]         # every fifth agent keeps one circle
    return X, goal, obs


def run_uni(X, goal, obs, N, io="f64"):
    import safe_control_amd as sca
    td = torch.float64 if io == "f64" else torch.float32
    ctl = sca.BatchedOptimalDecayMPCCBF(dict(UNI_SPEC), io_dtype=io, horizon=N, extension=True)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=td, device=DEV)
    B = X.shape[0]
    tX, tg, to = t(X), t(goal), t(obs)
    u, rho, st, it, z = ctl.solve(tX, torch.zeros((B, 2), dtype=td, device=DEV), tg, to, want_z=True)
    torch.cuda.synchronize()
    seen = tuple(a.double().cpu().numpy() for a in (tX, tg, to))
    return u.double().cpu().numpy(), rho.double().cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.double().cpu().numpy(), seen


@pytest.mark.parametrize("N,K,superell", [(20, 8, True), (10, 8, True), (10, 5, False), (7, 3, True)])
def test_unicycle2d_against_oracle(N, K, superell):
    B = 64
    X, goal, obs = uni_scene(B, K, seed=N * 10 + K, superell=superell)
    u, rho, st, it, z, (Xs, gs, os_) = run_uni(X, goal, obs, N)
    r = od_rd1_solve_many("od_uni", Xs[:, :3], np.zeros((B, 2)), gs, os_, params={"N": N})
    assert np.array_equal(st, r["st"]), np.flatnonzero(st != r["st"])
    ok = r["st"] == O.STATUS_OPTIMAL
    assert ok.sum() >= 0.7 * B
    assert np.abs(u[ok] - r["u"][ok]).max() <= 1e-6
    assert np.abs(z[ok] - r["z"][ok]).max() <= 2e-5
    assert np.abs(rho[ok][:, 0::2] - r["rho"][ok]).max() <= 1e-4            # omega1_k; omega2_k is inert
    assert np.all(rho[:, 1::2] == 1.0)
    assert np.abs(it[ok] - r["it"][ok]).max() <= 2
    P = O.uni_params(N=N)
    for i in np.flatnonzero(ok)[:16]:
        g = O.evaluate(Xs[i, :3], np.concatenate([z[i], rho[i, 0::2]]), gs[i], os_[i], P, level=0)["g"]
        assert g.min() >= -1e-6
    assert np.abs(rho[ok][:, 0::2] - 1.0).max() > 1e-3                       # the decay rates do move


def test_unicycle2d_f32_storage_and_batch_position_independence():
    X, goal, obs = uni_scene(512, 8, seed=5)
    u, rho, st, it, z, _ = run_uni(X, goal, obs, 20, io="f32")
    assert (st == 0).mean() > 0.7
    sel = np.arange(100, 164)
    u2, rho2, st2, it2, z2, _ = run_uni(X[sel], goal[sel], obs[sel], 20, io="f32")
    assert np.array_equal(u[sel], u2) and np.array_equal(st[sel], st2) and np.array_equal(rho[sel], rho2)
