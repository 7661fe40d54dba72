"""GPU: the two ways the MPC kernels raise a superellipsoid coordinate to its exponent -- ipm::pow3's multiply chain for integer
exponents, pow() otherwise (csrc/mpc_ipm_common.hpp: CHAIN and its history) -- held together: the same scenes solved with exponents
4 / 6 / 10 (chain) and with exponents a relative 1e-12 off those integers (pow()) must give the same statuses, iteration counts
and first moves.  Covers the three kernel families that have a superellipsoid branch: csrc/mpc_cbf.hip (N = 10: pow() both ways,
N = 20 and run-time horizons: chain), csrc/mpc_lin.hip (SingleIntegrator2D) and csrc/mpc_gn.hip (DoubleIntegrator2D)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import safe_control_amd as sca  # noqa: E402
from safe_control_amd import workloads as W  # noqa: E402

DEV = "cuda:0"


def t(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)


CASES = [("du", "DynamicUnicycle2D", 10, 8), ("du", "DynamicUnicycle2D", 20, 8), ("du", "DynamicUnicycle2D", 7, 5),
         ("si", "SingleIntegrator2D", 10, 8), ("si", "SingleIntegrator2D", 10, 6), ("si", "SingleIntegrator2D", 14, 4),
         ("di", "DoubleIntegrator2D", 10, 8), ("di", "DoubleIntegrator2D", 8, 5)]


@pytest.mark.parametrize("fam,name,N,K", CASES)
def test_chain_and_pow_agree_on_superellipsoid_scenes(fam, name, N, K):
    B = 96
    X, up, goal, _ = W.mpc_family_batch(fam, B, K, seed=N + K)
    obs = W.superellipsoid_obstacles(X[:, :2], K, seed=7, exponents=(4.0, 6.0, 10.0))
    obs[:, K - 1] = [1000.0, 1000.0, 0, 0, 0, 0, 0]                       # one dummy row (a circle) per agent, as update_tvp pads
    off = obs.copy()
    off[:, :, 4] *= np.where(off[:, :, 6] > 0.5, 1.0 + 1e-12, 1.0)        # not an integer any more: the pow() branch
    if fam == "du":
        ctl = sca.BatchedMPCCBF({"model": name, "a_max": 1.0, "w_max": 0.5, "radius": 0.25}, io_dtype="f64", horizon=N)
    elif fam == "si":
        ctl = sca.BatchedLinearMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    else:
        ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    a = ctl.solve(t(X), t(up), t(goal), t(obs), want_z=True)
    b = ctl.solve(t(X), t(up), t(goal), t(off), want_z=True)
    torch.cuda.synchronize()
    ua, sta, ita, za = (v.cpu().numpy() for v in a)
    ub, stb, itb, zb = (v.cpu().numpy() for v in b)
    same = sta == stb
    assert same.mean() >= 0.97, np.flatnonzero(~same)                      # (a solve at a branch of the line search may part on 1e-12)
    ok = same & (sta == 0)
    assert ok.sum() >= B // 3
    assert np.abs(ua[ok] - ub[ok]).max() <= 1e-6 and np.abs(za[ok] - zb[ok]).max() <= 1e-5
    assert (np.abs(ita[ok] - itb[ok]) <= 1).mean() >= 0.97
