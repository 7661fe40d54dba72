"""GPU + host cores: tests/test_od_vtol_gpu.py::test_batch_against_oracle on a larger batch.   python3 tools/exp_od_vtol_wide.py [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import safe_control_amd as sca
from _oracle_pool import od_vtol_solve_many
import test_od_vtol_gpu as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
X, up, goal, obs = T.hard_batch(n)
ctl = sca.BatchedOptimalDecayVtolMPCCBF(io_dtype="f64", max_iter=600)
u, rho, st, it, z = (a.cpu().numpy() for a in ctl.solve(T.t(X), T.t(up), T.t(goal), T.t(obs), want_z=True))
o = od_vtol_solve_many(X, up, goal, obs, params={"max_iter": 600}, timeout=6000)
same = st == o["st"]
ok = same & (o["st"] == 0)
du = np.abs(u - o["u"]).max(axis=1); dz = np.abs(z - o["z"]).max(axis=1); dr = np.abs(rho - o["rho"]).max(axis=1)
moved = np.abs(o["rho"] - 1.0).max(axis=1) > 1e-3
print(f"od vtol, {n} problems: optimal {np.mean(o['st'] == 0):.4f} infeasible {np.mean(o['st'] == 1):.4f} inaccurate {np.mean(o['st'] == 2):.4f}; "
      f"status differs on {int((~same).sum())} {np.flatnonzero(~same)[:10]}; iterations equal on {np.mean(it == o['it']):.4f}, mean {o['it'].mean():.1f} max {o['it'].max()}; "
      f"on the optimal ones max du {du[ok].max():.2e} dz {dz[ok].max():.2e} drho {dr[ok].max():.2e}; beyond 1e-6 in u: {int((du[ok] > 1e-6).sum())}; decay moved on {int((moved & ok).sum())}")
