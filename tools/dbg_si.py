import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_control_amd as sca
from oracle import mpc_cbf as M, mpc_lin as L
g = np.load("tests/golden/closed_loop_integrators.npz")
obs = g["si/obs"]
mdl = L.si_model({"v_max": 1.0, "radius": 0.25})
X = np.array([[2.0, 2.0], [6.0, 1.0], [1.0, 6.0], [3.0, 4.0]]); goal = np.array([[2.0, 12.0], [1.0, 4.0], [1.0, 12.0], [9.0, 9.0]])
for K in (6, 8):
    d2 = ((obs[None, :, :2] - X[:, None, :]) ** 2).sum(-1)
    O = np.stack([M.pad_obstacles(list(obs[np.argsort(d2[i])[:K]]), K) for i in range(len(X))])
    up = np.zeros((len(X), 2))
    ctl = sca.BatchedLinearMPCCBF({"model": "SingleIntegrator2D", "v_max": 1.0, "radius": 0.25}, io_dtype="f64", horizon=10)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda")
    u, st, it, z = ctl.solve(t(X), t(up), t(goal), t(O), want_z=True)
    u, st, it = u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    for i in range(len(X)):
        uo, so, ito, info = L.solve(mdl, X[i], up[i], goal[i], O[i], N=10, return_info=True)
        print("K", K, "prob", i, "gpu", u[i], st[i], it[i], " oracle", uo, so, ito, "n_resto", info["n_resto"])
