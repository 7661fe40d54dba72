import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_control_amd as sca
from oracle import od_mpc_gn as OG
from safe_control_amd import workloads as W
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
for fam, name, mk, N, K in (("kb", "KinematicBicycle2D", OG.kb_model, 10, 8), ("quad2d", "Quad2D", OG.quad2d_model, 10, 8)):
    B = 32
    X, up, goal, obs = W.mpc_family_batch(fam, B, K, seed=N + K)
    ctl = sca.BatchedOptimalDecayGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    u, rho, st, it, z = (a.cpu().numpy() for a in ctl.solve(t(X), t(up), t(goal), t(obs), want_z=True))
    mdl = mk()
    for i in range(B):
        uo, ro, so, ito, info = OG.solve(mdl, X[i], up[i], goal[i], obs[i], N=N, return_info=True)
        flag = "" if (st[i] == so and abs(it[i] - ito) <= 2) else "   <<<<"
        print(fam, i, "gpu", st[i], it[i], np.round(u[i], 6), "oracle", so, ito, np.round(uo, 6), f"err {info['err']:.2e} gmin {info['g'].min():.2e}", flag)
