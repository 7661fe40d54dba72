"""GPU experiment: Gauss-Newton MPC-CBF kernel (mpc_gn.hip) vs oracle: iterate after `SC_EXP_IT` iterations, per-problem
status / iterations, batch timing."""
import sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import safe_control_amd as sca
from oracle import mpc_gn as G
import test_mpcgn_gpu as T

name = os.environ.get("SC_EXP_MODEL", "KinematicBicycle2D"); N = int(os.environ.get("SC_EXP_N", 10)); K = int(os.environ.get("SC_EXP_K", 8))
B = int(os.environ.get("SC_EXP_B", 8)); IT = int(os.environ.get("SC_EXP_IT", 100))
mdl = T.MODELS[name]()
rng = np.random.default_rng(N * 10 + K)
X = np.zeros((B, mdl["nx"])); Gl = np.zeros((B, 2)); O = np.zeros((B, K, 7))
for i in range(B):
    X[i], Gl[i], O[i] = T.draw(mdl, rng, K)
up = np.tile(T.u_start(mdl), (B, 1))
ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N, max_iter=IT)
u, st, it, z = ctl.solve(T.t(X), T.t(up), T.t(Gl), T.t(O), want_z=True)
torch.cuda.synchronize()
u, st, it, z = u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.cpu().numpy()
np.set_printoptions(precision=5, linewidth=200)
for i in range(min(B, 64)):
    uo, so, ito, info = G.solve(mdl, X[i], up[i], Gl[i], O[i], N=N, params_over={"max_iter": IT}, return_info=True)
    print(i, "st", st[i], so, "it", it[i], ito, "dz %.2e" % np.abs(z[i] - info["z"]).max(), "err %.2e" % info["err"])
    if IT <= 2 and i < 2:
        print("  gpu", z[i][:8]); print("  ora", info["z"][:8])
for Bt in [int(a) for a in sys.argv[1:]]:
    Xb = np.zeros((Bt, mdl["nx"])); Gb = np.zeros((Bt, 2)); Ob = np.zeros((Bt, K, 7))
    r2 = np.random.default_rng(3)
    for i in range(Bt):
        Xb[i], Gb[i], Ob[i] = T.draw(mdl, r2, K)
    ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
    a = (T.t(Xb), T.t(np.tile(T.u_start(mdl), (Bt, 1))), T.t(Gb), T.t(Ob))
    ctl.solve(*a); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        r = ctl.solve(*a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    itn = r[2].cpu().numpy()
    print(f"B={Bt} ms={ms:.3f} solves/s={Bt/ms*1e3:.0f} iters mean={itn.mean():.2f} max={itn.max()} status={np.bincount(r[1].cpu().numpy(), minlength=4)}")
