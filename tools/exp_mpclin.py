"""GPU experiment: linear-model MPC-CBF kernel vs oracle (per-problem status / iterations / differences) and batch timing."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import safe_control_amd as sca
from oracle import mpc_lin as L
import test_mpclin_gpu as T

name = os.environ.get("SC_EXP_MODEL", "Quad3D"); N = int(os.environ.get("SC_EXP_N", 10)); K = int(os.environ.get("SC_EXP_K", 8))
B = int(os.environ.get("SC_EXP_B", 24))
mdl, X, G, O = T.batch(name, B, K, seed=N * 10 + K)
up = np.random.default_rng(1).uniform(-0.2, 0.2, (B, mdl["nu"]))
ctl = sca.BatchedLinearMPCCBF({"model": name}, io_dtype="f64", horizon=N)
u, st, it, z = ctl.solve(T.t(X), T.t(up), T.t(G), T.t(O), want_z=True)
torch.cuda.synchronize()
u, st, it, z = u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.cpu().numpy()
if B <= 64:
    for i in range(B):
        uo, so, ito, info = L.solve(mdl, X[i], up[i], G[i], O[i], N=N, return_info=True)
        print(i, "st", st[i], so, "it", it[i], ito, "du %.2e dz %.2e" % (np.abs(u[i] - uo).max(), np.abs(z[i] - info["z"]).max()),
              "err %.2e mu %.1e gmin %.2e" % (info["err"], info["mu"], info["g"][: N * K].min()))
for Bt in [int(a) for a in sys.argv[1:]]:
    mdl, X, G, O = T.batch(name, Bt, K, seed=3)
    tX, tu, tg, to = T.t(X), T.t(np.zeros((Bt, mdl["nu"]))), T.t(G), T.t(O)
    ctl.solve(tX, tu, tg, to); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        r = ctl.solve(tX, tu, tg, to)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    itn = r[2].cpu().numpy()
    print(f"B={Bt} ms={ms:.3f} solves/s={Bt/ms*1e3:.0f} iters mean={itn.mean():.2f} max={itn.max()} status={np.bincount(r[1].cpu().numpy(), minlength=4)}")
