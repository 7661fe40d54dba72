"""GPU experiment: per-phase shader-clock profile of mpc_gn.hip (needs a -DSC_GN_PROF build linked as SC_EXP_LIB)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from safe_control_amd import _lib as _L
_L.LIB_PATH = os.environ["SC_EXP_LIB"]
import safe_control_amd as sca
import test_mpcgn_gpu as T
name = os.environ.get("SC_EXP_MODEL", "Quad2D"); N = 10; K = 8; B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(3)
if name.startswith("KinematicBicycle2D"):
    import test_mpcgn_kb_gpu as TK
    from oracle import mpc_gn as G
    mdl = G.kb_model()
    draw = lambda m, r, k: TK.draw(m, r, k, clear=1.5)
    T.u_start = lambda m: np.zeros(2)
else:
    mdl = T.MODELS[name]()
    draw = T.draw
X = np.zeros((B, mdl["nx"])); Gl = np.zeros((B, 2)); O = np.zeros((B, K, 7))
for i in range(B):
    X[i], Gl[i], O[i] = draw(mdl, rng, K)
ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
u, st, it, z = ctl.solve(T.t(X), T.t(np.tile(T.u_start(mdl), (B, 1))), T.t(Gl), T.t(O), want_z=True)
torch.cuda.synchronize()
ph = z.cpu().numpy()[:, :12]; itn = it.cpu().numpy()
names = ["eval+sens", "grad+jt", "backward", "resid/rhs jt", "Psi", "Hc", "T+HV+M", "chol", "pdz/ds", "linesearch", "-", "update"]
tot = ph.sum()
print(name, "iterations mean", itn.mean(), "cycles per iteration", tot / itn.sum())
for i, nm in enumerate(names):
    print(f"  {nm:12s} {ph[:, i].sum() / itn.sum():10.0f}  {100 * ph[:, i].sum() / tot:5.1f} %")
