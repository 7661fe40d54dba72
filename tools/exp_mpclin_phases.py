"""GPU experiment: per-phase shader-clock profile of the linear-model MPC kernel (needs a -DSC_LIN_PROF build of
mpc_lin.hip linked as SC_EXP_LIB; the phase counters come back through z_out)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from safe_control_amd import _lib as _L
_L.LIB_PATH = os.environ["SC_EXP_LIB"]
import safe_control_amd as sca
import test_mpclin_gpu as T
name = os.environ.get("SC_EXP_MODEL", "Quad3D"); N = int(os.environ.get("SC_EXP_N", "10")); K = 8; B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mdl, X, G, O = T.batch(name, B, K, seed=3)
ctl = sca.BatchedLinearMPCCBF({"model": name}, io_dtype="f64", horizon=N)
u, st, it, z = ctl.solve(T.t(X), T.t(np.zeros((B, mdl["nu"]))), T.t(G), T.t(O), want_z=True)
torch.cuda.synchronize()
ph = z.cpu().numpy()[:, :12]; itn = it.cpu().numpy()
names = ["eval", "grad", "jt(lam)+", "resid/mu", "rhs jt", "Phi", "T+M", "chol", "pdz/ds", "linesearch", "-", "update"]
tot = ph.sum()
print("iterations mean", itn.mean(), "cycles per iteration", tot / itn.sum())
for i, nm in enumerate(names):
    print(f"  {nm:12s} {ph[:, i].sum() / itn.sum():10.0f}  {100 * ph[:, i].sum() / tot:5.1f} %")
