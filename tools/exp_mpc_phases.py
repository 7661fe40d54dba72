"""GPU experiment: per-phase shader-clock totals of the MPC-CBF kernel.
Needs a developer build:  make -C safe_control_amd/csrc EXTRA=-DSC_MPC_PROF   (then rebuild without it)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import os
from safe_control_amd import _lib as _L
if os.environ.get("SC_EXP_LIB"): _L.LIB_PATH = os.path.abspath(os.environ["SC_EXP_LIB"])
import safe_control_amd as sca
from safe_control_amd import workloads as W

NAMES = ["(eval_values tail)", "row pass", "stage pass", "col pass", "mu update", "T = Phi G", "rhs+condense MFMA",
         "cholesky", "chol_solve", "dp, dV", "step rows", "line search+update (excl. eval)",
         "eval(derivs): rollout", "eval(derivs): G", "eval(derivs): barrier", "eval(derivs): g, f",
         "eval(LS): rollout", "eval(LS): -", "eval(LS): barrier", "eval(LS): g, f"]
dev = torch.device("cuda:0")
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
ctl = sca.BatchedMPCCBF(dict(spec), io_dtype="f32", horizon=10)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256          # <= 1 wave per SIMD on the whole chip: pure latency
Xn, goal, un, on = W.du_cbfqp_batch(B, 8, seed=0)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
X, g, ob = t(Xn), t(goal), t(on)
up = torch.zeros((B, 2), dtype=torch.float32, device=dev)
u, st, it, z = ctl.solve(X, up, g, ob, want_z=True)
torch.cuda.synchronize()
ph = z.cpu().numpy()[:, :20].astype(np.float64)
itn = it.cpu().numpy().astype(np.float64)
tot = ph.sum(1)
print(f"B={B} mean iters {itn.mean():.2f}; cycles/iter {tot.sum()/itn.sum():.0f}")
for i, nm in enumerate(NAMES):
    print(f"  {nm:24s} {ph[:, i].sum()/itn.sum():10.0f} cyc/iter  {100*ph[:, i].sum()/tot.sum():5.1f}%")
