"""GPU experiment: per-phase shader-clock totals of the MPC-CBF kernel, plain against optimal decay, on the same 256 problems.
Needs a developer build:  make -C safe_control_amd/csrc EXTRA=-DSC_MPC_PROF   (then rebuild without it)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import safe_control_amd as sca
from safe_control_amd import workloads as W

NAMES = ["(eval_values tail)", "row pass", "stage pass", "col pass", "mu update", "T = Phi G", "rhs+condense MFMA",
         "cholesky", "chol_solve", "dp, dV", "step rows", "line search+update (excl. eval)",
         "eval(derivs): rollout", "eval(derivs): G", "eval(derivs): barrier", "eval(derivs): g, f",
         "eval(LS): rollout", "eval(LS): -", "eval(LS): barrier", "eval(LS): g, f"]
dev = torch.device("cuda:0")
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Xn, goal, un, on = W.du_cbfqp_batch(B, 8, seed=0)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
X, g, ob = t(Xn), t(goal), t(on)
up = torch.zeros((B, 2), dtype=torch.float32, device=dev)
res = {}
for name, ctl in (("plain", sca.BatchedMPCCBF(dict(spec), io_dtype="f32", horizon=10)),
                  ("optimal decay", sca.BatchedOptimalDecayMPCCBF(dict(spec), io_dtype="f32", horizon=10))):
    out = ctl.solve(X, up, g, ob, want_z=True)
    torch.cuda.synchronize()
    it, z = out[-2], out[-1]
    ph = z.cpu().numpy()[:, :20].astype(np.float64)
    itn = it.cpu().numpy().astype(np.float64)
    res[name] = (ph.sum(0) / itn.sum(), itn.mean())
print(f"B={B}; mean iterations plain {res['plain'][1]:.1f}, optimal decay {res['optimal decay'][1]:.1f}; cycles per iteration:")
for i, nm in enumerate(NAMES):
    a, b = res["plain"][0][i], res["optimal decay"][0][i]
    print(f"  {nm:34s} {a:10.0f} {b:10.0f}")
print(f"  {'total':34s} {res['plain'][0].sum():10.0f} {res['optimal decay'][0].sum():10.0f}")
