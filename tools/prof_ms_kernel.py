"""GPU: per-phase cycle counters of csrc/mpc_vtol_ms.hip (-DSC_MS_PROF build, SAFE_CONTROL_AMD_LIB=exp_libs/libsc_msprof.so).
    SAFE_CONTROL_AMD_LIB=exp_libs/libsc_msprof.so python3 tools/prof_ms_kernel.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
X, up, goal, obs = W.mpc_family_batch("vtol", max(B, 64), 8, seed=0)
if len(sys.argv) > 2: obs = obs[:, :int(sys.argv[2])]
t = lambda a: torch.tensor(np.ascontiguousarray(a[:B]), dtype=torch.float64, device="cuda:0")
ctl = sca.BatchedVtolMSMPCCBF(io_dtype="f64", fallback=False)
u, st, it, trace = ctl.solve(t(X), t(up), t(goal), t(obs), want_trace=True)
torch.cuda.synchronize()
t0 = time.time(); ctl.solve(t(X), t(up), t(goal), t(obs)); torch.cuda.synchronize(); dt = time.time() - t0
pr = trace[:, -1, :].cpu().numpy(); it = it.cpu().numpy()
names = ["eval2 (errors)", "errors + mu", "eval2 (build)", "riccati backward", "riccati forward", "finish_step", "line search", "update"]
tot = pr.sum(axis=1)
print(f"{B} problems, {dt * 1e3:.1f} ms, iterations mean {it.mean():.1f}; cycles per iteration {np.mean(tot / np.maximum(it, 1)):.0f}")
for i, n in enumerate(names):
    print(f"  {n:18s} {np.mean(pr[:, i] / np.maximum(it, 1)):9.0f} cycles / iteration  {100 * pr[:, i].sum() / tot.sum():5.1f} %")
