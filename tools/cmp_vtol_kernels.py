"""The two VTOL2D MPC-CBF kernels against each other on the vtol workload batch: one NLP per wavefront (kernel = 2) and one NLP per
lane (kernel = 1) run the same interior point; statuses, iteration counts and plans should agree to rounding.  MI355X only.
  python tools/cmp_vtol_kernels.py [B]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_control_amd as sca
from safe_control_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"
X, up, goal, obs = W.mpc_family_batch("vtol", B, 8, seed=0)
tt = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
X, up, goal, obs = tt(X), tt(up), tt(goal), tt(obs)
res = {}
for kern in (2, 1):
    ctl = sca.BatchedVtolMPCCBF(io_dtype="f64"); ctl.kernel = kern
    u, st, it, z = ctl.solve(X, up, goal, obs, want_z=True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); u, st, it, z = ctl.solve(X, up, goal, obs, want_z=True); e1.record(); torch.cuda.synchronize()
    res[kern] = (u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), z.cpu().numpy())
    print(f"kernel {kern}: {e0.elapsed_time(e1):9.2f} ms per {B} problems | optimal {np.mean(res[kern][1] == 0):.4f} | iterations mean {res[kern][2].mean():.1f} "
          f"max {res[kern][2].max()}", flush=True)
(u2, s2, i2, z2), (u1, s1, i1, z1) = res[2], res[1]
same = s1 == s2
both = same & (s1 == 0)
print(f"status equal {same.mean():.4f}, iterations equal {np.mean(i1 == i2):.4f}, max |du0| on common optima {np.abs(u1 - u2)[both].max():.2e}, "
      f"max |dz| {np.abs(z1 - z2)[both].max():.2e}; first differing problems {np.flatnonzero(~same | (i1 != i2))[:10]}")
