import os, sys
sys.path.insert(0,'/root/repo'); sys.argv=['x','4','4096']
import numpy as np, torch
import importlib.util
spec = importlib.util.spec_from_file_location("e", "/root/repo/tools/exp_ms_od_kernel.py")
e = importlib.util.module_from_spec(spec); spec.loader.exec_module(e)
import safe_control_amd as sca
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
c = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False)
r = c.solve(t(e.X), t(e.up), t(e.goal), t(e.obs))
st, it = r[2].cpu().numpy(), r[3].cpu().numpy()
bad = np.flatnonzero(it >= 3000)
print("crawlers", bad, "status", st[bad])
i = int(bad[0])
c2 = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False, max_iter=400)
r = c2.solve(t(e.X[i:i+1]), t(e.up[i:i+1]), t(e.goal[i:i+1]), t(e.obs[i:i+1]), want_trace=True)
K = r[-1][0].cpu().numpy()
np.set_printoptions(linewidth=220, precision=4)
resto = K[:, 7] < 0
print("resto iterations:", int(resto.sum()), "first", np.flatnonzero(resto)[:5], "transitions", np.flatnonzero(np.diff(resto.astype(int)) != 0)[:40])
for j in list(range(0, 400, 16)): print(j, K[j])
