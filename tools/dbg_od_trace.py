import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch
import safe_control_amd as sca
from oracle import ms_ipopt as MS
import importlib.util
spec = importlib.util.spec_from_file_location("e", "/root/repo/tools/exp_ms_od_kernel.py"); 
sys.argv=['x','64','64']
e = importlib.util.module_from_spec(spec); spec.loader.exec_module(e)
i = int(os.environ.get("PROB", "2"))
X, up, goal, obs = e.X, e.up, e.goal, e.obs
t = lambda a: torch.tensor(np.ascontiguousarray(a[i:i+1]), dtype=torch.float64, device="cuda:0")
ctl = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False, max_iter=int(os.environ.get("MAXIT", "60")))
u, rho, st, it, trace = ctl.solve(t(X), t(up), t(goal), t(obs), want_trace=True)
tr=[]
MS.solve(e.mdl, X[i], up[i], goal[i], obs[i], opts=dict(e.PROFILE, max_iter=int(os.environ.get("MAXIT", "60"))), trace=tr)
T = np.array([[q["E0"], q["dinf"], q["pinf"], q["comp"], q["mu"], q["theta"], q["delta"], q["alpha"]] for q in tr])
K = trace[0].cpu().numpy()
m = min(len(T), len(K), int(os.environ.get("MAXIT", "60")))
rel = np.abs(K[:m] - T[:m]) / np.maximum(1e-9, np.abs(T[:m]))
w = np.argwhere(rel > 1e-6)
r0 = w[0][0] if len(w) else m - 3
print("first parting at iteration", r0)
for r in range(max(0, r0 - 2), min(m, r0 + 4)):
    print('k', np.array2string(K[r], precision=6, max_line_width=200)); print('o', np.array2string(T[r], precision=6, max_line_width=200))
