import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca
import importlib.util
spec = importlib.util.spec_from_file_location("e", os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp_ms_od_kernel.py"))
sys.argv = ['x', '4', '4096']
e = importlib.util.module_from_spec(spec); spec.loader.exec_module(e)
X, up, goal, obs = e.X, e.up, e.goal, e.obs
B = 4096
t = lambda a: torch.tensor(np.ascontiguousarray(a[:B]), dtype=torch.float64, device="cuda:0")
ctl = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False)
u, rho, st, it = ctl.solve(t(X), t(up), t(goal), t(obs))
it = it.cpu().numpy(); st = st.cpu().numpy()
print("iteration histogram:", np.histogram(it, bins=[0, 50, 100, 200, 400, 800, 1600, 3001])[0])
i = int(np.argsort(it)[-40])
print("problem", i, "iterations", it[i], "status", st[i])
t1 = lambda a: torch.tensor(np.ascontiguousarray(a[i:i + 1]), dtype=torch.float64, device="cuda:0")
r = ctl.solve(t1(X), t1(up), t1(goal), t1(obs), want_trace=True)
K = r[-1][0].cpu().numpy()
n = int(r[3][0])
for q in list(range(0, min(n, 40), 4)) + list(range(40, n, max(1, n // 40))):
    print(q, 'E0 %.2e dinf %.2e pinf %.2e comp %.2e mu %.1e th %.2e dw %.1e a %.2e' % tuple(K[q]))
