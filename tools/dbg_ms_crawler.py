"""CPU: replay the long closed-loop solves tools/exp_vtol_fleet.py saved (gpurun_out/vtol_crawlers.npz) with oracle/ms_ipopt.py.
   python3 tools/dbg_ms_crawler.py [index] [max_iter]"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ms_ipopt as MS
d = np.load("gpurun_out/vtol_crawlers.npz")
j = int(sys.argv[1]) if len(sys.argv) > 1 else 0
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 600
X, up, g, ob = d[f"X_{j}"], d[f"up_{j}"], d[f"g_{j}"], d[f"ob_{j}"]
print("state", X.round(3), "u_prev", up.round(3), "goal", g, "kernel iterations", int(d[f"it_{j}"]), "status", int(d[f"st_{j}"]), "u", d[f"u_{j}"].round(3))
print("obstacles", ob[:, :3].round(2).tolist())
mdl = MS.vtol_model(dict(radius=0.6, v_max=20.0))
for name, opts in (("kernel profile", dict(MS.KERNEL_PROFILE)), ("full algorithm", dict())):
    tr = []
    u, st, it, info = MS.solve(mdl, X, up, g[:2], ob, return_info=True, opts=dict(opts, max_iter=mx), trace=tr)
    nr = sum(1 for q in tr if q["resto"])
    print(f"{name}: {info['status']} after {it} iterations ({nr} in the restoration), u0 {u[:4].round(4)}")
    for q in tr[::max(1, len(tr) // 20)]:
        print("   ", q["it"], "R" if q["resto"] else " ", "E0 %.2e dinf %.2e pinf %.2e comp %.2e mu %.1e a %.1e dw %.1e" % (q["E0"], q["dinf"], q["pinf"], q["comp"], q["mu"], q["alpha"], q["delta"]))
