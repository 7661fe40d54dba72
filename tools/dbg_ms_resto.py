"""GPU + CPU: kernel 12 with its in-kernel restoration phase against oracle/ms_ipopt.py (kernel profile, resto_elastic = "ineq"), iteration traces
side by side.   MODE=od|plain|scene PROB=<index in the bench batch> python3 tools/dbg_ms_resto.py"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca
from safe_control_amd import workloads as W
from oracle import ms_ipopt as MS

mode, i = os.environ.get("MODE", "od"), int(os.environ.get("PROB", "6"))
maxit = int(os.environ.get("MAXIT", "600"))
Xn, up0, gn, on = W.mpc_family_batch("vtol", 4096, 8, seed=0)
if mode == "od":
    on = on.copy()
    rng = np.random.default_rng(100)
    r = rng.uniform(0.8, 1.6, 4096); d = 10.0 + 20.0 * rng.uniform(size=4096); off = rng.uniform(-1.0, 1.0, 4096)
    on[::2, 0, 0], on[::2, 0, 1], on[::2, 0, 2] = (Xn[:, 0] + d + r)[::2], (Xn[:, 1] + off)[::2], r[::2]
    mdl, ctl = MS.vtol_od_model(), sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False, max_iter=maxit)
    x0, up, g, ob = Xn[i], up0[i], gn[i], on[i]
elif mode == "scene":                                                       # first NLP of the reference's example scene: no feasible point
    ob = np.hstack([np.array([[67.0, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[73.0, float(z), 0.5] for z in range(1, 7)]), np.zeros((10, 4))])
    spec = dict(model="VTOL2D", radius=0.6, v_max=20.0)
    mdl, ctl = MS.vtol_model(dict(radius=0.6, v_max=20.0)), sca.BatchedVtolMSMPCCBF(spec, io_dtype="f64", fallback=False, max_iter=maxit)
    x0, up, g = np.array([2.0 + i, 10.0 - 0.05 * i, 0.0, 20.0, 0.0, 0.0]), np.zeros(4), np.array([70.0, 10.0])       # (PROB shifts the start)
elif mode == "file":                                                      # a solve saved by tools/exp_vtol_fleet.py
    d = np.load(os.environ.get("FILE", "tools/data/vtol_crawlers.npz"))
    spec = dict(model="VTOL2D", radius=0.6, v_max=20.0)
    mdl, ctl = MS.vtol_model(dict(radius=0.6, v_max=20.0)), sca.BatchedVtolMSMPCCBF(spec, io_dtype="f64", fallback=False, max_iter=maxit)
    x0, up, g, ob = d[f"X_{i}"], d[f"up_{i}"], d[f"g_{i}"][:2], d[f"ob_{i}"]
else:
    mdl, ctl = MS.vtol_model(), sca.BatchedVtolMSMPCCBF(io_dtype="f64", fallback=False, max_iter=maxit)
    x0, up, g, ob = Xn[i], up0[i], gn[i], on[i]
t = lambda a: torch.tensor(np.ascontiguousarray(a[None]), dtype=torch.float64, device="cuda:0")
r = ctl.solve(t(x0), t(up), t(g), t(ob), want_trace=True)
torch.cuda.synchronize()
u, st, it, K = r[0][0].cpu().numpy(), int(r[-3][0]), int(r[-2][0]), r[-1][0].cpu().numpy()
tr = []
uo, so, ito, info = MS.solve(mdl, x0, up, g, ob, return_info=True, trace=tr, opts=dict(MS.KERNEL_PROFILE, max_iter=maxit))
T = np.array([[q["E0"], q["dinf"], q["pinf"], q["comp"], q["mu"], q["theta"], q["delta"], -q["alpha"] if q["resto"] else q["alpha"]] for q in tr])
print(f"kernel: status {st} iterations {it} u0 {u}\noracle: status {so} ({info['status']}) iterations {ito} u0 {uo[:4]}; restoration iterates {sum(1 for q in tr if q['resto'])}")
m = min(len(T), it + 1)
rel = np.abs(K[:m] - T[:m]) / np.maximum(1e-9, np.abs(T[:m]))
w = np.argwhere(rel > 1e-5)
r0 = int(w[0][0]) if len(w) else m - 3
first_resto = next((j for j, q in enumerate(tr) if q["resto"]), None)
print("first parting at iteration", r0, "; oracle enters the restoration at", first_resto)
show = sorted(set(list(range(max(0, r0 - 2), min(m, r0 + 4))) + ([] if first_resto is None else list(range(max(0, first_resto - 1), min(m, first_resto + 4))))))
if os.environ.get("TAIL"):
    show = sorted(set(show + list(range(max(0, len(T) - int(os.environ["TAIL"])), len(T))) + list(range(max(0, m - 3), m))))
for j in show:
    if j >= len(K) or j >= len(T): continue
    print(j, 'k', np.array2string(K[j], precision=5, max_line_width=220)); print(j, 'o', np.array2string(T[j], precision=5, max_line_width=220))
