"""GPU: the NLPs of the f64 oracle's flights from the starts of the fleet's lost aircraft (recorded by tools/exp_ms_vtol_flight.py with REC=..,
packed into tools/data/vtol_lost_nlps.npz) solved by kernel 12 -- f64 and f32 storage -- and compared solve by solve: is an aircraft lost
because the kernel returns something else than the oracle on the same NLP, or because the closed loop amplifies rounding?
    python3 tools/exp_vtol_lost_replay.py [nlps.npz]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca

f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vtol_lost_nlps.npz")
d = np.load(f, allow_pickle=False)
spec = {"model": "VTOL2D", "radius": 0.6, "v_max": 20.0}
names = {0: "optimal", 1: "infeasible", 2: "inaccurate", 3: "max_iter", 4: "needs_resto", 5: "error"}
for io in ("f64", "f32"):
    ctl = sca.BatchedVtolMSMPCCBF(dict(spec), io_dtype=io, fallback=False)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=ctl.torch_dtype, device="cuda:0")
    u, st, it = ctl.solve(t(d["X"]), t(d["up"]), t(d["goal"][:, :2]), t(d["ob"]))
    u, st, it = u.double().cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    du = np.abs(u - d["u"]).max(axis=1)
    print(f"== storage {io}: {len(du)} NLPs; same status {(st == d['st']).sum()}, same iteration count {(it == d['it']).sum()}, |u - u_oracle| <= 1e-6: {(du <= 1e-6).sum()}, <= 1e-3: {(du <= 1e-3).sum()}, > 0.1: {(du > 0.1).sum()}")
    for i in range(len(du)):
        flag = "" if du[i] <= 1e-6 else ("  <-- differs" if du[i] > 1e-3 else "  (small)")
        print(f"  aircraft {int(d['aircraft'][i]):3d} step {int(d['step'][i]):2d}  oracle {str(d['status'][i]):20s} it {int(d['it'][i]):4d} u {np.round(d['u'][i], 3)}   kernel {names.get(int(st[i]), st[i]):10s} it {int(it[i]):4d} u {np.round(u[i], 3)}  |du| {du[i]:.1e}{flag}")
