"""GPU: kernel 12's iteration traces (f64 storage) of chosen NLPs from tools/data/vtol_lost_nlps.npz -> gpurun_out/lost_trace.npz; on the CPU
(`cmp`) the oracle's trace beside it: the first iteration at which the two part.
    python3 tools/dbg_vtol_lost_trace.py gpu 44:2 73:0 ..      |      python3 tools/dbg_vtol_lost_trace.py cmp"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vtol_lost_nlps.npz"))
if sys.argv[1] == "gpu":
    import torch
    import safe_control_amd as sca
    idx = [int(np.nonzero((d["aircraft"] == int(a.split(":")[0])) & (d["step"] == int(a.split(":")[1])))[0][0]) for a in sys.argv[2:]]
    ctl = sca.BatchedVtolMSMPCCBF({"model": "VTOL2D", "radius": 0.6, "v_max": 20.0}, io_dtype="f64", fallback=False)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    u, st, it, tr = ctl.solve(t(d["X"][idx]), t(d["up"][idx]), t(d["goal"][idx][:, :2]), t(d["ob"][idx]), want_trace=True)
    np.savez_compressed("gpurun_out/lost_trace.npz", idx=np.array(idx), u=u.cpu().numpy(), st=st.cpu().numpy(), it=it.cpu().numpy(), trace=tr.cpu().numpy()[:, :1000])
    print(st.tolist(), it.tolist())
else:
    from oracle import ms_ipopt as MS
    g = np.load("gpurun_out/lost_trace.npz")
    mdl = MS.vtol_model(dict(radius=0.6, v_max=20.0))
    for n, i in enumerate(g["idx"]):
        tr = []
        u, st, it, info = MS.solve(mdl, d["X"][i], d["up"][i], d["goal"][i], d["ob"][i], return_info=True, opts=dict(MS.KERNEL_PROFILE), trace=tr)
        T = np.array([[q["E0"], q["dinf"], q["pinf"], q["comp"], q["mu"], q["theta"], q["delta"], q["alpha"]] for q in tr])
        K = g["trace"][n][:int(g["it"][n]) + 1].copy()
        K[:, 7] = np.abs(K[:, 7])
        # align: the oracle adds a row when it enters / leaves its restoration phase, the kernel overwrites one -- walk both, let the oracle skip a row
        same = lambda a, b: bool((np.abs(a[:6] - b[:6]) <= 1e-3 * np.maximum(1e-9, np.abs(b[:6]))).all())
        i_o, j_k, skipped = 0, 0, []
        while i_o < len(T) and j_k < len(K):
            if same(K[j_k], T[i_o]):
                i_o += 1; j_k += 1
            elif i_o + 1 < len(T) and same(K[j_k], T[i_o + 1]):
                skipped.append(i_o); i_o += 1
            else:
                break
        print(f"aircraft {int(d['aircraft'][i])} step {int(d['step'][i])}: oracle {info['status']} it {it}; kernel status {int(g['st'][n])} it {int(g['it'][n])}; "
              f"rows in step until kernel row {j_k} / oracle row {i_o} (oracle rows skipped: {skipped})")
        for r in range(-3, 3):
            if 0 <= i_o + r < len(T): print("   oracle", i_o + r, " ".join(f"{v: .6e}" for v in T[i_o + r]), "resto" if tr[i_o + r].get("resto") else "")
            if 0 <= j_k + r < len(K): print("   kernel", j_k + r, " ".join(f"{v: .6e}" for v in K[j_k + r]))
        resto = [r for r in range(len(tr)) if tr[r].get("resto")]
        print("   oracle restoration iterations:", (resto[0], resto[-1]) if resto else None, "; kernel's last rows:")
        for r in range(max(0, len(K) - 2), len(K)):
            print("   it", r, "kernel", " ".join(f"{v: .4e}" for v in K[r]))
