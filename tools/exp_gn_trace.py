"""GPU: per-iteration scalars of the mpc_gn interior point from -DSC_GN_TRACE builds (tools/build_variants.sh mpc_gn T3="-DSC_GN_TRACE" ...):
prints, for problem PROB of the family's bench batch, where two builds part.
    python3 tools/exp_gn_trace.py FAMILY PROB exp_libs/lib_A.so exp_libs/lib_B.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
NAMES = ["it", "e_d", "e_p", "e_c0", "lmx", "mu", "theta", "f", "delta", "chol_ok", "rs_min", "rl_min", "sum_ds_s", "sum_rp", "sum_log", "gdz",
         "nu_m", "phi0", "dphi", "alpha", "phit", "accepted", "sum_dz", "sum_rhs", "tr_M", "resto", "ft", "slog", "srp"]
if sys.argv[1] == "--child":
    fam, B, lib, out = sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    from safe_control_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(lib)
    import ctypes as C
    import torch
    from safe_control_amd.position_control import mpc_cbf_gn as G
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_tail as E
    E.B = B
    ctl, arrs = E.make(fam, 100)
    X, up, g, ob = [torch.tensor(np.ascontiguousarray(a[:B]), dtype=torch.float64, device="cuda:0") for a in arrs]
    ctl.io_dtype = _L.DTYPE_F64
    p = G.make_params(ctl.robot_spec, ctl._mc, ctl.cbf_param, ctl.horizon, ctl.dt, ctl.robot_spec["radius"], ctl.io_dtype, tol=ctl.tol, max_iter=100)
    u = torch.empty((B, 2), dtype=torch.float64, device="cuda:0"); st = torch.empty((B,), dtype=torch.int32, device="cuda:0")
    it = torch.empty((B,), dtype=torch.int32, device="cuda:0"); tr = torch.zeros((B, 4096), dtype=torch.float64, device="cuda:0")
    rc = ctl._lib.sc_mpcgn_solve_batch(C.byref(p), B, ob.shape[1], X.data_ptr(), up.data_ptr(), g.data_ptr(), ob.data_ptr(), u.data_ptr(),
                                       st.data_ptr(), it.data_ptr(), tr.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    np.savez(out, tr=tr.cpu().numpy(), st=st.cpu().numpy(), it=it.cpu().numpy())
    sys.exit(0)
fam, prob, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = []
for lib in libs:
    out = os.path.join(ROOT, "gpurun_out", "trace_" + os.path.basename(lib) + ".npz")
    assert subprocess.call([sys.executable, os.path.abspath(__file__), "--child", fam, str(max(prob + 1, 64)), lib, out]) == 0
    d = np.load(out); res.append(d)
    print(os.path.basename(lib), "status", d["st"][prob], "it", d["it"][prob], " batch status hist", np.bincount(d["st"], minlength=3))
a, b = res[0]["tr"][prob].reshape(-1, 32), res[1]["tr"][prob].reshape(-1, 32)
for i in range(min(12, max(res[0]["it"][prob], res[1]["it"][prob]))):
    diff = [NAMES[j] for j in range(len(NAMES)) if a[i, j] != b[i, j]]
    print(f"--- iteration {i + 1}: differing {diff}")
    for j, nm in enumerate(NAMES):
        mark = " <<<" if a[i, j] != b[i, j] else ""
        print(f"     {nm:10s} {a[i, j]: .17e}  {b[i, j]: .17e}{mark}")
