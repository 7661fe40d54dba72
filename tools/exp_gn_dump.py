"""GPU: the LDS image of KinematicBicycle2D MPC-CBF problems after max_iter = 1, 2 iterations from two -DSC_GN_DUMP=4096 builds of
mpc_gn.hip (tools/build_variants.sh): which arrays part first between a good and a miscompiled build?
    python3 tools/exp_gn_dump.py exp_libs/lib_good.so exp_libs/lib_bad.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
B, DUMP = 16, 4096
# carve_gn<4, 10, 2, 3> for N = 10, K = 8, circles: (name, length) in order
N, K, NX, n = 10, 8, 4, 20
m = N * K + 2 * N + 2 * n
RS = 6
LAYOUT = [("cq", 12), ("xg", NX), ("up", 2), ("z", n), ("zt", n), ("zb", n), ("dz", n), ("gs", n), ("rd", n), ("rhs", n), ("xs", (N + 1) * NX),
          ("Ph", (N + 1) * NX * n), ("pts", RS * N), ("y", RS * N), ("pdz", RS * N), ("G", RS * N * n), ("obs", 7 * K), ("hk", 3 * N * K),
          ("dh", 2 * 3 * N * K), ("g", m), ("s", m), ("lam", m), ("ds", m), ("dlam", m), ("tel", N * K), ("Psi", RS * RS * N), ("M", n * n),
          ("Hk", 10 * N), ("T/L/vb", RS * N * n)]
if sys.argv[1] == "--child":
    lib, out = sys.argv[2], sys.argv[3]
    from safe_control_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(lib)
    import ctypes as C
    import torch
    from safe_control_amd.position_control import mpc_cbf_gn as G
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_tail as E
    E.B = B
    res = {}
    for mi in (1, 2):
        ctl, arrs = E.make("kb", mi)
        ctl.io_dtype = _L.DTYPE_F64
        X, up, g, ob = [torch.tensor(np.ascontiguousarray(a[:B]), dtype=torch.float64, device="cuda:0") for a in arrs]
        p = G.make_params(ctl.robot_spec, ctl._mc, ctl.cbf_param, ctl.horizon, ctl.dt, ctl.robot_spec["radius"], ctl.io_dtype, tol=ctl.tol, max_iter=mi)
        u = torch.empty((B, 2), dtype=torch.float64, device="cuda:0"); st = torch.empty((B,), dtype=torch.int32, device="cuda:0")
        it = torch.empty((B,), dtype=torch.int32, device="cuda:0"); dump = torch.zeros((B, DUMP), dtype=torch.float64, device="cuda:0")
        rc = ctl._lib.sc_mpcgn_solve_batch(C.byref(p), B, ob.shape[1], X.data_ptr(), up.data_ptr(), g.data_ptr(), ob.data_ptr(), u.data_ptr(),
                                           st.data_ptr(), it.data_ptr(), dump.data_ptr(), None)
        torch.cuda.synchronize()
        assert rc == 0
        res[f"d{mi}"] = dump.cpu().numpy()
    np.savez(out, **res)
    sys.exit(0)
outs = []
for lib in sys.argv[1:3]:
    out = os.path.join(ROOT, "gpurun_out", "dump_" + os.path.basename(lib) + ".npz")
    assert subprocess.call([sys.executable, os.path.abspath(__file__), "--child", lib, out]) == 0
    outs.append(np.load(out))
a, b = outs
for mi in (1, 2):
    A, Bb = a[f"d{mi}"], b[f"d{mi}"]
    print(f"=== after max_iter = {mi}")
    o = 0
    for name, ln in LAYOUT:
        x, y = A[:, o:o + ln], Bb[:, o:o + ln]
        same = (x == y) | (np.isnan(x) & np.isnan(y))
        nd = int((~same).sum())
        if nd:
            idx = np.argwhere(~same)[:4]
            print(f"   {name:8s} differs in {nd:6d} of {x.size} entries; e.g. " + ", ".join(f"[p{i},{j}] {x[i, j]:.6g} vs {y[i, j]:.6g}" for i, j in idx))
        else:
            print(f"   {name:8s} equal")
        o += ln
