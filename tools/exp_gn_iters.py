"""GPU: the plan z after max_iter = 1, 2, 3, .. iterations from two builds of the library (no code change between the runs: the
iteration limit is a run-time parameter) -- where does a miscompiled build part from a good one?
    python3 tools/exp_gn_iters.py FAMILY exp_libs/lib_good.so exp_libs/lib_bad.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
B = 64
if sys.argv[1] == "--child":
    fam, lib, out = sys.argv[2], sys.argv[3], sys.argv[4]
    from safe_control_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(lib)
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_tail as E
    E.B = B
    res = {}
    for mi in (1, 2, 3, 4, 5, 6, 8, 10, 15, 100):
        ctl, arrs = E.make(fam, mi)
        ctl.io_dtype = _L.DTYPE_F64
        X, up, g, ob = [torch.tensor(np.ascontiguousarray(a[:B]), dtype=torch.float64, device="cuda:0") for a in arrs]
        u, st, it, z = ctl.solve(X, up, g, ob, want_z=True)
        torch.cuda.synchronize()
        res[f"z{mi}"] = z.cpu().numpy(); res[f"st{mi}"] = st.cpu().numpy(); res[f"it{mi}"] = it.cpu().numpy()
    np.savez(out, **res)
    sys.exit(0)
fam, libs = sys.argv[1], sys.argv[2:4]
outs = []
for lib in libs:
    out = os.path.join(ROOT, "gpurun_out", "iters_" + os.path.basename(lib) + ".npz")
    assert subprocess.call([sys.executable, os.path.abspath(__file__), "--child", fam, lib, out]) == 0
    outs.append(np.load(out))
a, b = outs
for mi in (1, 2, 3, 4, 5, 6, 8, 10, 15, 100):
    za, zb = a[f"z{mi}"], b[f"z{mi}"]
    bad = np.nonzero((za != zb).any(axis=1))[0]
    print(f"max_iter {mi:3d}: z differs on {len(bad):3d}/{B} problems; max |dz| {np.nanmax(np.abs(za - zb)):.3e}; status {np.bincount(a[f'st{mi}'], minlength=3)} vs {np.bincount(b[f'st{mi}'], minlength=3)}; "
          f"iters equal {int((a[f'it{mi}'] == b[f'it{mi}']).sum())}")
    if len(bad) and mi <= 3:
        i = bad[0]
        print("   problem", i, "good z", za[i][:6], "\n              bad z", zb[i][:6])
