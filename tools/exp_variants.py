"""GPU: one family's bench batch (first B problems) through every variant library given on the command line (tools/build_variants.sh):
status histogram, iteration statistics and agreement with the first variant.
    python3 tools/exp_variants.py FAMILY B exp_libs/lib_A.so exp_libs/lib_B.so ...
Each library is loaded in a child process (the ctypes handle of the parent would be the first one for all)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    fam, B, lib, out = sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    from safe_control_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(lib)
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_tail as E
    E.B = B
    ctl, arrs = E.make(fam, 100)
    X, up, g, ob = [E.t32(a[:B]) for a in arrs]
    u, st, it = ctl.solve(X, up, g, ob)[:3]
    torch.cuda.synchronize()
    np.savez(out, u=u.cpu().numpy(), st=st.cpu().numpy(), it=it.cpu().numpy())
    sys.exit(0)

fam, B, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
ref = None
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for lib in libs:
    out = os.path.join(ROOT, "gpurun_out", "var_" + os.path.basename(lib) + ".npz")
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", fam, str(B), lib, out])
    if rc != 0:
        print(f"{lib}: child failed ({rc})"); continue
    d = np.load(out)
    st, it, u = d["st"], d["it"], d["u"]
    line = f"{os.path.basename(lib):28s} status 0/1/2 = {[(int((st == s).sum())) for s in (0, 1, 2)]}  it mean {it.mean():6.2f} max {it.max():4d}"
    if ref is None:
        ref = (st, it, u)
    else:
        line += f"   vs first: status equal {int((st == ref[0]).sum())}/{B}, iters equal {int((it == ref[1]).sum())}, max |du| {np.nanmax(np.abs(u - ref[2])):.2e}"
    print(line, flush=True)
