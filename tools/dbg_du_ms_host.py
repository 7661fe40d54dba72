"""The multiple-shooting DynamicUnicycle2D kernel's solver (csrc/mpc_du_ms_solver.hpp) compiled for the host (tools/du_ms_host.cpp: one thread per
lane) against oracle/ms_ipopt.py in KERNEL_PROFILE on config-3 draws: status, iteration count, u_0, and the per-iteration trace.  Debugging aid,
CPU only.   python tools/dbg_du_ms_host.py [n] [first] [-v]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ms_ipopt as MS
from safe_control_amd import _lib, workloads as W
from safe_control_amd.position_control import mpc_cbf as PM
from safe_control_amd.robots.spec import complete_robot_spec

SO = "/tmp/libdu_ms_host.so"
PROFILE = dict(MS.KERNEL_PROFILE)


def build():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", os.path.join(ROOT, "tools", "du_ms_host.cpp"), "-o", SO])
    lib = C.CDLL(SO)
    lib.du_ms_host_solve.restype = C.c_int
    return lib


LDS_NAMES = "OB AB H G C KG PX LAM XS US YS Pa Pb T QU FP FT FP2 FT2 SC Y0 RW XR total".split()


def lds_layout(lib, N, KS):
    out = (C.c_int * 24)()
    lib.du_ms_host_lds_layout(N, KS, out)
    return dict(zip(LDS_NAMES, list(out)))


def host_solve(lib, x0, up, goal, obs, horizon=10, spec=None, want_lds=False, **ipopt):
    sp = complete_robot_spec(dict({"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25}, **(spec or {})))    # (the config-3 robot)
    Q, R = PM.default_mpc_weights("DynamicUnicycle2D")
    p = PM.make_params(sp, PM.default_mpc_cbf_param("DynamicUnicycle2D"), Q, R, horizon, 0.05, sp["radius"], _lib.DTYPE_F64)
    ip = _lib.default_ipopt(**ipopt)
    K = obs.shape[0]
    u = np.zeros(2); plan = np.zeros((horizon + 1) * 4 + horizon * 2); tr = np.zeros((ip.max_iter + 1, 8)); st = C.c_int(0); it = C.c_int(0)
    x0, up, goal, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (x0, up, goal, obs))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L = lds_layout(lib, horizon, K)
    lds = np.zeros(L["total"])
    lib.du_ms_host_solve(C.byref(p), C.byref(ip), K, vp(x0), vp(up), vp(goal), vp(obs), vp(u), vp(plan), vp(tr), C.byref(st), C.byref(it), vp(lds))
    if want_lds:
        return u, st.value, it.value, plan, tr[: it.value + 1], {k: lds[L[k]:] for k in LDS_NAMES[:-1]}
    return u, st.value, it.value, plan, tr[: it.value + 1]


def oracle_solve(x0, up, goal, obs, spec=None, opts=None):
    tr = []
    u, st, it, info = MS.solve(MS.du_model(spec), x0, up, goal, obs, return_info=True, opts=opts or PROFILE, trace=tr)
    T = np.array([[q["E0"], q["dinf"], q["pinf"], q["comp"], q["mu"], q["theta"], q["delta"], -q["alpha"] if q["resto"] else q["alpha"]] for q in tr])
    return u, st, it, np.concatenate([info["X"].reshape(-1), info["U"].reshape(-1)]), T


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    n = int(args[0]) if len(args) > 0 else 4
    first = int(args[1]) if len(args) > 1 else 0
    verbose = "-v" in sys.argv
    lib = build()
    X, up, goal, obs = W.mpc_family_batch("du", 4096, 8, seed=0)
    bad = 0
    for i in range(first, first + n):
        t = time.time()
        uh, sh, ih, ph, th = host_solve(lib, X[i], up[i], goal[i], obs[i])
        tt = time.time() - t
        uo, so, io, po, to = oracle_solve(X[i], up[i], goal[i], obs[i])
        m = min(len(th), len(to))
        rel = np.abs(th[:m] - to[:m]) / np.maximum(1e-9, np.abs(to[:m]))
        first_off = int(np.argmax(rel.max(axis=1) > 1e-5)) if (rel.max(axis=1) > 1e-5).any() else -1
        flag = "" if (sh == so and ih == io and np.abs(uh - uo).max() < 1e-7) else "   <<<<"
        bad += bool(flag)
        print(f"#{i}: host status {sh} it {ih} ({tt:.2f} s) | oracle status {so} it {io} | |du0| {np.abs(uh - uo).max():.2e} |dplan| {np.abs(ph - po).max():.2e}"
              f" | first trace row off by > 1e-5: {first_off}{flag}", flush=True)
        if verbose or (flag and "-t" in sys.argv):
            for q in range(m):
                print("   ", q, " ".join(f"{v:10.3e}" for v in th[q]), "|", " ".join(f"{v:10.3e}" for v in to[q]))
    print(f"{bad} of {n} differ")
