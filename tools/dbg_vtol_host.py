"""The lane-per-problem VTOL2D solver compiled for the host (tools/vtol_host.cpp) against oracle/mpc_vtol.py on the first n problems of
the vtol workload batch and on the probes of tools/exp_vtol.py.  Debugging aid, CPU only.   python tools/dbg_vtol_host.py [n]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mpc_vtol as V
from safe_control_amd import _lib, workloads as W
from safe_control_amd.position_control import mpc_cbf_vtol as PV
from safe_control_amd.robots.spec import complete_robot_spec

SO = "/tmp/libvtol_host.so"


def build():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "vtol_host.cpp"), "-o", SO])
    lib = C.CDLL(SO)
    lib.vtol_host_solve.restype = C.c_int
    return lib


def host_solve(lib, x0, up, goal, obs, v_max=None, **over):
    spec = complete_robot_spec({"model": "VTOL2D"} if v_max is None else {"model": "VTOL2D", "v_max": v_max})
    p = PV.make_params(spec, PV.CBF_VTOL, 30, 0.05, spec["radius"], _lib.DTYPE_F64, **over)
    K = obs.shape[0]
    u = np.zeros(4); z = np.zeros(120); st = C.c_int(0); it = C.c_int(0)
    x0, up, goal, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (x0, up, goal, obs))
    lib.vtol_host_solve(C.byref(p), K, x0.ctypes.data_as(C.c_void_p), up.ctypes.data_as(C.c_void_p), goal.ctypes.data_as(C.c_void_p),
                        obs.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), C.byref(st), C.byref(it))
    return u, st.value, it.value, z


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    lib = build()
    X, up, goal, obs = W.mpc_family_batch("vtol", 64, 8, seed=0)
    for i in range(n):
        t = time.time()
        uh, sh, ih, zh = host_solve(lib, X[i], up[i], goal[i], obs[i])
        th = time.time() - t
        t = time.time()
        uo, so, io, info = V.solve(X[i], up[i], goal[i], obs[i], return_info=True)
        print(f"#{i}: host status {sh} it {ih} ({th * 1e3:.0f} ms) | oracle status {so} it {io} ({time.time() - t:.0f} s) | "
              f"|du0| {np.abs(uh - uo).max():.2e} |dz| {np.abs(zh - info['z']).max():.2e}", flush=True)
