"""ctypes wrapper of oracle/c/*.c (float64 C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_cbfqp_batch.restype = C.c_int
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def cbfqp_batch(model, X, u_ref, obs, spec, cbf_param, dt=0.05, cbf_mode="cbf", n_obs=None, n_threads=1):
    """Batch CBF-QP in C double.  Same argument meaning as oracle.cbf_qp.solve_batch."""
    from . import cbf_qp, robots as R
    lib = load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    u_ref = np.ascontiguousarray(u_ref, dtype=np.float64)
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    B, K = X.shape[0], obs.shape[-2]
    shared = 1 if obs.ndim == 2 else 0
    lo, hi = cbf_qp.input_bounds(model, spec)
    lo = np.ascontiguousarray(lo, dtype=np.float64); hi = np.ascontiguousarray(hi, dtype=np.float64)
    a1 = cbf_param["alpha1"] if model in R.REL_DEG2 else cbf_param["alpha"]
    a2 = cbf_param.get("alpha2", 0.0)
    u = np.empty((B, 2)); st = np.empty(B, dtype=np.int32); h = np.empty((B, K))
    nptr = None
    if n_obs is not None:
        n_obs = np.ascontiguousarray(n_obs, dtype=np.int32)
        nptr = n_obs.ctypes.data_as(C.c_void_p)
    rc = lib.oracle_cbfqp_batch(
        C.c_int(model), C.c_long(B), C.c_int(K), X.ctypes.data_as(C.c_void_p), u_ref.ctypes.data_as(C.c_void_p),
        obs.ctypes.data_as(C.c_void_p), C.c_int(shared), nptr, C.c_double(spec["radius"]), C.c_double(dt),
        C.c_double(a1), C.c_double(a2), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
        C.c_double(spec.get("rear_ax_dist", 0.0)), C.c_int(1 if cbf_mode == "hard" else 0),
        u.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), C.c_int(n_threads),
        C.c_int(X.shape[1]), C.c_double(spec.get("mass", 1.0)))
    if rc != 0:
        raise ValueError("oracle_cbfqp_batch: bad arguments")
    return u, st, h
