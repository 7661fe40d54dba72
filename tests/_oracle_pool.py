"""Runs the numpy MPC oracle over a whole batch on the host cores: plain child processes (``python tests/_oracle_pool.py
in.npz out.npz``), one slice of the batch each, so nothing depends on fork/spawn semantics of a parent that may hold a HIP
context.  Test infrastructure only."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(inp, outp):
    sys.path.insert(0, ROOT)
    from oracle import mpc_cbf as M
    d = np.load(inp, allow_pickle=True)
    X, up, goal, obs = d["X"], d["up"], d["goal"], d["obs"]
    params = d["params"].item() if "params" in d.files else None
    B = X.shape[0]
    N = (params or {}).get("N", M.DEFAULTS["N"])
    u = np.zeros((B, 2)); st = np.zeros(B, dtype=np.int64); it = np.zeros(B, dtype=np.int64); z = np.zeros((B, 2 * N)); f = np.zeros(B)
    for i in range(B):
        u[i], st[i], it[i], info = M.solve(X[i], up[i], goal[i], obs[i], params=params, return_info=True)
        z[i], f[i] = info["z"], info["f"]
    np.savez(outp, u=u, st=st, it=it, z=z, f=f)


def mpc_cbf_solve_many(X, up, goal, obs, params=None, workers=None, timeout=900):
    """oracle.mpc_cbf.solve on every row; returns (u0[B,2], status[B], iters[B], z[B,n], f[B])."""
    B = X.shape[0]
    workers = max(1, min(workers or (os.cpu_count() or 2), 64, B))
    edges = np.linspace(0, B, workers + 1).astype(int)
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for w in range(workers):
            a, b = edges[w], edges[w + 1]
            inp, outp = os.path.join(tmp, f"in{w}.npz"), os.path.join(tmp, f"out{w}.npz")
            kw = dict(X=X[a:b], up=up[a:b], goal=goal[a:b], obs=obs[a:b])
            if params is not None:
                kw["params"] = np.array(params, dtype=object)
            np.savez(inp, **kw)
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), inp, outp], env=env), outp))
        parts = []
        for p, outp in procs:
            rc = p.wait(timeout=timeout)
            assert rc == 0, f"oracle worker failed with {rc}"
            parts.append(dict(np.load(outp)))
    cat = lambda k: np.concatenate([q[k] for q in parts])
    return cat("u"), cat("st"), cat("it"), cat("z"), cat("f")


if __name__ == "__main__":
    _worker(sys.argv[1], sys.argv[2])
