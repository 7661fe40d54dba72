"""Runs the numpy MPC oracle over a whole batch on the host cores: plain child processes (``python tests/_oracle_pool.py
in.npz out.npz``), one slice of the batch each, so nothing depends on fork/spawn semantics of a parent that may hold a HIP
context.  Test infrastructure only."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker_od_rd1(d, outp):
    """kind = "od_uni" | "od_quad3d": oracle.od_mpc_rd1.solve (the config-5 extension)."""
    from oracle import mpc_lin as L, od_mpc_rd1 as O
    X, up, goal, obs = d["X"], d["up"], d["goal"], d["obs"]
    over = d["params"].item() if "params" in d.files else {}
    N = over.get("N", 10)
    if str(d["kind"]) == "od_uni":
        P = O.uni_params(**over)
        nu = 2
    else:
        over = dict(over); over.pop("N", None)
        P = O.lin_params(dict(L.quad3d_model(), circles_only=False), N=N, **over)
        nu = 4
    B = X.shape[0]
    u = np.zeros((B, nu)); st = np.zeros(B, dtype=np.int64); it = np.zeros(B, dtype=np.int64)
    z = np.zeros((B, nu * N)); rho = np.zeros((B, N)); f = np.zeros(B)
    for i in range(B):
        u[i], _, st[i], it[i], info = O.solve(X[i], up[i], goal[i], obs[i], P, return_info=True)
        z[i], rho[i], f[i] = info["z"], info["rho"], info["f"]
    np.savez(outp, u=u, st=st, it=it, z=z, rho=rho, f=f)


def _worker(inp, outp):
    sys.path.insert(0, ROOT)
    from oracle import mpc_cbf as M
    d = np.load(inp, allow_pickle=True)
    if "kind" in d.files and str(d["kind"]).startswith("od_"):
        return _worker_od_rd1(d, outp)
    X, up, goal, obs = d["X"], d["up"], d["goal"], d["obs"]
    params = d["params"].item() if "params" in d.files else None
    B = X.shape[0]
    N = (params or {}).get("N", M.DEFAULTS["N"])
    u = np.zeros((B, 2)); st = np.zeros(B, dtype=np.int64); it = np.zeros(B, dtype=np.int64); z = np.zeros((B, 2 * N)); f = np.zeros(B)
    for i in range(B):
        u[i], st[i], it[i], info = M.solve(X[i], up[i], goal[i], obs[i], params=params, return_info=True)
        z[i], f[i] = info["z"], info["f"]
    np.savez(outp, u=u, st=st, it=it, z=z, f=f)


def od_rd1_solve_many(kind, X, up, goal, obs, params=None, workers=None, timeout=1800):
    """oracle.od_mpc_rd1.solve on every row (kind "od_uni" | "od_quad3d"); returns dict(u, st, it, z, rho, f)."""
    return _run(dict(kind=np.array(kind)), X, up, goal, obs, params, workers, timeout)


def mpc_cbf_solve_many(X, up, goal, obs, params=None, workers=None, timeout=900):
    """oracle.mpc_cbf.solve on every row; returns (u0[B,2], status[B], iters[B], z[B,n], f[B])."""
    r = _run({}, X, up, goal, obs, params, workers, timeout)
    return r["u"], r["st"], r["it"], r["z"], r["f"]


def _run(extra, X, up, goal, obs, params, workers, timeout):
    B = X.shape[0]
    workers = max(1, min(workers or (os.cpu_count() or 2), 64, B))
    edges = np.linspace(0, B, workers + 1).astype(int)
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for w in range(workers):
            a, b = edges[w], edges[w + 1]
            inp, outp = os.path.join(tmp, f"in{w}.npz"), os.path.join(tmp, f"out{w}.npz")
            kw = dict(extra, X=X[a:b], up=up[a:b], goal=goal[a:b], obs=obs[a:b])
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
    return {k: np.concatenate([q[k] for q in parts]) for k in parts[0]}


if __name__ == "__main__":
    _worker(sys.argv[1], sys.argv[2])
