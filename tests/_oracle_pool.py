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


def _worker_od_vtol(d, outp):
    """kind = "odvtol": oracle.od_mpc_vtol.solve (optimal-decay MPC-CBF of VTOL2D); zz = [u (4 N) | rho (2 N)]."""
    from oracle import od_mpc_vtol as OV
    X, up, goal, obs = d["X"], d["up"], d["goal"], d["obs"]
    over = d["params"].item() if "params" in d.files else {}
    over = dict(over or {})
    N = over.pop("N", 30)
    B = X.shape[0]
    u = np.zeros((B, 4)); st = np.zeros(B, dtype=np.int64); it = np.zeros(B, dtype=np.int64)
    z = np.zeros((B, 4 * N)); rho = np.zeros((B, 2 * N)); f = np.zeros(B); gmin = np.zeros(B); err = np.zeros(B); lam = np.zeros(B)
    for i in range(B):
        u[i], _, st[i], it[i], info = OV.solve(X[i], up[i], goal[i], obs[i], N=N, params_over=over, return_info=True)
        z[i], rho[i], f[i], gmin[i], err[i], lam[i] = info["zz"][: 4 * N], info["zz"][4 * N:], info["f"], info["g"].min(), info["err"], info["lam"].max()
    np.savez(outp, u=u, st=st, it=it, z=z, rho=rho, f=f, gmin=gmin, err=err, lam=lam)


def od_vtol_solve_many(X, up, goal, obs, params=None, workers=None, timeout=1800):
    """oracle.od_mpc_vtol.solve on every row; dict(u, st, it, z, rho, f, gmin, err, lam)."""
    return _run(dict(kind=np.array("odvtol")), X, up, goal, obs, params, workers, timeout)


def family_problem(family, N=10, over=None):
    """(params, evaluate function) of one model family of workloads.MPC_FAMILIES for oracle.mpc_cbf.solve."""
    from oracle import mpc_cbf as M, mpc_gn as G, mpc_kb_state as S, mpc_lin as L
    over = dict(over or {})
    if family == "du":
        P = dict(M.DEFAULTS, N=N); P.update(over); return P, M.evaluate
    if family in ("kb", "di", "quad2d"):
        mdl = {"kb": G.kb_model, "di": G.di_model, "quad2d": G.quad2d_model}[family]()
        return G.params(mdl, N, **over), G.evaluate
    if family == "vtol":
        from oracle import mpc_vtol as V
        return V.params(N=30 if N == 10 else N, **over), G.evaluate      # the reference's VTOL2D horizon is 30
    if family in ("c3bf", "dpcbf"):
        mdl = S.c3bf_model() if family == "c3bf" else S.dpcbf_model()
        P = S.params(mdl, N, **over); P["model"] = dict(mdl, circles_only=True); return P, S.evaluate
    mdl = L.quad3d_model() if family == "quad3d" else L.si_model()
    return L.params(mdl, N, **over), L.evaluate


def _worker_family(d, outp):
    """kind = "fam:<family>": the shared solver with that family's problem functions; also records the l1 violation of the CBF
    rows at the returned point and how often the restoration was entered."""
    from oracle import mpc_cbf as M
    fam = str(d["kind"])[4:]
    X, up, goal, obs = d["X"], d["up"], d["goal"], d["obs"]
    over = d["params"].item() if "params" in d.files else {}
    N = over.pop("N", 30 if fam == "vtol" else 10) if isinstance(over, dict) else 10
    B = X.shape[0]
    nu = up.shape[1]
    u = np.zeros((B, nu)); st = np.zeros(B, dtype=np.int64); it = np.zeros(B, dtype=np.int64); z = np.zeros((B, nu * N))
    theta = np.zeros(B); nr = np.zeros(B, dtype=np.int64); err = np.zeros(B); stalled = np.zeros(B, dtype=np.int64)
    for i in range(B):
        P, ev = family_problem(fam, N, over)
        u[i], st[i], it[i], info = M.solve(X[i], up[i], goal[i], obs[i], params=P, return_info=True, evaluate_fn=ev)
        z[i], theta[i], nr[i], err[i], stalled[i] = info["z"], info["theta"], info["n_resto"], info["err"], info.get("stalled", 0)
    np.savez(outp, u=u, st=st, it=it, z=z, theta=theta, n_resto=nr, err=err, stalled=stalled)


def _worker_phase1(d, outp):
    """kind = "p1:<family>": an independent feasibility search (tests/test_oracle_mpc_resto.py: phase_one) from the plan `z` a solver
    returned: best min_i g_i over the CBF rows it reaches."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_mpc_resto import phase_one
    fam = str(d["kind"])[3:]
    X, up, goal, obs, Z = d["X"], d["up"], d["goal"], d["obs"], d["z"]
    over = d["params"].item() if "params" in d.files else {}
    starts = int((over or {}).get("starts", 8))
    best = np.zeros(X.shape[0])
    for i in range(X.shape[0]):
        P, ev = family_problem(fam, 10, {})
        info = dict(z=Z[i], obs=obs[i])
        best[i] = phase_one(X[i], up[i], goal[i], P, ev, info, starts=starts)
    np.savez(outp, best=best)


def phase_one_many(family, X, up, goal, obs, z, starts=8, workers=None, timeout=3000):
    """phase_one on every row, from the plans `z`; returns best min g per row."""
    return _run(dict(kind=np.array("p1:" + family), z=z), X, up, goal, obs, dict(starts=starts), workers, timeout)["best"]


def family_solve_many(family, X, up, goal, obs, params=None, workers=None, timeout=1800):
    """oracle.mpc_cbf.solve with the problem functions of `family` on every row; dict(u, st, it, z, theta, n_resto, err, stalled)."""
    return _run(dict(kind=np.array("fam:" + family)), X, up, goal, obs, params, workers, timeout)


def _worker(inp, outp):
    sys.path.insert(0, ROOT)
    from oracle import mpc_cbf as M
    d = np.load(inp, allow_pickle=True)
    if "kind" in d.files and str(d["kind"]) == "odvtol":
        return _worker_od_vtol(d, outp)
    if "kind" in d.files and str(d["kind"]).startswith("od_"):
        return _worker_od_rd1(d, outp)
    if "kind" in d.files and str(d["kind"]).startswith("p1:"):
        return _worker_phase1(d, outp)
    if "kind" in d.files and str(d["kind"]).startswith("fam:"):
        return _worker_family(d, outp)
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
            per_row = {k: (v[a:b] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B else v) for k, v in extra.items()}
            kw = dict(per_row, X=X[a:b], up=up[a:b], goal=goal[a:b], obs=obs[a:b])
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
