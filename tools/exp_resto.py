"""Experiment driver (CPU, oracle only): status / iteration statistics of oracle.mpc_cbf.solve over the first n problems of
the bench batches, and an independent phase-1 (scipy L-BFGS-B on sum min(g,0)^2 over the input box) on every problem the
solver labels infeasible.   python tools/exp_resto.py du 400 [key=value ...]"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpc_cbf as M, mpc_gn as G, mpc_kb_state as S, mpc_lin as L   # noqa: E402
from safe_control_amd import workloads as W   # noqa: E402


def batch(fam, B=4096, K=8, seed=0):
    """The inputs of bench.py's legs (workloads.mpc_family_batch), rounded to f32 like the device arrays."""
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    return tuple(f32(a) for a in W.mpc_family_batch(fam, B, K, seed))


def problem(fam, over):
    """(params dict P, evaluate function) of one family."""
    if fam == "du":
        P = dict(M.DEFAULTS); P.update(over); return P, M.evaluate
    if fam == "kb":
        return G.params(G.kb_model(), 10, **over), G.evaluate
    if fam == "di":
        return G.params(G.di_model(), 10, **over), G.evaluate
    if fam == "quad2d":
        return G.params(G.quad2d_model(), 10, **over), G.evaluate
    if fam in ("c3bf", "dpcbf"):
        mdl = S.c3bf_model() if fam == "c3bf" else S.dpcbf_model()
        P = S.params(mdl, 10, **over); P["model"] = dict(mdl, circles_only=True); return P, S.evaluate
    mdl = L.quad3d_model() if fam == "quad3d" else L.si_model()
    return L.params(mdl, 10, **over), L.evaluate


def one(a):
    fam, i, x, up, g, o, over, phase1 = a
    P, ev = problem(fam, over)
    t0 = time.time()
    u, st, it, info = M.solve(x, up, g, o, params=P, return_info=True, evaluate_fn=ev)
    dt = time.time() - t0
    found = None
    if phase1 and st != M.STATUS_OPTIMAL:
        found = phase_one(fam, x, up, g, o, P, ev, info)
    return i, st, it, info["n_resto"], info["theta"], float(np.min(info["g"])), found, dt, u


def phase_one(fam, x, up, g, o, P, ev, info, starts=5):
    """Independent feasibility search: min sum min(g,0)^2 over the input box by L-BFGS-B from the solver's point, the initial
    guess and random points.  Returns the best min g found (>= -1e-9 means a feasible plan exists)."""
    from scipy.optimize import minimize
    obs = info["obs"]
    nz = info["z"].shape[0]
    N = P["N"]
    if "u_hi" in P:
        lo, hi = np.tile(np.asarray(P["u_lo"], float), N), np.tile(np.asarray(P["u_hi"], float), N)
    else:
        hi = np.tile([P["a_max"], P["w_max"]], N); lo = -hi
    m_el = info["g"].shape[0] - 2 * nz

    def fun(z):
        e = ev(x, z, up, g, obs, P, None, level=1)
        v = np.minimum(e["g"][:m_el], 0.0)
        return float(v @ v), 2.0 * e["J"][:m_el].T @ v
    rng = np.random.default_rng(1)
    best = -np.inf
    z0s = [info["z"], np.clip(np.tile(up, N), lo, hi)] + [rng.uniform(lo, hi) for _ in range(starts - 2)]
    for z0 in z0s:
        r = minimize(fun, np.clip(z0, lo, hi), jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)), options=dict(maxiter=500, ftol=1e-16, gtol=1e-12))
        gm = float(np.min(ev(x, r.x, up, g, obs, P, None, level=0)["g"][:m_el]))
        best = max(best, gm)
        if best >= -1e-9:
            break
    return best


if __name__ == "__main__":
    fam, n = sys.argv[1], int(sys.argv[2])
    over, phase1, start = {}, False, 0
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        if k == "phase1": phase1 = bool(int(v))
        elif k == "start": start = int(v)
        else: over[k] = float(v) if "." in v or "e" in v else int(v)
    X, up, g, o = batch(fam)
    t0 = time.time()
    with Pool(8) as pool:
        res = pool.map(one, [(fam, i, X[i], up[i], g[i], o[i], over, phase1) for i in range(start, start + n)], chunksize=4)
    st = np.array([r[1] for r in res]); it = np.array([r[2] for r in res]); nr = np.array([r[3] for r in res])
    th = np.array([r[4] for r in res])
    print(f"{fam}: n={n} optimal {np.mean(st == 0):.3f} infeasible {np.mean(st == 1):.3f} inaccurate {np.mean(st == 2):.3f}  "
          f"iters mean {it.mean():.1f} max {it.max()}  resto entered on {np.mean(nr > 0):.3f}  wall {time.time() - t0:.0f}s")
    for s_ in (1, 2):
        sel = st == s_
        if sel.any():
            print(f"  status {s_}: iters mean {it[sel].mean():.1f}  theta min/median/max {th[sel].min():.2e} {np.median(th[sel]):.2e} {th[sel].max():.2e}")
    if phase1:
        bad = [(r[0], r[1], r[6]) for r in res if r[1] == 1 and r[6] is not None and r[6] >= -1e-9]
        print(f"  labelled infeasible but phase-1 found a feasible plan: {len(bad)}", bad[:20])
        inacc = [(r[0], r[6]) for r in res if r[1] == 2 and r[6] is not None]
        print(f"  inaccurate: {len(inacc)}, of which phase-1 feasible {sum(1 for _, b in inacc if b >= -1e-9)}")
    if os.environ.get("DUMP"):
        np.savez(os.environ["DUMP"], st=st, it=it, nr=nr, th=th, u=np.array([r[8] for r in res]))
