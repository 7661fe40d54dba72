"""VTOL2D MPC-CBF on the MULTIPLE-SHOOTING form: does a primal-dual interior point converge on feasible VTOL2D probes when the
states are variables and the dynamics are equality constraints (the form IPOPT sees through do-mpc)?  Dense KKT solves in numpy:
an experiment, not the oracle.  CPU only.
  python tools/exp_vtol_ms.py [max_iter] [probe ...]"""
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpc_vtol as V
from tools.exp_vtol import PROBES

NX, NU = 6, 4
NS = NX + NU


def evaluate(w, x0, u_prev, goal, obs, mdl, N, lam, y, derivs=True):
    spec, dt = mdl["spec"], mdl["dt"]
    K = obs.shape[0]
    Q, Rw = mdl["Q"], mdl["R"]
    a1, a2 = mdl["alpha1"], mdl["alpha2"]
    w0, w1, w2 = 1.0 - (2.0 - a1 - a2) + (1 - a1) * (1 - a2), (2.0 - a1 - a2) - 2.0, 1.0
    from oracle import mpc_cbf as M
    w0, w1, w2 = M.cbf_weights(dict(alpha1=a1, alpha2=a2))
    U = np.array([w[k * NS:k * NS + NU] for k in range(N)])
    X = np.vstack([x0[None, :], np.array([w[k * NS + NU:(k + 1) * NS] for k in range(N)])])
    n = N * NS
    xg = np.zeros(NX); xg[:2] = goal[:2]
    f = 0.0
    grad = np.zeros(n)
    W = np.zeros((n, n))
    for k in range(1, N + 1):
        e = X[k] - xg
        f += float(Q @ (e * e))
        sl = slice((k - 1) * NS + NU, k * NS)
        grad[sl] += 2.0 * Q * e
        W[sl, sl] += np.diag(2.0 * Q)
    for k in range(N):
        up = u_prev if k == 0 else U[k - 1]
        du = U[k] - up
        f += float(Rw @ (du * du))
        sk = slice(k * NS, k * NS + NU)
        grad[sk] += 2.0 * Rw * du
        W[sk, sk] += np.diag(2.0 * Rw)
        if k > 0:
            sp = slice((k - 1) * NS, (k - 1) * NS + NU)
            grad[sp] -= 2.0 * Rw * du
            W[sp, sp] += np.diag(2.0 * Rw)
            W[sk, sp] -= np.diag(2.0 * Rw); W[sp, sk] -= np.diag(2.0 * Rw)
    # equalities c_k = F(x_k, u_k) - x_{k+1};  columns of stage k's (x_k, u_k): x_k sits in stage k-1's block
    c = np.zeros(N * NX)
    A = np.zeros((N * NX, n))
    rows_g, rows_J = [], []

    def cols(k):
        """column indices of (x_k, u_k) in w (x_0 is data: -1)."""
        cx = list(range((k - 1) * NS + NU, k * NS)) if k > 0 else [-1] * NX
        return np.array(cx + list(range(k * NS, k * NS + NU)))
    for k in range(N):
        xn, Ak, Bk = V.vt_F(X[k], U[k], spec, dt, True)
        c[k * NX:(k + 1) * NX] = xn - X[k + 1]
        ck = cols(k)
        Jk = np.hstack([Ak, Bk])
        for i in range(NX):
            for j, cj in enumerate(ck):
                if cj >= 0:
                    A[k * NX + i, cj] = Jk[i, j]
            A[k * NX + i, k * NS + NU + i] = -1.0
        if derivs:
            Hk = V.vt_H(X[k], U[k], spec, dt, y[k * NX:(k + 1) * NX])
            for a, ca in enumerate(ck):
                if ca < 0:
                    continue
                for b, cb in enumerate(ck):
                    if cb >= 0:
                        W[ca, cb] += Hk[a, b]
        # CBF rows of stage k (Gauss-Newton in the barrier points; the far-obstacle probes carry no multiplier here)
        y1, S1x, S1u = V.vt_S(X[k], U[k], spec, dt, True)
        y2, S2x, S2u = V.vt_S(y1, U[k], spec, dt, True)
        D1 = np.hstack([S1x, S1u])
        D2 = np.hstack([S2x @ S1x, S2x @ S1u + S2u])
        P0 = np.zeros((2, NS)); P0[0, 0] = P0[1, 1] = 1.0
        for j in range(K):
            d = mdl["radius"] + obs[j, 2]
            hs, gs = [], []
            for (pt, Dp) in ((X[k][:2], P0), (y1[:2], D1[:2]), (y2[:2], D2[:2])):
                e = pt - obs[j, :2]
                hs.append(e @ e - mdl["beta"] * d * d)
                gs.append(2.0 * e @ Dp)
            gval = w0 * hs[0] + w1 * hs[1] + w2 * hs[2]
            grow = w0 * gs[0] + w1 * gs[1] + w2 * gs[2]
            r = np.zeros(n)
            for a, ca in enumerate(ck):
                if ca >= 0:
                    r[ca] = grow[a]
            rows_g.append(gval); rows_J.append(r)
            if derivs:
                lj = lam[len(rows_g) - 1]
                for (wp, Dp) in ((w0, P0), (w1, D1[:2]), (w2, D2[:2])):
                    Hc = -lj * wp * 2.0 * Dp.T @ Dp
                    for a, ca in enumerate(ck):
                        if ca < 0:
                            continue
                        for b, cb in enumerate(ck):
                            if cb >= 0:
                                W[ca, cb] += Hc[a, b]
    # state bounds on x_{k+1}, input box on u_k
    for k in range(N):
        for (idx, lo, hi) in mdl["xb"]:
            col = k * NS + NU + idx
            if np.isfinite(hi):
                r = np.zeros(n); r[col] = -1.0; rows_g.append(hi - w[col]); rows_J.append(r)
            if np.isfinite(lo):
                r = np.zeros(n); r[col] = 1.0; rows_g.append(w[col] - lo); rows_J.append(r)
    for k in range(N):
        for j in range(NU):
            col = k * NS + j
            r = np.zeros(n); r[col] = -1.0; rows_g.append(mdl["u_hi"][j] - w[col]); rows_J.append(r)
            r = np.zeros(n); r[col] = 1.0; rows_g.append(w[col] - mdl["u_lo"][j]); rows_J.append(r)
    return dict(f=f, grad=grad, W=W, c=c, A=A, g=np.array(rows_g), J=np.array(rows_J))


def inertia_ok(Kmat, n, me):
    try:
        _, D, _ = sla.ldl(Kmat)
    except Exception:
        return False
    ev = np.linalg.eigvalsh(D)
    return int(np.sum(ev < 0)) == me and int(np.sum(ev > 0)) == n


def solve(x0, u_prev, goal, obs, N=30, max_iter=100, tol=1e-6, verbose=False, init="rollout"):
    mdl = V.vtol_model()
    spec, dt = mdl["spec"], mdl["dt"]
    lo, hi = mdl["u_lo"], mdl["u_hi"]
    u0 = np.clip(u_prev, lo + 0.005 * (hi - lo), hi - 0.005 * (hi - lo))
    w = np.zeros(N * NS)
    x = x0.copy()
    for k in range(N):
        w[k * NS:k * NS + NU] = u0
        if init == "rollout":
            x = V.vt_F(x, u0, spec, dt)
        w[k * NS + NU:(k + 1) * NS] = x
    n, me = N * NS, N * NX
    ev = evaluate(w, x0, u_prev, goal, obs, mdl, N, None, np.zeros(me), derivs=False)
    sf = min(1.0, 100.0 / max(1e-12, np.max(np.abs(ev["grad"]))))
    g = ev["g"]
    m = g.shape[0]
    mu = 0.1
    s = np.maximum(g, 1e-2)
    lam = mu / s
    y = np.zeros(me)
    nu, tau, delta_last = 10.0, 0.995, 0.0
    status = 2
    for it in range(1, max_iter + 1):
        ev = evaluate(w, x0, u_prev, goal, obs, mdl, N, lam / sf, y / sf)
        f, grad, W, c, A, g, J = sf * ev["f"], sf * ev["grad"], sf * ev["W"], ev["c"], ev["A"], ev["g"], ev["J"]
        r_p = g - s
        r_d = grad - J.T @ lam + A.T @ y
        e_opt = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(c)), np.max(np.abs(s * lam)))
        e_mu = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(c)), np.max(np.abs(s * lam - mu)))
        if e_opt <= tol:
            status = 0
            break
        while e_mu <= 10.0 * mu and mu > 1e-9:
            mu = max(1e-9, min(0.2 * mu, mu ** 1.5))
            e_mu = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(c)), np.max(np.abs(s * lam - mu)))
        sig = lam / s
        dl0 = -sig * r_p - lam + mu / s
        Mb = W + J.T @ (sig[:, None] * J)
        rhs = np.concatenate([-grad + J.T @ (lam + dl0), -c])
        delta = 0.0
        for _try in range(40):
            Kmat = np.block([[Mb + delta * np.eye(n), A.T], [A, np.zeros((me, me))]])
            if inertia_ok(Kmat, n, me):
                break
            delta = max(1e-4, delta_last / 3.0) if delta == 0.0 else delta * 8.0
        if delta > 0:
            delta_last = delta
        sol = np.linalg.solve(Kmat, rhs)
        dw, yn = sol[:n], sol[n:]
        dlam = dl0 - sig * (J @ dw)
        ds = J @ dw + r_p
        ap = min(1.0, *( -tau * s[ds < 0] / ds[ds < 0])) if np.any(ds < 0) else 1.0
        ad = min(1.0, *( -tau * lam[dlam < 0] / dlam[dlam < 0])) if np.any(dlam < 0) else 1.0

        def merit(wt, st):
            e0 = evaluate(wt, x0, u_prev, goal, obs, mdl, N, None, y, derivs=False)
            return sf * e0["f"] - mu * np.sum(np.log(st)) + nu * (np.sum(np.abs(e0["g"] - st)) + np.sum(np.abs(e0["c"])))
        srp = float(np.sum(np.abs(r_p)) + np.sum(np.abs(c)))
        dbar = float(grad @ dw - mu * np.sum(ds / s))
        if dbar - nu * srp >= 0 and srp > 0:
            nu = dbar / (0.9 * srp)
        nu = max(nu, 1.1 * max(np.max(np.abs(lam + dlam)), np.max(np.abs(yn))) if srp > 1e-12 else nu)
        dphi = dbar - nu * srp
        phi0 = f - mu * np.sum(np.log(s)) + nu * srp
        alpha, ok = ap, False
        for _h in range(12):
            st = s + alpha * ds
            if merit(w + alpha * dw, st) <= phi0 + 1e-4 * alpha * dphi + 1e-13 * abs(phi0):
                ok = True
                break
            alpha *= 0.5
        if verbose:
            print(f"{it:3d} e {e_opt:.2e} rd {np.max(np.abs(r_d)):.2e} rp {np.max(np.abs(r_p)):.2e} c {np.max(np.abs(c)):.2e} mu {mu:.1e} "
                  f"del {delta:.1e} a {alpha if ok else 0:.3g} ap {ap:.3g} ad {ad:.3g} dw {np.max(np.abs(dw)):.2e} nu {nu:.1e}", flush=True)
        if not ok:
            break
        w = w + alpha * dw
        s = s + alpha * ds
        lam = lam + min(ad, 1.0) * dlam
        y = y + alpha * (yn - y)
        lam = np.clip(lam, mu / (1e10 * s), 1e10 * mu / s)
    return w[:NU], status, it, dict(err=e_opt, f=ev["f"], w=w)


if __name__ == "__main__":
    mi = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    only = [a for a in sys.argv[2:] if not a.startswith("-")] or list(PROBES)
    verbose = "-v" in sys.argv
    for name in only:
        x0, goal, obs = PROBES[name]
        t = time.time()
        u, st, it, info = solve(x0, np.array([0.5, 0.5, 0.3, 0.0]), goal, obs, max_iter=mi, verbose=verbose)
        print(f"{name:9s} MS status {st} it {it} err {info['err']:.2e} f {info['f']:.4f} u0 {np.round(u, 4)}  {time.time() - t:.0f}s", flush=True)
