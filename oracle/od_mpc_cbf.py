"""Float64 statement of the optimal-decay MPC-CBF NLP (position_control/optimal_decay_mpc_cbf.py) and a solver.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **Parity unpinned**: the reference solves this NLP
with do-mpc -> casadi -> IPOPT (optimal_decay_mpc_cbf.py:104-107, 162-172), none installable here, no
reference test pins a result, and the reference copy is stale (5-wide obstacle rows, SURVEY section 2
rows 9-10).  What is restated from the reference is the *problem*, for DynamicUnicycle2D:

  model       x+ = x + (f(x) + g(x) u) dt                                          :135-141
  decay vars  omega1_k, omega2_k: two extra inputs per stage (here rho1_k, rho2_k)  :123-124
  cost        sum (x_k - goal)' Q (x_k - goal), stage and terminal                  :147-148,174-176
              + sum_k sum_i R_i u_{k,i}^2          (an expression r-term, not do-mpc's delta-u penalty) :178-179
              + sum_k p_sb1 (rho1_k - omega1)^2 + p_sb2 (rho2_k - omega2)^2          :181-184
  weights     Q = diag(50, 50, .01, 30), R = (.5, .5), N = 10                         :25,31-33
              omega1 = omega2 = 1, p_sb1 = p_sb2 = 10                                :88-91
  CBF         dd_h + (a1 rho1 + a2 rho2) d_h + a1 a2 rho1 rho2 h >= 0  per stage    :291-297
              with DT gains a1 = a2 = 0.01                                          :59-60
  bounds      |v_k| <= v_max, |a| <= a_max, |w| <= w_max; rho free                  :194-199
  obstacles   the reference pads to 5 rows of 5 values; here K rows of the 7-wide format of MPCCBF
              ("generalised to 7-wide obstacles")

The two r-term calls of the reference (:185-186) are read as a sum (what the author wrote them for); the
barrier and the roll-out are those of oracle/mpc_cbf.py.  With w2 = 1, w1_k = s_k - 2, w0_k = 1 - s_k + q_k,
s_k = a1 rho1_k + a2 rho2_k, q_k = a1 a2 rho1_k rho2_k, a CBF row is
    w2 h(p_{k+2}) + w1_k h(p_{k+1}) + w0_k h(p_k) >= 0,
the MPC-CBF row with stage-dependent weights.

Solver: the interior-point method of oracle/mpc_cbf.py on zz = (u_0..u_{N-1} | rho_0..rho_{N-1}).  The decay
variables of a stage only meet that stage's rows, so their 2x2 blocks D_k are eliminated first (Schur
complement onto the inputs, which is the 2N x 2N system the MPC-CBF kernel already factors); a D_k that is
not positive definite is shifted, the Schur complement gets the inertia correction.  ``linear_algebra="dense"``
solves the same Newton system without the elimination (cross-check in tests/test_oracle_od_mpc.py).
"""
import math

import numpy as np

from . import mpc_cbf as M

STATUS_OPTIMAL, STATUS_INFEASIBLE, STATUS_INACCURATE = M.STATUS_OPTIMAL, M.STATUS_INFEASIBLE, M.STATUS_INACCURATE

DEFAULTS = dict(M.DEFAULTS, alpha1=0.01, alpha2=0.01, omega1=1.0, omega2=1.0, p_sb1=10.0, p_sb2=10.0)


def stage_weights(rho, P):
    """w0_k, w1_k (w2 = 1) of the stage rows; rho (N, 2)."""
    a1, a2 = P["alpha1"], P["alpha2"]
    s = a1 * rho[:, 0] + a2 * rho[:, 1]
    q = a1 * a2 * rho[:, 0] * rho[:, 1]
    return 1.0 - s + q, s - 2.0


def evaluate(x0, zz, goal, obs, P, lam=None, level=2):
    """Problem functions at zz = (z | rho).  Same levels and row order as oracle.mpc_cbf.evaluate."""
    N, dt = P["N"], P["dt"]
    n = 2 * N
    a1, a2 = P["alpha1"], P["alpha2"]
    Q, Rw = np.asarray(P["Q"], dtype=np.float64), np.asarray(P["R"], dtype=np.float64)
    pen = np.array([P["p_sb1"], P["p_sb2"]], dtype=np.float64)
    ref = np.array([P["omega1"], P["omega2"]], dtype=np.float64)
    K = obs.shape[0]
    z, rho = zz[:n], zz[n:].reshape(N, 2)
    w0, w1 = stage_weights(rho, P)
    X, p_extra = M.rollout(x0, z, P)
    pos = np.vstack([X[:, 0:2], p_extra[None, :]])
    gpos = np.asarray(goal, dtype=np.float64)[0:2]
    out = {}
    f = 0.0
    for k in range(1, N + 1):
        e = pos[k] - gpos
        f += Q[0] * e[0] ** 2 + Q[1] * e[1] ** 2 + Q[2] * X[k, 2] ** 2 + Q[3] * X[k, 3] ** 2
    Rd = np.tile(Rw, N)
    f += float(np.sum(Rd * z * z)) + float(np.sum(pen * (rho - ref) ** 2))
    hk = np.zeros((N + 2, K)); dh = np.zeros((N + 2, K, 2)); Hh = np.zeros((N + 2, K, 2, 2))
    for k in range(N + 2):
        for j in range(K):
            hk[k, j], dh[k, j], Hh[k, j] = M.barrier(pos[k], obs[j], P)
    m = N * K + 2 * N + 2 * n
    g = np.zeros(m)
    for k in range(N):
        g[k * K:(k + 1) * K] = hk[k + 2] + w1[k] * hk[k + 1] + w0[k] * hk[k]
    o = N * K
    for k in range(1, N + 1):
        g[o + 2 * (k - 1)] = P["v_max"] - X[k, 3]
        g[o + 2 * (k - 1) + 1] = P["v_max"] + X[k, 3]
    o += 2 * N
    ub = np.tile([P["a_max"], P["w_max"]], N)
    g[o:o + n] = ub - z
    g[o + n:o + 2 * n] = ub + z
    out.update(f=float(f), g=g, X=X, pts=pos)
    if level == 0:
        return out
    dP = M.position_jacobians(X, P)
    dTh = np.zeros((N + 1, n)); dV = np.zeros((N + 1, n))
    for k in range(N + 1):
        for j in range(k):
            dTh[k, 2 * j + 1] = dt
            dV[k, 2 * j] = dt
    grad = np.zeros(2 * n)
    for k in range(1, N + 1):
        grad[:n] += dP[k].T @ (2.0 * Q[0:2] * (pos[k] - gpos)) + 2.0 * Q[2] * X[k, 2] * dTh[k] + 2.0 * Q[3] * X[k, 3] * dV[k]
    grad[:n] += 2.0 * Rd * z
    grad[n:] = (2.0 * pen * (rho - ref)).reshape(-1)
    J = np.zeros((m, 2 * n))
    # d row / d rho_i = a_i (h1 - h0) + a1 a2 rho_other h0
    A1 = np.zeros((N, K)); A2 = np.zeros((N, K))
    for k in range(N):
        A1[k] = a1 * (hk[k + 1] - hk[k]) + a1 * a2 * rho[k, 1] * hk[k]
        A2[k] = a2 * (hk[k + 1] - hk[k]) + a1 * a2 * rho[k, 0] * hk[k]
        for j in range(K):
            J[k * K + j, :n] = dh[k + 2, j] @ dP[k + 2] + w1[k] * dh[k + 1, j] @ dP[k + 1] + w0[k] * dh[k, j] @ dP[k]
            J[k * K + j, n + 2 * k] = A1[k, j]
            J[k * K + j, n + 2 * k + 1] = A2[k, j]
    o = N * K
    for k in range(1, N + 1):
        J[o + 2 * (k - 1), :n] = -dV[k]
        J[o + 2 * (k - 1) + 1, :n] = dV[k]
    o += 2 * N
    J[o:o + n, :n] = -np.eye(n)
    J[o + n:o + 2 * n, :n] = np.eye(n)
    out.update(grad=grad, J=J)
    if level == 1:
        return out
    lam = np.zeros(m) if lam is None else lam
    lc = lam[: N * K].reshape(N, K)
    mu = np.zeros((N + 2, K))                                               # multipliers touching position k
    for k in range(N + 2):
        if k - 2 >= 0: mu[k] += lc[k - 2]
        if 1 <= k <= N: mu[k] += w1[k - 1] * lc[k - 1]
        if k <= N - 1: mu[k] += w0[k] * lc[k]
    W = np.zeros((2 * n, 2 * n))
    Wzz = np.diag(2.0 * Rd)
    q = np.zeros((N + 2, 2))
    for k in range(N + 2):
        Om = -np.einsum("j,jab->ab", mu[k], Hh[k])
        qk = -mu[k] @ dh[k]
        if 1 <= k <= N:
            Om = Om + np.diag(2.0 * Q[0:2])
            qk = qk + 2.0 * Q[0:2] * (pos[k] - gpos)
            Wzz += 2.0 * Q[2] * np.outer(dTh[k], dTh[k]) + 2.0 * Q[3] * np.outer(dV[k], dV[k])
        q[k] = qk
        Wzz += dP[k].T @ Om @ dP[k]
    for i in range(N + 1):
        qbar = q[i + 1:].sum(axis=0)
        th, v = X[i, 2], X[i, 3]
        Ai = qbar @ np.array([-math.sin(th), math.cos(th)])
        Bi = v * (qbar @ np.array([math.cos(th), math.sin(th)]))
        Wzz += dt * (Ai * (np.outer(dV[i], dTh[i]) + np.outer(dTh[i], dV[i])) - Bi * np.outer(dTh[i], dTh[i]))
    W[:n, :n] = Wzz
    for k in range(N):
        c = a1 * a2 * float(lc[k] @ hk[k])
        W[n + 2 * k, n + 2 * k] = 2.0 * pen[0]
        W[n + 2 * k + 1, n + 2 * k + 1] = 2.0 * pen[1]
        W[n + 2 * k, n + 2 * k + 1] -= c
        W[n + 2 * k + 1, n + 2 * k] -= c
        # d2 row / d rho_i d z = a_i (dh1 dP_{k+1} - dh0 dP_k) + a1 a2 rho_other dh0 dP_k
        d1 = np.einsum("j,ja->a", lc[k], dh[k + 1]) @ dP[k + 1]
        d0 = np.einsum("j,ja->a", lc[k], dh[k]) @ dP[k]
        c1 = -(a1 * (d1 - d0) + a1 * a2 * rho[k, 1] * d0)
        c2 = -(a2 * (d1 - d0) + a1 * a2 * rho[k, 0] * d0)
        W[:n, n + 2 * k] = c1; W[n + 2 * k, :n] = c1
        W[:n, n + 2 * k + 1] = c2; W[n + 2 * k + 1, :n] = c2
    out.update(W=W)
    return out


def block_eig(D):
    """Eigen form of a symmetric 2x2 block, shifted to positive definite: (vs, vw, ls, lw) with
    D + sh I = ls vs vs' + lw vw vw', sh = max(0, eps - lambda_min), eps = 1e-8 max(1, |trace|).
    The decay blocks have one huge (sum sig a a') and one small (penalty) eigenvalue; applying D^-1 through
    this form keeps the huge direction accurate, which an explicit inverse does not."""
    a, b, c = D[0, 0], D[0, 1], D[1, 1]
    tr, df = a + c, a - c
    rad = math.sqrt(df * df + 4.0 * b * b)
    ls = 0.5 * (tr + rad)
    lw = (a * c - b * b) / ls
    vx, vy = (df + rad, 2.0 * b) if df >= 0.0 else (2.0 * b, rad - df)
    vn = math.hypot(vx, vy)
    vs = np.array([vx / vn, vy / vn]) if vn > 0 else np.array([1.0, 0.0])
    vw = np.array([-vs[1], vs[0]])
    sh = max(0.0, 1e-8 * max(1.0, abs(a) + abs(c)) - lw)
    return vs, vw, ls + sh, lw + sh


def solve(x0, u_prev, goal, obs, params=None, return_info=False, linear_algebra="schur", evaluate_fn=None):
    """One optimal-decay MPC-CBF solve.  Returns u_0 (2,), rho_0 (2,), status, iterations [, info].
    evaluate_fn: problem functions of another model with the same layout zz = (z | rho_0 .. rho_{N-1}), two decay variables
    per stage (oracle/od_mpc_gn.py: KinematicBicycle2D, Quad2D); its input box comes in P["u_lo"], P["u_hi"]."""
    P = dict(DEFAULTS)
    if params:
        P.update(params)
    N = P["N"]
    n = int(P.get("nu", 2)) * N                                           # inputs of all stages (VTOL2D: four per stage)
    nu_in = n // N
    sreset = int(P.get("slack_reset", 0))                                 # line search: 2 = s_i <- g_i where g_i >= mu / nu (oracle/mpc_cbf.py: solve)
    x0 = np.asarray(x0, dtype=np.float64)
    obs = np.asarray(obs, dtype=np.float64)
    if evaluate_fn is not None:
        evaluate = lambda x0_, zz_, goal_, obs_, P_, lam_=None, level=2: evaluate_fn(x0_, zz_, u_prev, goal_, obs_, P_, lam_, level)
        lo_, hi_ = np.tile(np.asarray(P["u_lo"], dtype=np.float64), N), np.tile(np.asarray(P["u_hi"], dtype=np.float64), N)
        z = np.clip(np.tile(np.asarray(u_prev, dtype=np.float64), N), lo_ + 0.005 * (hi_ - lo_), hi_ - 0.005 * (hi_ - lo_))
    else:
        evaluate = globals()["evaluate"]
        ub = np.tile([P["a_max"], P["w_max"]], N)
        z = np.clip(np.tile(np.asarray(u_prev, dtype=np.float64), N), -0.99 * ub, 0.99 * ub)   # set_initial_guess
    zz = np.concatenate([z, np.tile([P["omega1"], P["omega2"]], N)])
    ev = evaluate(x0, zz, goal, obs, P, None, level=1)
    circles_only = "model" in P and P["model"].get("circles_only", False)
    if np.any(obs[:, 6] >= 0.5) and not circles_only:
        obs = M.barrier_scales(ev["pts"], obs, P)                          # steep (superellipsoid) barriers: IPOPT-style scaling
        if np.any(obs[:, 7] < 1.0):
            ev = evaluate(x0, zz, goal, obs, P, None, level=1)
    sf = min(1.0, 100.0 / max(1e-12, float(np.max(np.abs(ev["grad"][:n])))))
    g = ev["g"]
    mu = P["mu_init"]
    s = np.maximum(g, 1e-2)
    lam = mu / s
    status, it = STATUS_INACCURATE, 0
    tau, nu, delta_last = 0.995, 10.0, 0.0
    n_acc = 0
    err = np.inf
    e_best, zz_best = np.inf, zz.copy()
    trace = []
    for it in range(1, P["max_iter"] + 1):
        ev = evaluate(x0, zz, goal, obs, P, lam / sf, level=2)
        f, grad, W, g, J = sf * ev["f"], sf * ev["grad"], sf * ev["W"], ev["g"], ev["J"]
        r_d = grad - J.T @ lam
        r_p = g - s
        e_opt = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam)))
        e_mu = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam - mu)))
        err = e_opt
        trace.append((float(np.max(np.abs(r_d))), float(np.max(np.abs(r_p))), float(np.max(np.abs(s * lam))), mu))
        if e_opt < e_best:
            e_best, zz_best = e_opt, zz.copy()
        if e_opt <= P["tol"]:
            status = STATUS_OPTIMAL
            break
        n_acc = n_acc + 1 if e_opt <= P["acceptable_tol"] else 0          # IPOPT's acceptable_iter rule
        if n_acc >= P["acceptable_iter"]:
            break
        if np.max(lam) > 1e10:
            status = STATUS_INFEASIBLE
            break
        while e_mu <= 10.0 * mu and mu > P["mu_min"]:
            mu = max(P["mu_min"], min(0.2 * mu, mu ** 1.5))
            e_mu = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam - mu)))
        sig = lam / s
        Mb = W + J.T @ (sig[:, None] * J)
        rhs = -r_d + J.T @ (mu / s - sig * r_p - lam)
        Muu, Mur, Mrr = Mb[:n, :n], Mb[:n, n:], Mb[n:, n:].copy()
        eig = []
        for k in range(N):                                                  # decay blocks: eigen form, shifted to PD
            vs, vw, ls, lw = block_eig(Mrr[2 * k:2 * k + 2, 2 * k:2 * k + 2])
            eig.append((vs, vw, ls, lw))
            Mrr[2 * k:2 * k + 2, 2 * k:2 * k + 2] = ls * np.outer(vs, vs) + lw * np.outer(vw, vw)

        def apply_dinv(y):                                                  # D^-1 y, block by block in eigen form
            out = np.zeros_like(y)
            for k, (vs, vw, ls, lw) in enumerate(eig):
                yk = y[2 * k:2 * k + 2]
                out[2 * k:2 * k + 2] = vs * ((vs @ yk) / ls) + vw * ((vw @ yk) / lw)
            return out
        delta, dzz = 0.0, None
        for _try in range(40):
            try:
                if linear_algebra == "dense":
                    full = np.block([[Muu + delta * np.eye(n), Mur], [Mur.T, Mrr]])
                    Lf = np.linalg.cholesky(full)
                    dzz = np.linalg.solve(Lf.T, np.linalg.solve(Lf, rhs))
                else:
                    S = Muu + delta * np.eye(n)
                    for k, (vs, vw, ls, lw) in enumerate(eig):
                        cs, cw = Mur[:, 2 * k:2 * k + 2] @ vs, Mur[:, 2 * k:2 * k + 2] @ vw
                        S = S - np.outer(cs, cs) / ls - np.outer(cw, cw) / lw
                    L = np.linalg.cholesky(S)
                    du = np.linalg.solve(L.T, np.linalg.solve(L, rhs[:n] - Mur @ apply_dinv(rhs[n:])))
                    dr = apply_dinv(rhs[n:] - Mur.T @ du)
                    dzz = np.concatenate([du, dr])
                break
            except np.linalg.LinAlgError:
                delta = max(1e-4, delta_last / 3.0) if delta == 0.0 else delta * 8.0
        if dzz is None:
            break
        if delta > 0:
            delta_last = delta
        ds = J @ dzz + r_p
        dlam = -sig * ds - (lam - mu / s)
        neg = ds < 0
        ap = min(1.0, float(np.min(-tau * s[neg] / ds[neg]))) if np.any(neg) else 1.0
        neg = dlam < 0
        ad = min(1.0, float(np.min(-tau * lam[neg] / dlam[neg]))) if np.any(neg) else 1.0
        nu = max(nu, 1.1 * float(np.max(np.abs(lam))))
        srp, dbar = float(np.sum(np.abs(r_p))), float(grad @ dzz - mu * np.sum(ds / s))
        if dbar - nu * srp >= 0.0 and srp > 0.0:
            nu = dbar / (0.9 * srp)                       # no descent direction of the merit: raise the penalty (oracle/mpc_cbf.py: solve)
        phi0 = f - mu * np.sum(np.log(s)) + nu * srp
        dphi = dbar - nu * srp
        alpha, accepted = ap, False
        noise_rows = P.get("row_noise", 0.0) * nu * float(np.sum(np.abs(g)))   # round-off of far dummy-obstacle rows (oracle/mpc_cbf.py)
        for _ in range(12):
            zt, st = zz + alpha * dzz, s + alpha * ds
            e0 = evaluate(x0, zt, goal, obs, P, level=0)
            if sreset == 2:
                st = np.where(e0["g"] >= mu / nu, e0["g"], st)
            phit = sf * e0["f"] - mu * np.sum(np.log(st)) + nu * np.sum(np.abs(e0["g"] - st))
            if phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * abs(phi0) + noise_rows:
                accepted = True
                break
            alpha *= 0.5
        if not accepted:
            break
        zz, s = zz + alpha * dzz, (st if sreset == 2 else s + alpha * ds)
        lam = lam + ad * dlam
        lam = np.minimum(np.maximum(lam, mu / (1e10 * s)), 1e10 * mu / s)
    if status != STATUS_OPTIMAL and e_best <= P["acceptable_tol"]:
        zz, status, err = zz_best, STATUS_OPTIMAL, e_best
    ev = evaluate(x0, zz, goal, obs, P, level=0)
    if status != STATUS_OPTIMAL:
        if np.min(ev["g"]) < -1e-6:
            status = STATUS_INFEASIBLE
        elif status != STATUS_INFEASIBLE:
            status = STATUS_INACCURATE
    u0, rho0 = zz[0:nu_in].copy(), zz[n:n + 2].copy()
    if return_info:
        return u0, rho0, status, it, dict(zz=zz, X=ev["X"], f=ev["f"], g=ev["g"], lam=lam / sf, s=s, err=err, mu=mu, scale=sf, obs=obs, trace=trace)
    return u0, rho0, status, it
