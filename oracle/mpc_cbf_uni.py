"""Float64 statement of the MPC-CBF NLP for the kinematic Unicycle2D model (position_control/mpc_cbf.py with
robots/unicycle2D.py) -- the problem functions; the solver is oracle.mpc_cbf.solve(evaluate_fn=evaluate).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **Parity unpinned** like oracle/mpc_cbf.py (do-mpc / casadi /
IPOPT absent).  Restated from the reference:

  model       x = [x, y, theta], u = [v, omega],  x+ = x + g(x) u dt  (f = 0)          mpc_cbf.py:135-141, unicycle2D.py:42-67
  cost        sum_k (x_k - goal)' Q (x_k - goal), Q = diag(50, 50, .01); r-term R = (.5, .5) on delta u    mpc_cbf.py:22-24,176-180
  CBF         d_h + alpha h >= 0 per stage, alpha = 0.05, h = |p - o|^2 - beta d_min^2 (circle only, beta = 1.01),
              d_h = h(step(x_k, u_k)) - h(x_k):  row_k = h(p_{k+1}) - (1 - alpha) h(p_k)   mpc_cbf.py:52-53,312-315; unicycle2D.py:127-145
  bounds      |v| <= v_max, |omega| <= w_max (inputs only)                              mpc_cbf.py:188-192

Rows: [CBF (stage major, obstacle minor) | u_max - z | u_max + z]  (no state bound rows).
P["a_max"] carries v_max (the bound of the first input), so oracle.mpc_cbf.solve's box / initial guess code applies.
"""
import math

import numpy as np

from . import mpc_cbf as M

DEFAULTS = dict(M.DEFAULTS, Q=(50.0, 50.0, 0.01), R=(0.5, 0.5), alpha=0.05, a_max=1.0, w_max=0.5)


def rollout(x0, z, P):
    N, dt = P["N"], P["dt"]
    X = np.zeros((N + 1, 3))
    X[0] = x0[:3]
    for k in range(N):
        x, y, th = X[k]
        v, w = z[2 * k], z[2 * k + 1]
        X[k + 1] = [x + dt * v * math.cos(th), y + dt * v * math.sin(th), th + dt * w]
    return X


def evaluate(x0, z, u_prev, goal, obs, P, lam=None, level=2):
    N, dt = P["N"], P["dt"]
    n = 2 * N
    Q, Rw = np.asarray(P["Q"], dtype=np.float64), np.asarray(P["R"], dtype=np.float64)
    K = obs.shape[0]
    # per-stage gain: the optimal-decay extension (oracle/od_mpc_rd1.py) scales alpha by the stage's decay variable
    al = np.broadcast_to(np.asarray(P.get("alpha_k", P["alpha"]), dtype=np.float64), (N,))
    w0, w1 = -(1.0 - al), 1.0                                                # w0[k]: weight of h(p_k) in row k
    u2 = P.get("rterm", "du") == "u2"                                        # optimal decay: R u^2, not do-mpc's delta-u penalty
    X = rollout(x0, z, P)
    pos = X[:, 0:2]
    gpos = np.asarray(goal, dtype=np.float64)[0:2]
    out = {}
    f = 0.0
    for k in range(1, N + 1):
        e = pos[k] - gpos
        f += Q[0] * e[0] ** 2 + Q[1] * e[1] ** 2 + Q[2] * X[k, 2] ** 2
    up = np.concatenate([np.asarray(u_prev, dtype=np.float64), z])
    du = z.copy() if u2 else up[2:] - up[:-2]
    Rd = np.tile(Rw, N)
    f += float(np.sum(Rd * du * du))
    hk = np.zeros((N + 1, K)); dh = np.zeros((N + 1, K, 2)); Hh = np.zeros((N + 1, K, 2, 2))
    for k in range(N + 1):
        for j in range(K):
            hk[k, j], dh[k, j], Hh[k, j] = M.barrier(pos[k], obs[j], P)
    m = N * K + 2 * n
    g = np.zeros(m)
    for k in range(N):
        g[k * K:(k + 1) * K] = w1 * hk[k + 1] + w0[k] * hk[k]
    o = N * K
    ub = np.tile([P["a_max"], P["w_max"]], N)
    g[o:o + n] = ub - z
    g[o + n:o + 2 * n] = ub + z
    out.update(f=float(f), g=g, X=X, pts=pos)
    if level == 0:
        return out
    # first derivatives: theta_k = th0 + dt sum_{i<k} w_i ; p_k = p0 + dt sum_{l<k} v_l (cos, sin)(theta_l)
    dTh = np.zeros((N + 1, n))
    for k in range(N + 1):
        for j in range(k):
            dTh[k, 2 * j + 1] = dt
    C = np.stack([np.cos(X[:, 2]), np.sin(X[:, 2])], axis=1)
    D = np.stack([-np.sin(X[:, 2]), np.cos(X[:, 2])], axis=1)
    dP = np.zeros((N + 1, 2, n))
    for k in range(N + 1):
        for l in range(k):
            dP[k, :, 2 * l] += dt * C[l]
            dP[k] += dt * z[2 * l] * np.outer(D[l], dTh[l])
    grad = np.zeros(n)
    for k in range(1, N + 1):
        grad += dP[k].T @ (2.0 * Q[0:2] * (pos[k] - gpos)) + 2.0 * Q[2] * X[k, 2] * dTh[k]
    Dm = np.eye(n) if u2 else np.eye(n) - np.eye(n, k=-2)
    grad += 2.0 * Dm.T @ (Rd * du)
    J = np.zeros((m, n))
    for k in range(N):
        for j in range(K):
            J[k * K + j] = w1 * dh[k + 1, j] @ dP[k + 1] + w0[k] * dh[k, j] @ dP[k]
    o = N * K
    J[o:o + n] = -np.eye(n)
    J[o + n:o + 2 * n] = np.eye(n)
    out.update(grad=grad, J=J)
    if P.get("want_internals"):                                              # h(a_kj) and its Jacobian in z, a_k = p_k
        out.update(ha=hk[:N].copy(), Ja=np.einsum("kja,kan->kjn", dh[:N], dP[:N]))
    if level == 1:
        return out
    lam = np.zeros(m) if lam is None else lam
    lc = lam[: N * K].reshape(N, K)
    mu = np.zeros((N + 1, K))
    for k in range(N + 1):
        if 1 <= k: mu[k] += w1 * lc[k - 1]
        if k <= N - 1: mu[k] += w0[k] * lc[k]
    W = 2.0 * Dm.T @ (Rd[:, None] * Dm)
    q = np.zeros((N + 1, 2))
    for k in range(N + 1):
        Om = -np.einsum("j,jab->ab", mu[k], Hh[k])
        qk = -mu[k] @ dh[k]
        if 1 <= k <= N:
            Om = Om + np.diag(2.0 * Q[0:2])
            qk = qk + 2.0 * Q[0:2] * (pos[k] - gpos)
            W += 2.0 * Q[2] * np.outer(dTh[k], dTh[k])
        q[k] = qk
        W += dP[k].T @ Om @ dP[k]
    # second derivatives of the positions: sum_k q_k . d2 p_k = sum_l qbar_l . d2 [dt v_l (cos, sin)(theta_l)],  qbar_l = sum_{k>l} q_k
    #   = dt sum_l [ A_l (e_vl dTh_l' + dTh_l e_vl') - B_l dTh_l dTh_l' ],  A_l = qbar_l . (-sin, cos)_l,  B_l = v_l qbar_l . (cos, sin)_l
    for l in range(N):
        qbar = q[l + 1:].sum(axis=0)
        Al = qbar @ D[l]
        Bl = z[2 * l] * (qbar @ C[l])
        ev = np.zeros(n); ev[2 * l] = 1.0
        W += dt * (Al * (np.outer(ev, dTh[l]) + np.outer(dTh[l], ev)) - Bl * np.outer(dTh[l], dTh[l]))
    out.update(W=W)
    return out


def solve(x0, u_prev, goal, obs, params=None, return_info=False):
    P = dict(DEFAULTS)
    if params:
        P.update(params)
    return M.solve(x0, u_prev, goal, obs, params=P, return_info=return_info, evaluate_fn=evaluate)
