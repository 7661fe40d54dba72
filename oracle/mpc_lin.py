"""Float64 statement of the MPC-CBF NLP for the reference's LINEAR robot models -- SingleIntegrator2D and Quad3D
(position_control/mpc_cbf.py over robots/single_integrator2D.py, robots/quad3D.py) -- as problem functions for
oracle.mpc_cbf.solve(evaluate_fn=...).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **Parity unpinned** like oracle/mpc_cbf.py (do-mpc / casadi /
IPOPT absent); the model matrices are pinned on the reference's own f / g / step (tests/golden/linear_models.npz).

  prediction  x+ = x + (f(x) + g(x) u) dt = Ae x + Be u  (Euler, every model)                mpc_cbf.py:135-141
              SI: f = 0, g = I                                                             single_integrator2D.py:45-62
              Quad3D: f = A x, g = B = B1 B2 (linearised 12-state quadrotor)                  quad3D.py:77-119
  cost        sum_{k=1..N} (x_k - xg)' Q (x_k - xg), xg = [goal, 0..]; r-term R on delta u   mpc_cbf.py:144,176-180,267
              SI Q = diag(50,50), R = (5,5); Quad3D Q = diag(30,30,5,20,20,1,10,10,10,20,20,1), R = 1   mpc_cbf.py:19-21,37-39
  CBF         d_h + alpha h >= 0 per stage and obstacle, d_h = h(step(x_k, u_k)) - h(x_k)        mpc_cbf.py:312-315
              SI: step = Euler, alpha = 0.05                      single_integrator2D.py:64-66,148-195 ; mpc_cbf.py:48-50
              Quad3D: step = RK4 of the linear system, alpha = 0.15, circles only (cylinders)  quad3D.py:121-158,275-296 ; mpc_cbf.py:77-81
              h is a function of the planar position only, so the angle wrap inside Quad3D.step cannot change it.
  bounds      SI |u| <= v_max; Quad3D u_min <= u <= u_max                                      mpc_cbf.py:183-187,219-223

Both models are linear, so with z = (u_0..u_{N-1}) every barrier point is affine in z:  a_k = pos(x_k),
b_k = pos(As x_k + Bs u_k)  (As, Bs: the barrier's own one-step map), row(k, j) = h_j(b_k) - (1 - alpha) h_j(a_k).
Rows: [CBF (stage major, obstacle minor) | u_hi - z | z - u_lo].
"""
import numpy as np

from . import mpc_cbf as M

GRAVITY_Q3D = 9.8                                           # quad3D.py:70


def si_model(spec=None, dt=0.05):
    spec = {**dict(v_max=1.0, radius=0.25), **(spec or {})}
    A = np.zeros((2, 2)); B = np.eye(2)
    Ae, Be = np.eye(2) + dt * A, dt * B
    return dict(name="SingleIntegrator2D", nx=2, nu=2, ng=2, Ae=Ae, Be=Be, As=Ae.copy(), Bs=Be.copy(),
                Q=np.array([50.0, 50.0]), R=np.array([5.0, 5.0]), alpha=0.05,
                u_lo=np.full(2, -spec["v_max"]), u_hi=np.full(2, spec["v_max"]), radius=spec["radius"], dt=dt, circles_only=False)


def quad3d_matrices(spec):
    """A (12x12), B = B1 B2 (12x4), quad3D.py:70-96."""
    m, Ix, Iy, Iz, L, nu_ = spec["mass"], spec["Ix"], spec["Iy"], spec["Iz"], spec["L"], spec["nu"]
    B2 = np.array([[1, 1, 1, 1], [0, L, 0, -L], [L, 0, -L, 0], [nu_, -nu_, nu_, -nu_]], dtype=np.float64)
    A = np.zeros((12, 12))
    for i in range(6):
        A[i, 6 + i] = 1.0
    A[6, 3] = GRAVITY_Q3D
    A[7, 4] = -GRAVITY_Q3D
    B1 = np.zeros((12, 4))
    B1[8, 0] = 1.0 / m; B1[9, 1] = 1.0 / Iy; B1[10, 2] = 1.0 / Ix; B1[11, 3] = 1.0 / Iz
    return A, B1 @ B2


def quad3d_model(spec=None, dt=0.05):
    s = dict(mass=3.0, Ix=0.5, Iy=0.5, Iz=0.5, L=0.3, nu=0.1, u_max=10.0, u_min=-10.0, radius=0.25)   # quad3D.py:50-59
    s.update(spec or {})
    A, B = quad3d_matrices(s)
    I = np.eye(12)
    Ae, Be = I + dt * A, dt * B
    # RK4 of x' = A x + B u with u held (quad3D.py:140-146): k1 = A x + B u, k2 = A (x + dt/2 k1) + B u, ...
    K1x, K1u = A, B
    K2x, K2u = A @ (I + dt / 2 * K1x), A @ (dt / 2 * K1u) + B
    K3x, K3u = A @ (I + dt / 2 * K2x), A @ (dt / 2 * K2u) + B
    K4x, K4u = A @ (I + dt * K3x), A @ (dt * K3u) + B
    As = I + dt / 6 * (K1x + 2 * K2x + 2 * K3x + K4x)
    Bs = dt / 6 * (K1u + 2 * K2u + 2 * K3u + K4u)
    return dict(name="Quad3D", nx=12, nu=4, ng=3, Ae=Ae, Be=Be, As=As, Bs=Bs,
                Q=np.array([30, 30, 5, 20, 20, 1, 10, 10, 10, 20, 20, 1], dtype=np.float64), R=np.ones(4), alpha=0.15,
                u_lo=np.full(4, s["u_min"]), u_hi=np.full(4, s["u_max"]), radius=s["radius"], dt=dt, circles_only=True)


def condensed(model, N, rterm="du"):
    """Constant matrices of the condensed problem: Hc = Hessian of the cost in z (n x n), G = d points / d z
    (4N x n, points ordered a_0..a_{N-1}, b_0..b_{N-1}, two rows each)."""
    nx, nu = model["nx"], model["nu"]
    n = N * nu
    Ae, Be, As, Bs = model["Ae"], model["Be"], model["As"], model["Bs"]
    Phi = np.zeros((N + 1, nx, n))                        # d x_k / d z
    for k in range(N):
        Phi[k + 1] = Ae @ Phi[k]
        Phi[k + 1][:, k * nu:(k + 1) * nu] += Be
    Hc = np.zeros((n, n))
    for k in range(1, N + 1):
        Hc += 2.0 * Phi[k].T @ (model["Q"][:, None] * Phi[k])
    Dm = np.eye(n) if rterm == "u2" else np.eye(n) - np.eye(n, k=-nu)      # optimal decay: R u^2 instead of the delta-u penalty
    Rd = np.tile(model["R"], N)
    Hc += 2.0 * Dm.T @ (Rd[:, None] * Dm)
    G = np.zeros((4 * N, n))
    for k in range(N):
        G[2 * k:2 * k + 2] = Phi[k][0:2]
        gb = As[0:2] @ Phi[k]
        gb[:, k * nu:(k + 1) * nu] += Bs[0:2]
        G[2 * N + 2 * k:2 * N + 2 * k + 2] = gb
    return Hc, G, Phi


def params(model, N=10, **over):
    P = dict(M.DEFAULTS, N=N, dt=model["dt"], nu=model["nu"], u_lo=model["u_lo"], u_hi=model["u_hi"], radius=model["radius"],
             alpha=model["alpha"], model=model,
             slack_reset=2)     # round 4: the line search resets slacks (oracle/mpc_cbf.py: solve) -- Quad3D at N = 20 crawled for up to 351
                                # iterations at step lengths of 1e-2 without it (71 with it), SingleIntegrator2D 34 -> 20; N = 10 Quad3D unchanged
    P.update(over)
    P["quadratic_cost"] = condensed(model, P["N"], P.get("rterm", "du"))[0]   # exact merit differences in the line search
    P.setdefault("row_noise", 1e-15)
    return P


def barrier(p, obs, P):
    if P["model"]["circles_only"]:                        # quad3D.py:283-291: no superellipsoid branch
        d = P["radius"] + obs[2]
        e = p - obs[0:2]
        return e @ e - P["beta"] * d * d, 2.0 * e, 2.0 * np.eye(2)
    return M.barrier(p, obs, P)


def evaluate(x0, z, u_prev, goal, obs, P, lam=None, level=2):
    mdl = P["model"]
    N, nx, nu = P["N"], mdl["nx"], mdl["nu"]
    n = N * nu
    K = obs.shape[0]
    Q, Rw = mdl["Q"], mdl["R"]
    # per-stage gain: the optimal-decay extension (oracle/od_mpc_rd1.py) scales alpha by the stage's decay variable
    w0 = -(1.0 - np.broadcast_to(np.asarray(P.get("alpha_k", P["alpha"]), dtype=np.float64), (N,)))
    u2 = P.get("rterm", "du") == "u2"
    xg = np.zeros(nx); xg[: mdl["ng"]] = np.asarray(goal, dtype=np.float64)[: mdl["ng"]]   # mpc_cbf.py:267
    X = np.zeros((N + 1, nx)); X[0] = np.asarray(x0, dtype=np.float64)[:nx]
    U = z.reshape(N, nu)
    for k in range(N):
        X[k + 1] = mdl["Ae"] @ X[k] + mdl["Be"] @ U[k]
    a = X[:N, 0:2]
    b = np.array([(mdl["As"] @ X[k] + mdl["Bs"] @ U[k])[0:2] for k in range(N)])
    f = 0.0
    for k in range(1, N + 1):
        e = X[k] - xg
        f += float(Q @ (e * e))
    up = np.concatenate([np.asarray(u_prev, dtype=np.float64)[:nu], z])
    du = z.copy() if u2 else up[nu:] - up[:-nu]
    Rd = np.tile(Rw, N)
    f += float(np.sum(Rd * du * du))
    ha = np.zeros((N, K)); hb = np.zeros((N, K)); da = np.zeros((N, K, 2)); db = np.zeros((N, K, 2))
    Ha = np.zeros((N, K, 2, 2)); Hb = np.zeros((N, K, 2, 2))
    for k in range(N):
        for j in range(K):
            ha[k, j], da[k, j], Ha[k, j] = barrier(a[k], obs[j], P)
            hb[k, j], db[k, j], Hb[k, j] = barrier(b[k], obs[j], P)
    m = N * K + 2 * n
    g = np.zeros(m)
    g[: N * K] = (hb + w0[:, None] * ha).reshape(-1)
    hi, lo = np.tile(mdl["u_hi"], N), np.tile(mdl["u_lo"], N)
    g[N * K:N * K + n] = hi - z
    g[N * K + n:] = z - lo
    out = dict(f=float(f), g=g, X=X, pts=np.vstack([a, b]))
    if level == 0:
        return out
    Hc, G, Phi = condensed(mdl, N, "u2" if u2 else "du")
    grad = np.zeros(n)
    for k in range(1, N + 1):
        grad += Phi[k].T @ (2.0 * Q * (X[k] - xg))
    Dm = np.eye(n) if u2 else np.eye(n) - np.eye(n, k=-nu)
    grad += 2.0 * Dm.T @ (Rd * du)
    J = np.zeros((m, n))
    Ga, Gb = G[: 2 * N].reshape(N, 2, n), G[2 * N:].reshape(N, 2, n)
    for k in range(N):
        for j in range(K):
            J[k * K + j] = db[k, j] @ Gb[k] + w0[k] * (da[k, j] @ Ga[k])
    J[N * K:N * K + n] = -np.eye(n)
    J[N * K + n:] = np.eye(n)
    out.update(grad=grad, J=J)
    if P.get("want_internals"):                                              # h(a_kj) and its Jacobian in z
        out.update(ha=ha.copy(), Ja=np.einsum("kja,kan->kjn", da, Ga))
    if level == 1:
        return out
    lam = np.zeros(m) if lam is None else lam
    lc = lam[: N * K].reshape(N, K)
    W = Hc.copy()
    for k in range(N):
        Om_b = -np.einsum("j,jab->ab", lc[k], Hb[k])
        Om_a = -w0[k] * np.einsum("j,jab->ab", lc[k], Ha[k])
        W += Gb[k].T @ Om_b @ Gb[k] + Ga[k].T @ Om_a @ Ga[k]
    out.update(W=W)
    return out


def solve(model, x0, u_prev, goal, obs, N=10, params_over=None, return_info=False):
    P = params(model, N, **(params_over or {}))
    return M.solve(x0, u_prev, goal, obs, params=P, return_info=return_info, evaluate_fn=evaluate)
