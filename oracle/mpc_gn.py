"""Float64 statement of the MPC-CBF NLP for the reference's remaining planar models with a rel-deg-2 distance barrier --
KinematicBicycle2D, DoubleIntegrator2D, Quad2D (position_control/mpc_cbf.py over robots/kinematic_bicycle2D.py,
robots/double_integrator2D.py, robots/quad2D.py) -- as problem functions for oracle.mpc_cbf.solve(evaluate_fn=...).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **Parity unpinned** like oracle/mpc_cbf.py (do-mpc / casadi / IPOPT
absent); the model functions F / S below are pinned on the reference's own f / g / step (tests/golden: callbacks.npz,
integrators.npz, quad2d.npz).

  prediction  x+ = F(x, u) = x + (f(x) + g(x) u) dt   (Euler, no wrap, no clipping)               mpc_cbf.py:135-141
  cost        sum_{k=1..N} (x_k - xg)' Q (x_k - xg), xg = [goal, 0..]; r-term R on delta u        mpc_cbf.py:144,176-180,267
              KB Q = diag(50,50,1,1), R = (.5, 5000); DI Q = diag(50,50,20,20), R = (.5,.5);
              Quad2D Q = diag(25,25,50,10,10,50), R = (.5,.5)                                       mpc_cbf.py:28-36
  CBF         dd_h + (a1 + a2) d_h + a1 a2 h >= 0 per stage and obstacle with x1 = S(x_k, u_k), x2 = S(x1, u_k), S = the
              robot's own step()                                                                    mpc_cbf.py:316-321
              KB: a = .1, beta 1.1, S = Euler + speed clipped to [v_min, v_max]      kinematic_bicycle2D.py:113-123,175-199
              DI: a = .2, beta 1.01, S = Euler + speed rescaled to v_max, circle or superellipsoid
                                                                                     double_integrator2D.py:79-107,222-272
              Quad2D: a = .15, beta 1.01, S = Euler                                  quad2D.py:81-84,179-206
              h is a function of the planar position only.
  bounds      KB |v_k| <= v_max (k = 1..N), |a| <= a_max, |beta| <= beta_max; DI |a| <= (ax_max, ay_max);
              Quad2D f_min <= u <= f_max                                                            mpc_cbf.py:193-216

Because S applies u_k twice (and clips), the barrier points b_k = pos(S(x_k,u_k)), c_k = pos(S(S(x_k,u_k),u_k)) are NOT the
predicted positions; they are functions of (x_k, u_k).  Hessian: the Gauss-Newton part (exact second derivatives of the cost
in the states and of h in the barrier points) plus, per model flag `exact`, the second derivatives of the dynamics and of
step o step weighted by the costates of the Lagrangian (Quad2D: on; DoubleIntegrator2D: off -- its dynamics are linear and
the curvature of the speed rescaling changes nothing measurable).  The HIP kernel csrc/mpc_gn.hip follows this iterate for
iterate.
Rows: [CBF (stage major, obstacle minor) | state bounds hi - x, x - lo (stage major) | u_hi - z | z - u_lo].
"""
import math

import numpy as np

from . import mpc_cbf as M


# ---- models: F (prediction), S (robot.step), their Jacobians ------------------------------------------------------
def _kb_fg(x, spec):
    th, v = x[2], x[3]
    f = np.array([v * math.cos(th), v * math.sin(th), 0.0, 0.0])
    g = np.array([[0.0, -v * math.sin(th)], [0.0, v * math.cos(th)], [0.0, v / spec["rear_ax_dist"]], [1.0, 0.0]])
    return f, g


def kb_F(x, u, spec, dt, jac=False):
    f, g = _kb_fg(x, spec)
    xn = x + (f + g @ u) * dt
    if not jac:
        return xn
    th, v, b = x[2], x[3], u[1]
    s, c = math.sin(th), math.cos(th)
    Lr = spec["rear_ax_dist"]
    A = np.eye(4)
    A[0, 2] += dt * (-v * s - v * c * b); A[0, 3] += dt * (c - s * b)
    A[1, 2] += dt * (v * c - v * s * b);  A[1, 3] += dt * (s + c * b)
    A[2, 3] += dt * b / Lr
    B = dt * np.array([[0.0, -v * s], [0.0, v * c], [0.0, v / Lr], [1.0, 0.0]])
    return xn, A, B


def kb_S(x, u, spec, dt, jac=False):
    """robot.step: Euler, heading wrap (irrelevant for h), speed clipped to [v_min, v_max] (kinematic_bicycle2D.py:113-123)."""
    r = kb_F(x, u, spec, dt, jac)
    xn = (r[0] if jac else r).copy()
    clipped = not (spec["v_min"] <= xn[3] <= spec["v_max"])
    xn[3] = min(max(xn[3], spec["v_min"]), spec["v_max"])
    if not jac:
        return xn
    A, B = r[1].copy(), r[2].copy()
    if clipped:
        A[3, :] = 0.0; B[3, :] = 0.0
    return xn, A, B


def di_F(x, u, spec, dt, jac=False):
    xn = np.array([x[0] + dt * x[2], x[1] + dt * x[3], x[2] + dt * u[0], x[3] + dt * u[1]])
    if not jac:
        return xn
    A = np.eye(4); A[0, 2] = dt; A[1, 3] = dt
    B = np.zeros((4, 2)); B[2, 0] = dt; B[3, 1] = dt
    return xn, A, B


def di_S(x, u, spec, dt, jac=False):
    """robot.step: Euler, then the velocity rescaled to norm v_max when above it (double_integrator2D.py:79-107)."""
    r = di_F(x, u, spec, dt, jac)
    xn = (r[0] if jac else r).copy()
    w = xn[2:4].copy()
    vm = math.hypot(w[0], w[1])
    over = vm > spec["v_max"]
    if over:
        xn[2:4] = w * (spec["v_max"] / vm)
    if not jac:
        return xn
    A, B = r[1].copy(), r[2].copy()
    if over:
        Jc = spec["v_max"] * (np.eye(2) / vm - np.outer(w, w) / vm ** 3)
        A[2:4, :] = Jc @ A[2:4, :]; B[2:4, :] = Jc @ B[2:4, :]
    return xn, A, B


def q2_F(x, u, spec, dt, jac=False):
    m, I, r = spec["mass"], spec["inertia"], spec["radius"]
    th = x[2]
    s, c = math.sin(th), math.cos(th)
    T = u[0] + u[1]
    xn = np.array([x[0] + dt * x[3], x[1] + dt * x[4], x[2] + dt * x[5], x[3] + dt * (-s / m) * T,
                   x[4] + dt * (-9.81 + (c / m) * T), x[5] + dt * (r / I) * (u[0] - u[1])])
    if not jac:
        return xn
    A = np.eye(6); A[0, 3] = dt; A[1, 4] = dt; A[2, 5] = dt
    A[3, 2] = dt * (-c / m) * T; A[4, 2] = dt * (-s / m) * T
    B = np.zeros((6, 2))
    B[3, :] = dt * (-s / m); B[4, :] = dt * (c / m); B[5, 0] = dt * r / I; B[5, 1] = -dt * r / I
    return xn, A, B


# ---- contracted second derivatives  sum_i c_i grad^2 F_i(x, u)  over the variables (x, u), (nx + 2) x (nx + 2) -----------
def kb_H(x, u, spec, dt, c, step=False):
    """KinematicBicycle2D, F and S alike: the clipped component (the speed) is linear in (x, u) either way."""
    th, v, b = x[2], x[3], u[1]
    s, co = math.sin(th), math.cos(th)
    H = np.zeros((6, 6))
    H[2, 2] = c[0] * dt * (-v * co + v * s * b) + c[1] * dt * (-v * s - v * co * b)
    H[2, 3] = c[0] * dt * (-s - co * b) + c[1] * dt * (co - s * b)
    H[2, 5] = c[0] * (-dt * v * co) + c[1] * (-dt * v * s)
    H[3, 5] = c[0] * (-dt * s) + c[1] * (dt * co) + c[2] * dt / spec["rear_ax_dist"]
    return H + np.triu(H, 1).T


def di_H(x, u, spec, dt, c, step=False):
    """DoubleIntegrator2D: F is linear; S rescales w = v + dt u to norm v_max when above it."""
    H = np.zeros((6, 6))
    if not step:
        return H
    w = np.array([x[2] + dt * u[0], x[3] + dt * u[1]])
    vm = math.hypot(w[0], w[1])
    if vm <= spec["v_max"]:
        return H
    Hw = np.zeros((2, 2))
    for d in range(2):
        for a in range(2):
            for b in range(2):
                Hw[a, b] += c[2 + d] * spec["v_max"] * (-((d == a) * w[b] + (d == b) * w[a] + (a == b) * w[d]) / vm ** 3
                                                      + 3.0 * w[d] * w[a] * w[b] / vm ** 5)
    Jw = np.zeros((2, 6)); Jw[0, 2] = 1.0; Jw[1, 3] = 1.0; Jw[0, 4] = dt; Jw[1, 5] = dt
    return Jw.T @ Hw @ Jw


def q2_H(x, u, spec, dt, c, step=False):
    m = spec["mass"]
    s, co = math.sin(x[2]), math.cos(x[2])
    T = u[0] + u[1]
    H = np.zeros((8, 8))
    H[2, 2] = c[3] * (dt * s * T / m) + c[4] * (-dt * co * T / m)
    H[2, 6] = H[2, 7] = c[3] * (-dt * co / m) + c[4] * (-dt * s / m)
    return H + np.triu(H, 1).T


def kb_model(spec=None, dt=0.05):
    """EXPERIMENTAL, not served by the HIP kernel: with the Gauss-Newton matrix this interior point converges on well
    under half of the test draws for the bicycle (fast heading dynamics, large tracking residuals: the dropped second
    derivatives of the dynamics dominate).  Kept for the exact-Hessian version planned in DESIGN.md (f)."""
    s = dict(wheel_base=0.4, radius=0.3, rear_ax_dist=0.2, v_max=3.5, a_max=5.0, v_min=0.2)
    s["beta_max"] = math.atan((0.2 / 0.4) * math.tan(math.radians(32)))
    s.update(spec or {})
    return dict(name="KinematicBicycle2D", nx=4, nu=2, F=kb_F, S=kb_S, H=kb_H, spec=s, dt=dt, Q=np.array([50.0, 50.0, 1.0, 1.0]),
                R=np.array([0.5, 5000.0]), alpha1=0.1, alpha2=0.1, beta=1.1, radius=s["radius"],
                u_lo=np.array([-s["a_max"], -s["beta_max"]]), u_hi=np.array([s["a_max"], s["beta_max"]]),
                xb=[(3, -s["v_max"], s["v_max"])], circles_only=True,
                slack_reset=2)    # the bicycles' line search resets the slacks (oracle/mpc_cbf.py: solve): fewer crawlers, more optima


def di_model(spec=None, dt=0.05):
    s = dict(a_max=1.0, v_max=1.0, radius=0.25)
    s.update(spec or {})
    s.setdefault("ax_max", s["a_max"]); s.setdefault("ay_max", s["a_max"])
    return dict(name="DoubleIntegrator2D", nx=4, nu=2, F=di_F, S=di_S, H=di_H, spec=s, dt=dt, Q=np.array([50.0, 50.0, 20.0, 20.0]),
                R=np.array([0.5, 0.5]), alpha1=0.2, alpha2=0.2, beta=1.01, radius=s["radius"],
                u_lo=np.array([-s["ax_max"], -s["ay_max"]]), u_hi=np.array([s["ax_max"], s["ay_max"]]), xb=[], circles_only=False,
                exact=False)      # linear dynamics: Gauss-Newton is exact except for the curvature of the speed rescaling, which is left out


def quad2d_model(spec=None, dt=0.05):
    s = dict(mass=1.0, inertia=0.01, f_min=1.0, f_max=10.0, radius=0.25)
    s.update(spec or {})
    return dict(name="Quad2D", nx=6, nu=2, F=q2_F, S=q2_F, H=q2_H, spec=s, dt=dt, Q=np.array([25.0, 25.0, 50.0, 10.0, 10.0, 50.0]),
                R=np.array([0.5, 0.5]), alpha1=0.15, alpha2=0.15, beta=1.01, radius=s["radius"],
                u_lo=np.full(2, s["f_min"]), u_hi=np.full(2, s["f_max"]), xb=[], circles_only=True)


def params(model, N=10, **over):
    P = dict(M.DEFAULTS, N=N, dt=model["dt"], nu=model.get("nu", 2), u_lo=model["u_lo"], u_hi=model["u_hi"], radius=model["radius"],
             alpha1=model["alpha1"], alpha2=model["alpha2"], beta=model["beta"], model=model)
    if "slack_reset" in model:
        P["slack_reset"] = model["slack_reset"]
    P.update(over)
    P.setdefault("row_noise", 1e-15)                       # Armijo allowance for the round-off of far dummy-obstacle rows (as mpc_lin)
    return P


def barrier(p, obs, P):
    if P["model"]["circles_only"]:
        d = P["radius"] + obs[2]
        e = p - obs[0:2]
        return e @ e - P["beta"] * d * d, 2.0 * e, 2.0 * np.eye(2)
    return M.barrier(p, obs, P)


def evaluate(x0, z, u_prev, goal, obs, P, lam=None, level=2):
    mdl = P["model"]
    N, nx, nu, dt, spec = P["N"], mdl["nx"], mdl.get("nu", 2), mdl["dt"], mdl["spec"]
    n = N * nu
    K = obs.shape[0]
    Q, Rw, xb = mdl["Q"], mdl["R"], mdl["xb"]
    w0, w1, w2 = M.cbf_weights(P)
    # optimal decay (oracle/od_mpc_gn.py): the row weights depend on the stage, the input term is R u^2
    SW = np.asarray(P["stage_w"], dtype=np.float64) if P.get("stage_w") is not None else np.tile([w0, w1, w2], (N, 1))
    r_on_u = P.get("rterm") == "u"
    xg = np.zeros(nx); xg[:2] = np.asarray(goal, dtype=np.float64)[:2]
    U = z.reshape(N, nu)
    der = level >= 1
    X = np.zeros((N + 1, nx)); X[0] = np.asarray(x0, dtype=np.float64)[:nx]
    Phi = np.zeros((N + 1, nx, n))
    pts = np.zeros((N, 3, 2)); G = np.zeros((N, 3, 2, n))
    jac = [None] * N
    for k in range(N):
        E = np.zeros((nu, n)); E[:, k * nu:(k + 1) * nu] = np.eye(nu)
        if der:
            xn, A, B = mdl["F"](X[k], U[k], spec, dt, True)
            Phi[k + 1] = A @ Phi[k] + B @ E
            y1, S1x, S1u = mdl["S"](X[k], U[k], spec, dt, True)
            y2, S2x, S2u = mdl["S"](y1, U[k], spec, dt, True)
            Y1 = S1x @ Phi[k] + S1u @ E
            Y2 = S2x @ Y1 + S2u @ E
            G[k, 0] = Phi[k][0:2]; G[k, 1] = Y1[0:2]; G[k, 2] = Y2[0:2]
            jac[k] = (A, S1x, S1u, S2x, y1)
        else:
            xn = mdl["F"](X[k], U[k], spec, dt)
            y1 = mdl["S"](X[k], U[k], spec, dt)
            y2 = mdl["S"](y1, U[k], spec, dt)
        X[k + 1] = xn
        pts[k, 0], pts[k, 1], pts[k, 2] = X[k][0:2], y1[0:2], y2[0:2]
    f = 0.0
    for k in range(1, N + 1):
        e = X[k] - xg
        f += float(Q @ (e * e))
    up = np.concatenate([np.asarray(u_prev, dtype=np.float64)[:nu], z])
    du = z.copy() if r_on_u else up[nu:] - up[:-nu]
    Rd = np.tile(Rw, N)
    f += float(np.sum(Rd * du * du))
    hv = np.zeros((N, 3, K)); dh = np.zeros((N, 3, K, 2)); Hh = np.zeros((N, 3, K, 2, 2))
    for k in range(N):
        for p in range(3):
            for j in range(K):
                hv[k, p, j], dh[k, p, j], Hh[k, p, j] = barrier(pts[k, p], obs[j], P)
    nb = len(xb)
    one_sided = any(not (np.isfinite(lo) and np.isfinite(hi)) for (_, lo, hi) in xb)    # VTOL2D: descent-speed floor only
    nbr = sum(int(np.isfinite(lo)) + int(np.isfinite(hi)) for (_, lo, hi) in xb) if one_sided else 2 * nb
    m = N * K + nbr * N + 2 * n
    g = np.zeros(m)
    g[: N * K] = (SW[:, 0, None] * hv[:, 0] + SW[:, 1, None] * hv[:, 1] + SW[:, 2, None] * hv[:, 2]).reshape(-1)
    o = N * K
    for k in range(1, N + 1):
        for (idx, lo, hi) in xb:
            if not one_sided:
                g[o] = hi - X[k, idx]; g[o + 1] = X[k, idx] - lo
                o += 2
                continue
            if np.isfinite(hi):
                g[o] = hi - X[k, idx]; o += 1
            if np.isfinite(lo):
                g[o] = X[k, idx] - lo; o += 1
    hi_, lo_ = np.tile(mdl["u_hi"], N), np.tile(mdl["u_lo"], N)
    g[o:o + n] = hi_ - z
    g[o + n:] = z - lo_
    out = dict(f=float(f), g=g, X=X, pts=pts.reshape(-1, 2), hv=hv)
    if level == 0:
        return out
    grad = np.zeros(n)
    for k in range(1, N + 1):
        grad += Phi[k].T @ (2.0 * Q * (X[k] - xg))
    Dm = np.eye(n) if r_on_u else np.eye(n) - np.eye(n, k=-nu)
    grad += 2.0 * Dm.T @ (Rd * du)
    J = np.zeros((m, n))
    JP = np.einsum("kpjd,kpdn->kpjn", dh, G)                               # gradient of h_j(point p of stage k) in z
    for k in range(N):
        for j in range(K):
            J[k * K + j] = sum(SW[k, p] * JP[k, p, j] for p in range(3))
    o = N * K
    for k in range(1, N + 1):
        for (idx, lo, hi) in xb:
            if not one_sided:
                J[o] = -Phi[k][idx]; J[o + 1] = Phi[k][idx]
                o += 2
                continue
            if np.isfinite(hi):
                J[o] = -Phi[k][idx]; o += 1
            if np.isfinite(lo):
                J[o] = Phi[k][idx]; o += 1
    J[o:o + n] = -np.eye(n)
    J[o + n:] = np.eye(n)
    out.update(grad=grad, J=J, JP=JP)
    if level == 1:
        return out
    lam = np.zeros(m) if lam is None else lam
    lc = lam[: N * K].reshape(N, K)
    W = 2.0 * Dm.T @ (Rd[:, None] * Dm)
    for k in range(1, N + 1):
        W += 2.0 * Phi[k].T @ (Q[:, None] * Phi[k])
    for k in range(N):
        for p in range(3):
            Om = -SW[k, p] * np.einsum("j,jab->ab", lc[k], Hh[k, p])
            W += G[k, p].T @ Om @ G[k, p]
    if P.get("exact_hessian", mdl.get("exact", True)):
        # second derivatives of the dynamics and of step o step, weighted by the costates of the Lagrangian
        #   p_N = mu_N,  p_k = mu_k + (d points_k / d x_k)' nu + A_k' p_{k+1};   mu_k = 2 Q (x_k - xg) + bound multipliers,
        #   nu_p = -w_p sum_j lam_kj dh_j(point_p);   W += V_k' H_k V_k,  V_k = [Phi_k; E_k],
        #   H_k = H_F(x_k,u_k; p_{k+1}) + H_S(x_k,u_k; P'nu_1) + D' H_S(y1,u_k; P'nu_2) D + H_S(x_k,u_k; S2x' P'nu_2)
        ls = lam[N * K:N * K + nbr * N].reshape(N, nbr) if nb else None     # per stage, rows in the order g lists them
        pk = np.zeros(nx)
        for k in range(N, -1, -1):
            mu_k = np.zeros(nx)
            if k >= 1:
                mu_k = 2.0 * Q * (X[k] - xg)
                o = 0
                for (idx, lo, hi) in xb:
                    if not one_sided or np.isfinite(hi):
                        mu_k[idx] += ls[k - 1, o]; o += 1                  # row hi - x >= 0: -lam * d(row)/dx = +lam
                    if not one_sided or np.isfinite(lo):
                        mu_k[idx] -= ls[k - 1, o]; o += 1
            if k == N:
                pk = mu_k
                continue
            A, S1x, S1u, S2x, y1 = jac[k]
            nu_ = [-SW[k, p] * (lc[k] @ dh[k, p]) for p in range(3)]
            c1 = np.zeros(nx); c1[0:2] = nu_[1]
            c2 = np.zeros(nx); c2[0:2] = nu_[2]
            D = np.zeros((nx + nu, nx + nu)); D[:nx, :nx] = S1x; D[:nx, nx:] = S1u; D[nx:, nx:] = np.eye(nu)
            Hk = mdl["H"](X[k], U[k], spec, dt, pk) + mdl["H"](X[k], U[k], spec, dt, c1, True) \
                + D.T @ mdl["H"](y1, U[k], spec, dt, c2, True) @ D + mdl["H"](X[k], U[k], spec, dt, S2x[0:2].T @ nu_[2], True)
            E = np.zeros((nu, n)); E[:, k * nu:(k + 1) * nu] = np.eye(nu)
            V = np.vstack([Phi[k], E])
            W += V.T @ Hk @ V
            g_pts = np.zeros(nx); g_pts[0:2] = nu_[0]
            g_pts += S1x[0:2].T @ nu_[1] + (S2x[0:2] @ S1x).T @ nu_[2]
            pk = mu_k + g_pts + A.T @ pk
    out.update(W=W)
    return out


def solve(model, x0, u_prev, goal, obs, N=10, params_over=None, return_info=False):
    P = params(model, N, **(params_over or {}))
    return M.solve(x0, u_prev, goal, obs, params=P, return_info=return_info, evaluate_fn=evaluate)
