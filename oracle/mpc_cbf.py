"""Float64 restatement of the MPC-CBF NLP (position_control/mpc_cbf.py) and a reference solver.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **Parity unpinned**: the
reference solves this NLP with do-mpc -> casadi -> IPOPT (mpc_cbf.py:163,384),
none of which can be installed here, the NLP is non-convex, and no reference
test pins its result.  What is restated from the reference is the *problem*:

  model       x+ = x + (f(x) + g(x) u) dt, no angle wrap           mpc_cbf.py:135-141
  cost        sum_{k<N} (x_k-goal)'Q(x_k-goal) + terminal same      mpc_cbf.py:144,176-178,267
              + sum_k sum_i R_i (u_k,i - u_{k-1,i})^2  (do-mpc rterm) mpc_cbf.py:180
  weights     DU: Q = diag(50,50,.01,30), R = (.5,.5), N = 10       mpc_cbf.py:15,25-27
  CBF         dd_h + (a1+a2) d_h + a1 a2 h >= 0 at stages 0..N-1     mpc_cbf.py:304,316-321
              with x1 = step(x_k,u_k), x2 = step(x1,u_k)             robots/dynamic_unicycle2D.py:188-238
              DT gains a1 = a2 = 0.15                                mpc_cbf.py:56-59
  obstacles   padded to num_obs rows with [1000,1000,0,...]          mpc_cbf.py:338-364
  bounds      |v_k| <= v_max (all k), |a| <= a_max, |w| <= w_max     mpc_cbf.py:193-199
  protocol    constant initial guess x0 / u_prev every call          mpc_cbf.py:366-384 (set_initial_guess)

The barrier h depends on the position only, so the heading wrap inside
``step`` (fmod) cannot change it; with x1 = x_{k+1} and x2's position equal to
the next predicted position, every CBF row is a function of three consecutive
predicted positions p_k, p_{k+1}, p_{k+2}.

Solver (the algorithm the HIP kernel implements, here in numpy float64): the
NLP is condensed by single shooting onto z = (u_0..u_{N-1}) and solved by a
primal-dual interior-point method with slacks on every inequality, a
Gauss-Newton Hessian (always positive definite thanks to the input-rate
penalty), fraction-to-the-boundary steps and an l1-merit backtracking line
search, started from the same constant guess the reference uses.
"""
import math

import numpy as np

STATUS_OPTIMAL = 0
STATUS_INFEASIBLE = 1
STATUS_INACCURATE = 2

DEFAULTS = dict(N=10, dt=0.05, Q=(50.0, 50.0, 0.01, 30.0), R=(0.5, 0.5), alpha1=0.15, alpha2=0.15,
                v_max=1.0, a_max=1.0, w_max=0.5, radius=0.25, beta=1.01,
                tol=1e-6, acceptable_tol=1e-5, acceptable_iter=15, max_iter=3000, mu_init=0.1, mu_min=1e-9,
                resto_rho=1000.0, resto_kappa=0.1, resto_theta_tol=1e-6, resto_max=2, resto_tol=1e-2,
                resto_small_alpha=0.02, resto_small_iter=4, resto_slack_reset=True,
                resto_retry=3, resto_stall_iter=40, resto_stall_theta=1e-3)

DUMMY_OBS = np.array([1000.0, 1000.0, 0.0, 0.0, 0.0, 0.0, 0.0])


def pad_obstacles(obs, num_obs):
    """update_tvp, mpc_cbf.py:338-364: 3-wide rows get zero tails, missing rows are far-away dummies."""
    out = np.tile(DUMMY_OBS, (num_obs, 1))
    if obs is None or len(obs) == 0:
        return out
    rows = []
    for ob in obs:
        ob = np.asarray(ob, dtype=np.float64)
        if ob.shape[0] == 3:
            ob = np.concatenate([ob, np.zeros(4)])
        elif ob.shape[0] != 7:
            raise ValueError(f"Invalid obstacle format: {ob}")
        rows.append(ob)
    rows = np.array(rows)[:num_obs]
    out[: len(rows)] = rows
    return out


def rollout(x0, z, P):
    """Predicted states x_0..x_N plus one extra position p_{N+1} (second step of the last DT-CBF row)."""
    N, dt = P["N"], P["dt"]
    X = np.zeros((N + 1, 4))
    X[0] = x0
    for k in range(N):
        x, y, th, v = X[k]
        a, w = z[2 * k], z[2 * k + 1]
        X[k + 1] = [x + dt * v * math.cos(th), y + dt * v * math.sin(th), th + dt * w, v + dt * a]
    xe = X[N]
    p_extra = np.array([xe[0] + dt * xe[3] * math.cos(xe[2]), xe[1] + dt * xe[3] * math.sin(xe[2])])
    return X, p_extra


def position_jacobians(X, P):
    """dP[k] = d p_k / d z for k = 0..N+1, shape (N+2, 2, 2N).

    d p_k / d a_j = dt^2 sum_{i=j+1}^{k-1} (cos th_i, sin th_i),
    d p_k / d w_j = dt^2 sum_{i=j+1}^{k-1} v_i (-sin th_i, cos th_i).
    """
    N, dt = P["N"], P["dt"]
    n = 2 * N
    C = np.stack([np.cos(X[:, 2]), np.sin(X[:, 2])], axis=1)              # (N+1,2)
    D = np.stack([-X[:, 3] * np.sin(X[:, 2]), X[:, 3] * np.cos(X[:, 2])], axis=1)
    PC = np.vstack([np.zeros((1, 2)), np.cumsum(C, axis=0)])                # PC[k] = sum_{i<k} C_i, k = 0..N+1
    PD = np.vstack([np.zeros((1, 2)), np.cumsum(D, axis=0)])
    dP = np.zeros((N + 2, 2, n))
    for k in range(N + 2):
        for j in range(N):
            if j + 1 <= k - 1:
                dP[k, :, 2 * j] = dt * dt * (PC[k] - PC[j + 1])
                dP[k, :, 2 * j + 1] = dt * dt * (PD[k] - PD[j + 1])
    return dP


def barrier(p, obs, P):
    """h, dh/dp (2,), d2h/dp2 (2,2) at position p.

    Circle robots/dynamic_unicycle2D.py:194-202, superellipsoid :204-220 (fabs, clamps a,b >= 1e-3, e >= 2).
    """
    R, beta = P["radius"], P["beta"]
    if obs[6] < 0.5:
        d = R + obs[2]
        e = p - obs[0:2]
        return e @ e - beta * d * d, 2.0 * e, 2.0 * np.eye(2)
    a = max(abs(obs[2]), 1e-3) + R
    b = max(abs(obs[3]), 1e-3) + R
    ex = max(abs(obs[4]), 2.0)
    ct, st = math.cos(obs[5]), math.sin(obs[5])
    dx, dy = p[0] - obs[0], p[1] - obs[1]
    px, py = ct * dx + st * dy, -st * dx + ct * dy
    ax, ay = abs(px) / a, abs(py) / b
    h = ax ** ex + ay ** ex - 1.0
    gpx = ex * ax ** (ex - 1) / a * np.sign(px)
    gpy = ex * ay ** (ex - 1) / b * np.sign(py)
    hxx = ex * (ex - 1) * ax ** (ex - 2) / (a * a)
    hyy = ex * (ex - 1) * ay ** (ex - 2) / (b * b)
    Rm = np.array([[ct, st], [-st, ct]])                                    # p' = Rm (p - o)
    sc = obs[7] if len(obs) > 7 else 1.0                                    # row scaling of a steep barrier: barrier_scales
    return sc * h, sc * (Rm.T @ np.array([gpx, gpy])), sc * (Rm.T @ np.diag([hxx, hyy]) @ Rm)


SCALE_MAX_GRADIENT = 100.0


def barrier_scales(pts, obs, P, barrier_fn=None):
    """Gradient-based scaling of the steep barriers, IPOPT's default NLP scaling (nlp_scaling_method = gradient-based,
    nlp_scaling_max_gradient = 100: a constraint whose gradient exceeds 100 at the starting point is scaled down to 100)
    applied per superellipsoid obstacle: h_j <- sc_j h_j with
        sc_j = min(1, 100 / max_pts |grad h_j(pt)|_inf)   over the barrier points of the initial guess.
    A superellipsoid of exponent 6 seen from 4 m away has h ~ 5e5 and |grad h| ~ 7e5 (a circle: h ~ 16, |grad h| ~ 8);
    unscaled, the linearisation error of such far-away, irrelevant rows dominates the merit function and the interior
    point stalls (fewer than one in five of BASELINE config 5's scenes converged; all do with the scaling).  Circles are
    left alone: their gradient 2 |p - o| stays below the threshold in any scene the reference draws.  Returns obs with an
    eighth column (1 for circles) that ``barrier`` multiplies into h, grad h and the Hessian."""
    barrier_fn = barrier_fn or barrier
    obs = np.asarray(obs, dtype=np.float64)[:, :7]
    sc = np.ones(obs.shape[0])
    for j, o in enumerate(obs):
        if o[6] >= 0.5:
            gm = 0.0
            for pt in np.asarray(pts, dtype=np.float64).reshape(-1, 2):
                gm = max(gm, float(np.max(np.abs(barrier_fn(pt, o, P)[1]))))
            sc[j] = max(min(1.0, SCALE_MAX_GRADIENT / max(gm, 1e-300)), 1e-30)
    return np.hstack([obs, sc[:, None]])


def cbf_weights(P):
    """c = w2 h(p_{k+2}) + w1 h(p_{k+1}) + w0 h(p_k)  ==  dd_h + (a1+a2) d_h + a1 a2 h  (mpc_cbf.py:316-321)."""
    g1 = P["alpha1"] + P["alpha2"]
    g2 = P["alpha1"] * P["alpha2"]
    return 1.0 - g1 + g2, g1 - 2.0, 1.0


def evaluate(x0, z, u_prev, goal, obs, P, lam=None, level=2):
    """Problem functions at z.

    level 0: f, g.   level 1: + grad f, J.   level 2: + W = Hessian of f - lam' g  (exact).
    Inequalities g >= 0 are ordered [CBF (k major, obstacle minor) | v_max - v_k, v_max + v_k (k=1..N) |
    u_max - z | u_max + z].
    """
    N, dt = P["N"], P["dt"]
    n = 2 * N
    Q, Rw = np.asarray(P["Q"], dtype=np.float64), np.asarray(P["R"], dtype=np.float64)
    K = obs.shape[0]
    w0, w1, w2 = cbf_weights(P)
    X, p_extra = rollout(x0, z, P)
    pos = np.vstack([X[:, 0:2], p_extra[None, :]])                          # p_0..p_{N+1}
    gpos = np.asarray(goal, dtype=np.float64)[0:2]
    out = {}
    # ---- values -------------------------------------------------------------------------
    f = 0.0
    for k in range(1, N + 1):
        e = pos[k] - gpos
        f += Q[0] * e[0] ** 2 + Q[1] * e[1] ** 2 + Q[2] * X[k, 2] ** 2 + Q[3] * X[k, 3] ** 2
    up = np.concatenate([np.asarray(u_prev, dtype=np.float64), z])
    du = up[2:] - up[:-2]
    Rd = np.tile(Rw, N)
    f += float(np.sum(Rd * du * du))
    hk = np.zeros((N + 2, K)); dh = np.zeros((N + 2, K, 2)); Hh = np.zeros((N + 2, K, 2, 2))
    for k in range(N + 2):
        for j in range(K):
            hk[k, j], dh[k, j], Hh[k, j] = barrier(pos[k], obs[j], P)
    m = N * K + 2 * N + 2 * n
    g = np.zeros(m)
    for k in range(N):
        g[k * K:(k + 1) * K] = w2 * hk[k + 2] + w1 * hk[k + 1] + w0 * hk[k]
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
    # ---- first derivatives ------------------------------------------------------------------
    dP = position_jacobians(X, P)                                           # (N+2, 2, n)
    dTh = np.zeros((N + 1, n)); dV = np.zeros((N + 1, n))                   # d theta_k / dz, d v_k / dz
    for k in range(N + 1):
        for j in range(k):
            dTh[k, 2 * j + 1] = dt
            dV[k, 2 * j] = dt
    grad = np.zeros(n)
    for k in range(1, N + 1):
        grad += dP[k].T @ (2.0 * Q[0:2] * (pos[k] - gpos)) + 2.0 * Q[2] * X[k, 2] * dTh[k] + 2.0 * Q[3] * X[k, 3] * dV[k]
    Dm = np.eye(n) - np.eye(n, k=-2)
    grad += 2.0 * Dm.T @ (Rd * du)
    J = np.zeros((m, n))
    for k in range(N):
        for j in range(K):
            J[k * K + j] = w2 * dh[k + 2, j] @ dP[k + 2] + w1 * dh[k + 1, j] @ dP[k + 1] + w0 * dh[k, j] @ dP[k]
    o = N * K
    for k in range(1, N + 1):
        J[o + 2 * (k - 1)] = -dV[k]
        J[o + 2 * (k - 1) + 1] = dV[k]
    o += 2 * N
    J[o:o + n] = -np.eye(n)
    J[o + n:o + 2 * n] = np.eye(n)
    out.update(grad=grad, J=J)
    if level == 1:
        return out
    # ---- exact Hessian of the Lagrangian ------------------------------------------------------
    # L = sum_k phi_k(p_k) + (terms quadratic in z),  phi_k(p) = Qp-weighted |p - goal|^2 - sum_j mu_kj h_j(p)
    # mu_kj = w2 lam_{k-2,j} + w1 lam_{k-1,j} + w0 lam_{k,j}   (CBF multipliers touching position k)
    lam = np.zeros(m) if lam is None else lam
    lc = lam[: N * K].reshape(N, K)
    mu = np.zeros((N + 2, K))
    for k in range(N + 2):
        if k - 2 >= 0: mu[k] += w2 * lc[k - 2]
        if 1 <= k <= N: mu[k] += w1 * lc[k - 1]
        if k <= N - 1: mu[k] += w0 * lc[k]
    W = 2.0 * Dm.T @ (Rd[:, None] * Dm)
    q = np.zeros((N + 2, 2))                                                # d L / d p_k
    for k in range(N + 2):
        Om = -np.einsum("j,jab->ab", mu[k], Hh[k])
        qk = -mu[k] @ dh[k]
        if 1 <= k <= N:
            Om = Om + np.diag(2.0 * Q[0:2])
            qk = qk + 2.0 * Q[0:2] * (pos[k] - gpos)
            W += 2.0 * Q[2] * np.outer(dTh[k], dTh[k]) + 2.0 * Q[3] * np.outer(dV[k], dV[k])
        q[k] = qk
        W += dP[k].T @ Om @ dP[k]
    # second derivatives of the positions: p_k = p_0 + dt sum_{i<k} v_i (cos th_i, sin th_i), th_i and v_i linear in z
    #   sum_k q_k . d2 p_k = dt sum_i [ A_i (dV_i dTh_i' + dTh_i dV_i') - B_i dTh_i dTh_i' ],
    #   qbar_i = sum_{k>i} q_k,  A_i = qbar_i . (-sin, cos)_i,  B_i = v_i qbar_i . (cos, sin)_i
    for i in range(N + 1):
        qbar = q[i + 1:].sum(axis=0)
        th, v = X[i, 2], X[i, 3]
        Ai = qbar @ np.array([-math.sin(th), math.cos(th)])
        Bi = v * (qbar @ np.array([math.cos(th), math.sin(th)]))
        W += dt * (Ai * (np.outer(dV[i], dTh[i]) + np.outer(dTh[i], dV[i])) - Bi * np.outer(dTh[i], dTh[i]))
    out.update(W=W)
    return out


def problem_functions(x0, z, u_prev, goal, obs, P, want_jac=True):
    """Back-compat helper used by the scipy cross-checks: (f, g, X) or (f, grad, W0, g, J, X)."""
    if not want_jac:
        r = evaluate(x0, z, u_prev, goal, obs, P, level=0)
        return r["f"], r["g"], r["X"]
    r = evaluate(x0, z, u_prev, goal, obs, P, level=2)
    return r["f"], r["grad"], r["W"], r["g"], r["J"], r["X"]


def _resto_central_path(g, mu, rho):
    """Slack of an elastic row on the central path of the restoration problem:  mu/s + mu/t = rho  with  t = s - g  > 0."""
    return ((2.0 * mu + rho * g) + np.sqrt(rho * rho * g * g + 4.0 * mu * mu)) / (2.0 * rho)


def violation(g, m_el):
    """theta(z): l1 norm of the violated part of the ELASTIC rows (everything but the input box, which stays hard)."""
    return float(np.sum(np.maximum(0.0, -g[:m_el])))


def solve(x0, u_prev, goal, obs, params=None, return_info=False, evaluate_fn=None):
    """One MPC-CBF solve.  Returns u_0 (2,), status, iterations [, info dict].

    Regular phase: primal-dual interior point on  min f(z) s.t. g(z) - s = 0, s >= 0  with the exact Hessian of the
    Lagrangian, inertia correction (W + delta I until the condensed matrix is positive definite),
    fraction-to-the-boundary 0.995, l1 merit backtracking, monotone barrier decrease
    mu <- max(mu_min, min(0.2 mu, mu^1.5)) once the barrier problem is solved to 10 mu.
    The objective is scaled by min(1, 100 / |grad f(z0)|_inf) like IPOPT's gradient-based scaling.

    Feasibility restoration (Waechter & Biegler 2006, section 3.3, on the condensed problem): when the regular phase
    cannot continue at an infeasible iterate z_R -- its line search fails after 12 halvings, or the multipliers pass
    1e10 -- the solver switches to
        min_z  rho_R * sum_i t_i + zeta/2 |z - z_R|^2    s.t.  g_i(z) + t_i >= 0, t_i >= 0  (elastic rows i: the N K
               CBF rows),   state bounds and the input box stay hard (both linear in z),   rho_R = 1000, zeta = sqrt(mu),
    solved by the SAME primal-dual iteration.  An elastic row  g_i + t_i - s_i = 0  keeps its slack s_i and multiplier
    lam_i and gains ONE number, t_i; the multiplier of t_i >= 0 is rho_R - lam_i (stationarity in t_i, kept exactly by
    a common dual step) and dt_i is eliminated from the Newton system, so the restoration differs from the regular phase
    only per row: Sigma_i = Sigma_s Sigma_t / (Sigma_s + Sigma_t) with Sigma_t = (rho_R - lam) / t, another right-hand-
    side entry, two more fraction-to-the-boundary ratios (t, rho_R - lam) and the merit terms rho_R t - mu log t.  It
    starts on the central path of the elastic rows (s, t from g and mu_R = max(mu, |violation|_inf)) with the box rows
    re-centred as at the start of the solve, and returns to the regular phase -- a fresh start at the current z with the
    barrier parameter it left with -- as soon as the violation theta(z) = sum max(0, -g_i) has dropped to resto_kappa *
    theta(z_R) (first entry; a later entry means the regular phase came back to the same stall, and runs until no violation
    is left).  If instead it CONVERGES (its own KKT error <= tol, or the acceptable rule) with theta > resto_theta_tol,
    z is a stationary point of the violation: status INFEASIBLE is that certificate and u_0 of that minimiser is what
    is returned (what IPOPT reports as "converged to a point of local infeasibility").  At most resto_max entries.
    Both phases raise the merit penalty when a step is not a descent direction of the merit function (the penalty was
    below the multipliers lam + dlam of the step).
    Slack reset (Byrd, Hribar & Nocedal 1999, section 3): for fixed z the merit function is separable in s, and
    -mu log s_i + nu |g_i - s_i| is smallest at s_i = max(g_i, mu / nu); after a trial step the line search may therefore
    replace the linearly updated slack of a row by the row's value.  P["slack_reset"] (regular phase; 0 off, 1 only raises
    slacks, 2 takes the minimiser where g_i >= mu / nu: the bicycles and VTOL2D) and P["resto_slack_reset"] (the same on
    g_i + t_i inside the restoration; on by default, off for VTOL2D) switch it on.  A row whose curvature beat its
    linearisation -- a far obstacle's row, never active -- is then not charged for it, which is what kept those models'
    searches at step lengths of 1e-3 and their restorations crawling to the iteration limit.
    Stalled restorations (round 4).  A restoration whose line search fails is retried from the same z with a Levenberg-damped
    step, delta >= 1, 1e2, 1e4 (P["resto_retry"] = 3 retries, each one iteration; the damping decays through the inertia
    correction's memory delta_last / 3 like a shrinking trust region): the plateaus where the Newton direction of the minimal
    inertia correction makes no progress are left this way, towards a proper stationary point of the violation or back to the
    regular phase.  A restoration that has not lowered theta by 1 % within P["resto_stall_iter"] = 40 iterations while theta >
    P["resto_stall_theta"] = 1e-3 is stopped with STATUS_INFEASIBLE (with less violation than that: stopped as well, STATUS_INACCURATE --
    the crawl gets nowhere either way, and a violation of 1e-5 is not evidence of infeasibility): it sits at a local minimiser of the violation at a kink of the
    rows (C3BF's sqrt(max(|p|^2 - r^2, 0))), where no KKT error goes to zero and the steps crawl at lengths of 1e-3 for the rest of
    the budget (tests/test_oracle_mpc_resto.py: an independent phase-1 finds no feasible plan for such problems).
    Gauss-Newton restoration (P["resto_gn"], VTOL2D; round 4).  The Hessian of the restoration's Lagrangian is zeta I - sum_i lam_i
    grad^2 g_i with multipliers of the violated rows near rho_R = 1000: for the tilt-rotor (exact Hessian through the aero model, R =
    50000 on the elevator) it is so indefinite that the inertia correction adds delta ~ 1e6 and the steps shrink to 1e-3 -- restorations
    that lower a violation of 4 - 90 by 10 % per 40 iterations and then fail a line search.  With the flag the restoration's Newton
    system is J' Sigma J + zeta I (the second-order terms of the rows dropped: a convex model of a problem whose objective is the l1
    violation): the eight bench draws traced (one of them 517 iterations to a failed line search before) end OPTIMAL in 61 - 114.
    Every other unsuccessful exit is STATUS_INACCURATE.
    """
    P = dict(DEFAULTS)
    if params:
        P.update(params)
    N = P["N"]
    evaluate = evaluate_fn or globals()["evaluate"]       # other models plug their problem functions in here
    x0 = np.asarray(x0, dtype=np.float64)
    obs = np.asarray(obs, dtype=np.float64)
    if "u_hi" in P:                                       # models with their own input box (oracle/mpc_lin.py)
        lo_, hi_ = np.tile(np.asarray(P["u_lo"], dtype=np.float64), N), np.tile(np.asarray(P["u_hi"], dtype=np.float64), N)
        z = np.clip(np.tile(np.asarray(u_prev, dtype=np.float64), N), lo_ + 0.005 * (hi_ - lo_), hi_ - 0.005 * (hi_ - lo_))
    else:
        ub = np.tile([P["a_max"], P["w_max"]], N)
        z = np.clip(np.tile(np.asarray(u_prev, dtype=np.float64), N), -0.99 * ub, 0.99 * ub)   # set_initial_guess
    nz = z.shape[0]
    ev = evaluate(x0, z, u_prev, goal, obs, P, None, level=1)
    circles_only = "model" in P and P["model"].get("circles_only", False)  # models whose DT barrier has no superellipsoid branch
    if np.any(obs[:, 6] >= 0.5) and not circles_only:
        obs = barrier_scales(ev["pts"], obs, P)                            # steep (superellipsoid) barriers: IPOPT-style scaling
        if np.any(obs[:, 7] < 1.0):
            ev = evaluate(x0, z, u_prev, goal, obs, P, None, level=1)
    sf0 = min(1.0, 100.0 / max(1e-12, float(np.max(np.abs(ev["grad"])))))  # objective scaling
    g = ev["g"]
    m = g.shape[0]
    m_el = N * obs.shape[0]                                                 # elastic rows in the restoration: the CBF rows come first
    mu = P["mu_init"]
    s = np.maximum(g, 1e-2)
    lam = mu / s
    status, it = STATUS_INACCURATE, 0
    tau, nu, delta_last = 0.995, 10.0, 0.0
    n_acc = 0
    err = np.inf
    e_best, z_best = np.inf, z.copy()
    n_eval = 1
    # restoration state
    resto = False
    rho_R, kappa_R, theta_tol = P["resto_rho"], P["resto_kappa"], P["resto_theta_tol"]
    n_resto, it_resto, theta_R, mu_reg, z_R = 0, 0, 0.0, mu, z.copy()
    delta_force, n_retry, theta_ref, n_stall, n_stalled, n_retried = 0.0, 0, 0.0, 0, 0, 0   # stalled restorations: damped retries, stall counter
    n_small = 0                                                             # consecutive regular iterations with a tiny step at an infeasible z
    SF_OFF = 1e-40                                                          # "no objective": evaluate() divides lam by it
    el = np.arange(m) < m_el
    Hq = P.get("quadratic_cost")                                            # linear models: f is exactly quadratic in z
    t_ = np.zeros(m)                                                        # elastic variables of the restoration (0 on hard rows)
    sreset = int(P.get("slack_reset", 0))
    sreset_r = bool(P["resto_slack_reset"])                               # the restoration's line search resets the slack of a row to g + t
    for it in range(1, P["max_iter"] + 1):
        sf = SF_OFF if resto else sf0
        ev = evaluate(x0, z, u_prev, goal, obs, P, lam / sf, level=2)      # multipliers of the unscaled problem
        f, grad, W, g, J = sf * ev["f"], sf * ev["grad"], sf * ev["W"], ev["g"], ev["J"]
        if resto and violation(g, m_el) <= max(kappa_R * theta_R if n_resto == 1 else 0.0, theta_tol):
            # enough of the violation is gone: back to the regular phase from here (slacks kept, multipliers on the
            # central path of the regular barrier problem, merit penalty and best iterate reset)
            resto, mu, sf = False, mu_reg, sf0
            s = np.maximum(g, 1e-2)                                         # a fresh start of the regular phase at this z
            lam = mu / s
            nu, n_acc, e_best, z_best = 10.0, 0, np.inf, z.copy()
            ev = evaluate(x0, z, u_prev, goal, obs, P, lam / sf, level=2)
            f, grad, W, g, J = sf * ev["f"], sf * ev["grad"], sf * ev["W"], ev["g"], ev["J"]
        Wc = W
        if resto and P.get("resto_gn"):
            Wc = 0.0 * W                                                    # Gauss-Newton restoration (VTOL2D): see the docstring
        if resto:
            zeta = math.sqrt(mu)
            f, grad, W = 0.5 * zeta * float((z - z_R) @ (z - z_R)), zeta * (z - z_R), Wc + zeta * np.eye(nz)
            nu_t = rho_R - lam
            r_p = g + t_ - s
            ct, ct_mu = np.where(el, np.abs(t_ * nu_t), 0.0), np.where(el, np.abs(t_ * nu_t - mu), 0.0)
        else:
            r_p = g - s
            ct, ct_mu = 0.0, 0.0
        r_d = grad - J.T @ lam
        e_opt = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam)), np.max(ct))
        e_mu = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam - mu)), np.max(ct_mu))
        err = e_opt
        if not resto and e_opt < e_best:                                    # remember the best iterate
            e_best, z_best = e_opt, z.copy()
        if resto:
            # A stationary point of the violation.  The restoration's KKT error is in units of its objective rho_R * theta, so
            # |grad theta| <= e_opt / rho_R; over the input box (a few units across) theta cannot fall by more than ~10 e_opt /
            # rho_R from here: the certificate asks for more violation than that.
            theta = violation(g, m_el)
            if e_opt <= P["resto_tol"] and theta > max(theta_tol, 10.0 * e_opt / rho_R):
                status = STATUS_INFEASIBLE
                break
            if e_opt <= P["tol"]:                                           # solved, and (nearly) no violation left: nothing to certify
                break
            if P["resto_stall_iter"] > 0:
                if theta <= 0.99 * theta_ref:
                    theta_ref, n_stall = theta, 0
                else:
                    n_stall += 1
                    if n_stall >= P["resto_stall_iter"]:
                        if theta > P["resto_stall_theta"]:
                            status, n_stalled = STATUS_INFEASIBLE, 1       # stalled at a kink of the rows with violation left
                        break                                               # (a stall with next to no violation left stays INACCURATE)
        elif e_opt <= P["tol"]:
            status = STATUS_OPTIMAL
            break
        n_acc = n_acc + 1 if e_opt <= P["acceptable_tol"] else 0          # IPOPT's acceptable_iter rule
        if n_acc >= P["acceptable_iter"]:
            if resto and violation(g, m_el) > theta_tol:
                status = STATUS_INFEASIBLE
            break
        want_resto = not resto and np.max(lam) > 1e10                       # multipliers diverge: locally infeasible
        if not want_resto:
            mu_old = mu
            while e_mu <= 10.0 * mu and mu > P["mu_min"]:
                mu = max(P["mu_min"], min(0.2 * mu, mu ** 1.5))
                if resto:
                    ct_mu = np.where(el, np.abs(t_ * nu_t - mu), 0.0)
                e_mu = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam - mu)), np.max(ct_mu))
            if resto and mu != mu_old:                                      # zeta = sqrt(mu): the proximity term follows the new mu
                zeta = math.sqrt(mu)
                f, grad, W = 0.5 * zeta * float((z - z_R) @ (z - z_R)), zeta * (z - z_R), Wc + zeta * np.eye(nz)
            # ---- row quantities: Sigma_i and the multiplier step at dz = 0 -------------------------------------------------
            sig = lam / s
            dl0 = -sig * r_p - lam + mu / s
            if resto:
                # elastic row:  g + t - s = 0 (lam),  t >= 0 (rho_R - lam);  t is eliminated from the Newton system
                tt_ = np.where(el, t_, 1.0)
                sg_t = nu_t / tt_
                se = sig * sg_t / (sig + sg_t)
                dl0 = np.where(el, -se * (r_p + mu / nu_t - t_) - (se / sig) * (lam - mu / s), dl0)
                sig = np.where(el, se, sig)
            Mb = W + J.T @ (sig[:, None] * J)
            rhs = -grad + J.T @ (lam + dl0)
            delta = delta_force                                             # 0 unless a failed restoration step is being retried
            L = None
            for _try in range(40):                                          # inertia correction
                try:
                    L = np.linalg.cholesky(Mb + delta * np.eye(nz))
                    break
                except np.linalg.LinAlgError:
                    delta = max(1e-4, delta_last / 3.0) if delta == 0.0 else delta * 8.0
            if L is None:
                break
            if delta > 0:
                delta_last = delta
            dz = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            jd = J @ dz
            dlam = -sig * jd + dl0
            if resto:
                dt_ = np.where(el, (mu / nu_t - t_) + dlam / sg_t, 0.0)
                ds = jd + dt_ + r_p
            else:
                ds = jd + r_p
            neg = ds < 0
            ap = min(1.0, float(np.min(-tau * s[neg] / ds[neg]))) if np.any(neg) else 1.0
            neg = dlam < 0
            ad = min(1.0, float(np.min(-tau * lam[neg] / dlam[neg]))) if np.any(neg) else 1.0
            if resto:
                neg = el & (dt_ < 0)
                if np.any(neg):
                    ap = min(ap, float(np.min(-tau * t_[neg] / dt_[neg])))
                neg = el & (dlam > 0)
                if np.any(neg):
                    ad = min(ad, float(np.min(tau * nu_t[neg] / dlam[neg])))
            nu = max(nu, 1.1 * float(np.max(np.abs(lam))))
            srp = float(np.sum(np.abs(r_p)))
            if resto:
                bar0 = f + rho_R * float(np.sum(t_[el])) - mu * (np.sum(np.log(s)) + np.sum(np.log(t_[el])))
                dbar = grad @ dz + rho_R * float(np.sum(dt_[el])) - mu * (np.sum(ds / s) + np.sum(dt_[el] / t_[el]))
            else:
                bar0 = f - mu * np.sum(np.log(s))
                dbar = grad @ dz - mu * np.sum(ds / s)
            if dbar - nu * srp >= 0.0 and srp > 0.0:
                # the step is no descent direction of the merit function: the penalty is below the multipliers of the step
                # (lam + dlam); raise it so that the directional derivative is -0.1 nu |r_p|_1  (Nocedal & Wright (18.36))
                nu = dbar / (0.9 * srp)
            phi0 = bar0 + nu * srp
            dphi = dbar - nu * srp
            alpha, accepted = ap, False
            curv = sf * float(dz @ Hq @ dz) if Hq is not None else 0.0
            # round-off of the constraint part of the merit: a far-away dummy obstacle row has h ~ 2e6, so |g - s| carries an
            # absolute error of ~ulp(2e6) per such row, times nu (only the linear models ask for this allowance)
            noise_rows = P.get("row_noise", 0.0) * nu * float(np.sum(np.abs(g)))
            for _ in range(12):                                             # at most 12 halvings, then give up (best iterate)
                zt, st = z + alpha * dz, s + alpha * ds
                e0 = evaluate(x0, zt, u_prev, goal, obs, P, level=0)
                n_eval += 1
                if sreset and not resto:
                    # slack reset (Byrd, Hribar & Nocedal 1999, section 3): for fixed z the merit function is separable in s and
                    # -mu log s_i + nu |g_i - s_i| is smallest at s_i = max(g_i, mu / nu).  Mode 1 only raises slacks
                    # (s = max(s, g): a row whose curvature beat its linearisation is not charged for it), mode 2 takes the minimiser
                    gt = e0["g"]
                    st = np.maximum(st, gt) if sreset == 1 else np.where(gt >= mu / nu, gt, st)
                if resto:
                    tt = t_ + alpha * dt_
                    if sreset_r:                                            # the same reset on g + t of the restoration's rows
                        tot = e0["g"] + tt
                        st = np.where(tot >= mu / nu, tot, st)
                    phit = 0.5 * zeta * float((zt - z_R) @ (zt - z_R)) + rho_R * float(np.sum(tt[el])) \
                        - mu * (np.sum(np.log(st)) + np.sum(np.log(tt[el]))) + nu * np.sum(np.abs(e0["g"] + tt - st))
                elif Hq is not None:
                    # f(z + a dz) - f(z) = a grad.dz + a^2/2 dz'H dz without the cancellation of two sums of size |f|
                    # (a quadrotor far from its goal has f ~ 1e3 and a decrease of 1e-9 to resolve)
                    phit = phi0 + alpha * float(grad @ dz) + 0.5 * alpha * alpha * curv \
                        - mu * float(np.sum(np.log(st) - np.log(s))) \
                        + nu * float(np.sum(np.abs(e0["g"] - st)) - np.sum(np.abs(r_p)))
                else:
                    phit = sf * e0["f"] - mu * np.sum(np.log(st)) + nu * np.sum(np.abs(e0["g"] - st))
                # Armijo, with an allowance for round-off in the merit function near convergence
                # (f is a sum of a few hundred terms of size |phi|: its noise is ~1e-13 |phi|)
                if phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * abs(phi0) + noise_rows:
                    accepted = True
                    break
                alpha *= 0.5
            if P.get("trace") is not None:
                P["trace"].append(dict(it=it, resto=resto, e_opt=e_opt, r_d=float(np.max(np.abs(r_d))), r_p=float(np.max(np.abs(r_p))), mu=mu,
                                       delta=delta, alpha=alpha if accepted else 0.0, ap=ap, ad=ad, dz=float(np.max(np.abs(dz))),
                                       theta=violation(g, m_el), z=z.copy()))
            if not accepted:
                if resto:
                    if n_retry >= P["resto_retry"]:
                        break
                    # the same z again, Levenberg-damped (the retry is an iteration of its own: everything is re-evaluated)
                    n_retry, delta_force = n_retry + 1, max(1.0, 100.0 * max(delta_force, delta))
                    n_retried += 1
                    continue
                want_resto = True
            elif not resto:
                # IPOPT enters the restoration when the step length falls below its alpha_min; here: resto_small_iter consecutive
                # accepted steps shorter than resto_small_alpha at an infeasible iterate
                n_small = n_small + 1 if (alpha < P["resto_small_alpha"] and violation(g, m_el) > theta_tol) else 0
                if n_small >= P["resto_small_iter"] and n_resto < P["resto_max"] and e_best > P["acceptable_tol"]:
                    want_resto = True                                       # (the accepted step is not taken)
        if want_resto:
            # the regular phase cannot continue from z.  Nothing to restore at a feasible point (kinks of step(), round-off
            # at the precision limit) or once the restoration has been entered resto_max times.
            theta_R = violation(g, m_el)
            if e_best <= P["acceptable_tol"] or theta_R <= theta_tol or n_resto >= P["resto_max"]:
                break
            resto, n_resto, it_resto, n_small = True, n_resto + 1, it, 0
            delta_force, n_retry, theta_ref, n_stall = 0.0, 0, theta_R, 0
            z_R, mu_reg = z.copy(), mu
            mu = max(mu, float(np.max(np.maximum(0.0, -g[:m_el]))))        # IPOPT: mu_R = max(mu, |c|_inf)
            s = np.where(el, _resto_central_path(g, mu, rho_R), np.maximum(g, 1e-2))   # elastic rows start on their central path,
            t_ = np.where(el, s - g, 0.0)                                   # the box rows like at the start of the solve
            lam = mu / s
            nu, n_acc = 10.0, 0
            continue
        z, s = z + alpha * dz, (st if (sreset and not resto) or (sreset_r and resto) else s + alpha * ds)
        delta_force, n_retry = 0.0, 0
        lam = lam + ad * dlam
        lam = np.minimum(np.maximum(lam, mu / (1e10 * s)), 1e10 * mu / s)   # IPOPT eq. (16) safeguard
        if resto:
            t_ = np.where(el, t_ + alpha * dt_, 0.0)
            # the same safeguard for the multiplier rho_R - lam of t
            tn = np.where(el, t_, 1.0)
            lam = np.where(el, np.minimum(np.maximum(lam, rho_R - 1e10 * mu / tn), rho_R - mu / (1e10 * tn)), lam)
            lam = np.where(el, np.minimum(np.maximum(lam, 1e-300), rho_R * (1.0 - 1e-15)), lam)
    if status != STATUS_OPTIMAL and status != STATUS_INFEASIBLE and e_best <= P["acceptable_tol"] and not resto:
        # stalled at the precision limit (ill-conditioned condensed system at mu ~ 1e-9): the best iterate is
        # within the acceptable tolerance, like IPOPT's acceptable_tol exit
        z, status, err = z_best, STATUS_OPTIMAL, e_best
    ev = evaluate(x0, z, u_prev, goal, obs, P, level=0)
    u0 = z[0:P.get("nu", 2)].copy()
    if return_info:
        return u0, status, it, dict(z=z, X=ev["X"], f=ev["f"], g=ev["g"], lam=lam / sf0, s=s, err=err, mu=mu,
                                    n_eval=n_eval, scale=sf0, obs=obs, n_resto=n_resto, it_resto=it_resto, in_resto=resto, stalled=n_stalled, retried=n_retried,
                                    theta=violation(ev["g"], m_el))
    return u0, status, it
