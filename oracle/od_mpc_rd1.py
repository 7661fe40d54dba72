"""Optimal-decay MPC-CBF for the relative-degree-1 models (Unicycle2D, Quad3D): float64 statement and solver.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **EXTENSION, parity unpinned.**  BASELINE config 5 asks for a
Unicycle2D + Quad3D fleet under optimal-decay MPC-CBF.  The reference's OptimalDecayMPCCBF
(position_control/optimal_decay_mpc_cbf.py) names Quad3D in its model list (:19) but gives the relative-degree-1
models the PLAIN row  d_h + alpha h_k >= 0  (:284-287: its omega inputs only appear in the rel-degree-2 branch,
:288-297), and rejects Unicycle2D.  SURVEY 8d therefore defines config 5 as a build extension.  The semantics built
here follow the reference where it has them:

  decay variable  one extra input per stage, rho_k ("omega1"; optimal_decay_mpc_cbf.py:123)
  CBF row         d_h + alpha rho_k h_k >= 0  -- the rel-degree-1 form of the reference's own optimal-decay CBF-QP
                  (position_control/optimal_decay_cbf_qp.py:96-101,113-125: `A u + b + alpha h omega1 >= 0`, one
                  variable, penalty on omega1 only :66-70)
  cost            stage/terminal (x - goal)' Q (x - goal) (:147-148,174-176) + sum_k R u_k^2 (:178-179, an expression
                  r-term, not do-mpc's delta-u penalty) + sum_k p_sb1 (rho_k - omega1)^2 (:181)
  weights/gains   Q, R, alpha, bounds: those of MPCCBF for the model (mpc_cbf.py, pinned by tests/golden/mpc_functions.npz);
                  omega1 = 1, p_sb1 = 10 (:88-89); rho free (no bounds, like the reference's omega inputs)
  obstacles       K rows of the 7-wide format, circles and superellipsoids (SURVEY 8d "generalised to 7-wide obstacles")
  barrier points  h_k = h(a_k), d_h = h(b_k) - h(a_k) with the model's own points: Unicycle2D a_k = p_k, b_k = p_{k+1}
                  (unicycle2D.py:127-145); Quad3D a_k = pos(x_k), b_k = pos(RK4 step) (quad3D.py:121-158,275-296)

The problem functions are those of oracle/mpc_cbf_uni.py / oracle/mpc_lin.py with a per-stage gain alpha rho_k; this
module adds the derivatives in rho and the interior-point method of oracle/mpc_cbf.py on zz = (z | rho): the decay
variable of a stage only meets that stage's rows and enters them linearly, so its (positive) scalar block
D_k = 2 sf p_sb + sum_j sigma_kj (alpha h_kj)^2 is eliminated first -- the Schur complement onto the inputs is the
n x n system the plain kernels already factor.  ``linear_algebra="dense"`` solves the same Newton system without the
elimination (cross-check in tests/test_oracle_od_rd1.py).
"""
import numpy as np

from . import mpc_cbf as M
from . import mpc_cbf_uni as MU
from . import mpc_lin as L

STATUS_OPTIMAL, STATUS_INFEASIBLE, STATUS_INACCURATE = M.STATUS_OPTIMAL, M.STATUS_INFEASIBLE, M.STATUS_INACCURATE

OD_DEFAULTS = dict(omega1=1.0, p_sb1=10.0, rterm="u2",
                   slack_reset=2)   # round 4: the line search resets slacks (oracle/mpc_cbf.py: solve).  On the config-5 batches (N = 20,
                                    # superellipsoids) Quad3D went from 245 of 256 optimal, 33.5 iterations mean, 477 max to 256 of 256, 14.9, 34;
                                    # Unicycle2D from 255, 31.3, 222 to 256, 20.7, 40.


def uni_params(N=10, **over):
    P = dict(MU.DEFAULTS, **OD_DEFAULTS, N=N, nu=2)
    P.update(over)
    return P


def lin_params(model, N=10, **over):
    """model: oracle.mpc_lin model dict (quad3d_model() with circles_only switched off for superellipsoids)."""
    return L.params(model, N, **dict(OD_DEFAULTS, **over))


def _base(P):
    return L.evaluate if "model" in P else MU.evaluate


def evaluate(x0, zz, goal, obs, P, lam=None, level=2):
    """Problem functions at zz = (z | rho_0..rho_{N-1}); same levels and row order as the base evaluate."""
    N = P["N"]
    nu = P.get("nu", 2)
    n = N * nu
    al, ps, ref = P["alpha"], P["p_sb1"], P["omega1"]
    z, rho = zz[:n], zz[n:]
    Pk = dict(P, alpha_k=al * rho, want_internals=True)
    ev = _base(P)(x0, z, np.zeros(nu), goal, obs, Pk, lam, level)
    K = obs.shape[0]
    out = dict(f=ev["f"] + float(ps * np.sum((rho - ref) ** 2)), g=ev["g"], X=ev["X"], pts=ev["pts"])
    if level == 0:
        return out
    m = ev["g"].shape[0]
    ha, Ja = ev["ha"], ev["Ja"]
    grad = np.concatenate([ev["grad"], 2.0 * ps * (rho - ref)])
    J = np.zeros((m, n + N))
    J[:, :n] = ev["J"]
    for k in range(N):
        J[k * K:(k + 1) * K, n + k] = al * ha[k]                            # d row / d rho_k = alpha h(a_kj)
    out.update(grad=grad, J=J, ha=ha)
    if level == 1:
        return out
    lam = np.zeros(m) if lam is None else lam
    lc = lam[: N * K].reshape(N, K)
    W = np.zeros((n + N, n + N))
    W[:n, :n] = ev["W"]
    for k in range(N):
        W[n + k, n + k] = 2.0 * ps
        c = -al * np.einsum("j,jn->n", lc[k], Ja[k])                         # -sum_j lam_kj d2 row / d rho_k d z
        W[:n, n + k] = c
        W[n + k, :n] = c
    out.update(W=W)
    return out


def solve(x0, u_prev, goal, obs, P, return_info=False, linear_algebra="schur"):
    """One solve.  Returns u_0 (nu,), rho_0, status, iterations [, info]."""
    N = P["N"]
    nu = P.get("nu", 2)
    n = N * nu
    x0 = np.asarray(x0, dtype=np.float64)
    obs = np.asarray(obs, dtype=np.float64)
    if "u_hi" in P:
        lo_, hi_ = np.tile(np.asarray(P["u_lo"], dtype=np.float64), N), np.tile(np.asarray(P["u_hi"], dtype=np.float64), N)
        z = np.clip(np.tile(np.asarray(u_prev, dtype=np.float64)[:nu], N), lo_ + 0.005 * (hi_ - lo_), hi_ - 0.005 * (hi_ - lo_))
    else:
        ub = np.tile([P["a_max"], P["w_max"]], N)
        z = np.clip(np.tile(np.asarray(u_prev, dtype=np.float64), N), -0.99 * ub, 0.99 * ub)   # set_initial_guess
    zz = np.concatenate([z, np.full(N, P["omega1"])])
    ev = evaluate(x0, zz, goal, obs, P, None, level=1)
    circles_only = "model" in P and P["model"]["circles_only"]
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
    tau, nu_m, delta_last = 0.995, 10.0, 0.0
    n_acc = 0
    err = np.inf
    e_best, zz_best = np.inf, zz.copy()
    Hq = P.get("quadratic_cost")
    for it in range(1, P["max_iter"] + 1):
        ev = evaluate(x0, zz, goal, obs, P, lam / sf, level=2)
        f, grad, W, g, J = sf * ev["f"], sf * ev["grad"], sf * ev["W"], ev["g"], ev["J"]
        r_d = grad - J.T @ lam
        r_p = g - s
        e_opt = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam)))
        e_mu = max(np.max(np.abs(r_d)), np.max(np.abs(r_p)), np.max(np.abs(s * lam - mu)))
        err = e_opt
        if e_opt < e_best:
            e_best, zz_best = e_opt, zz.copy()
        if e_opt <= P["tol"]:
            status = STATUS_OPTIMAL
            break
        n_acc = n_acc + 1 if e_opt <= P["acceptable_tol"] else 0
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
        Muu, Mur, d = Mb[:n, :n], Mb[:n, n:], np.diag(Mb[n:, n:]).copy()    # the rho block is diagonal and positive
        delta, dzz = 0.0, None
        for _try in range(40):
            try:
                if linear_algebra == "dense":
                    full = np.block([[Muu + delta * np.eye(n), Mur], [Mur.T, np.diag(d)]])
                    Lf = np.linalg.cholesky(full)
                    dzz = np.linalg.solve(Lf.T, np.linalg.solve(Lf, rhs))
                else:
                    S = Muu + delta * np.eye(n) - (Mur / d) @ Mur.T
                    Lc = np.linalg.cholesky(S)
                    du = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs[:n] - Mur @ (rhs[n:] / d)))
                    dr = (rhs[n:] - Mur.T @ du) / d
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
        nu_m = max(nu_m, 1.1 * float(np.max(np.abs(lam))))
        srp, dbar = float(np.sum(np.abs(r_p))), float(grad @ dzz - mu * np.sum(ds / s))
        if dbar - nu_m * srp >= 0.0 and srp > 0.0:
            nu_m = dbar / (0.9 * srp)                     # no descent direction of the merit: raise the penalty (oracle/mpc_cbf.py: solve)
        phi0 = f - mu * np.sum(np.log(s)) + nu_m * srp
        dphi = dbar - nu_m * srp
        # linear models: the cost is exactly quadratic in (z, rho) -- merit differences without cancellation (oracle/mpc_cbf.py: solve)
        curv = sf * (float(dzz[:n] @ Hq @ dzz[:n]) + 2.0 * P["p_sb1"] * float(dzz[n:] @ dzz[n:])) if Hq is not None else 0.0
        noise_rows = P.get("row_noise", 0.0) * nu_m * float(np.sum(np.abs(g)))
        alpha, accepted = ap, False
        for _ in range(12):
            zt, st = zz + alpha * dzz, s + alpha * ds
            e0 = evaluate(x0, zt, goal, obs, P, level=0)
            if P.get("slack_reset", 0) == 2:                               # s_i = g_i where g_i >= mu / nu: the minimiser of the merit function in s
                st = np.where(e0["g"] >= mu / nu_m, e0["g"], st)
            if Hq is not None:
                phit = phi0 + alpha * float(grad @ dzz) + 0.5 * alpha * alpha * curv \
                    - mu * float(np.sum(np.log(st) - np.log(s))) \
                    + nu_m * float(np.sum(np.abs(e0["g"] - st)) - np.sum(np.abs(r_p)))
            else:
                phit = sf * e0["f"] - mu * np.sum(np.log(st)) + nu_m * np.sum(np.abs(e0["g"] - st))
            if phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * abs(phi0) + noise_rows:
                accepted = True
                break
            alpha *= 0.5
        if not accepted:
            break
        zz, s = zz + alpha * dzz, (st if P.get("slack_reset", 0) == 2 else s + alpha * ds)
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
    u0, rho0 = zz[0:nu].copy(), float(zz[n])
    if return_info:
        return u0, rho0, status, it, dict(zz=zz, z=zz[:n].copy(), rho=zz[n:].copy(), X=ev["X"], f=ev["f"], g=ev["g"], lam=lam / sf, s=s, err=err, mu=mu, scale=sf, obs=obs)
    return u0, rho0, status, it
