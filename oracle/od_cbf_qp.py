"""Float64 restatement of OptimalDecayCBFQP (position_control/optimal_decay_cbf_qp.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference class is STALE: control_step hands it the (k,7) array of nearest
obstacles while its barrier call expects one obstacle, so it cannot run as checked in, and its solver
(cvxpy -> GUROBI) is not installable.  Semantics restated here, for ONE obstacle (the nearest, row 0):

  variables   u (2), omega1, omega2                                            :57-60
  objective   ||u - u_ref||^2 + p_sb1 (omega1 - 1)^2 + p_sb2 (omega2 - 1)^2      :72-76  (p_sb = 1e4, :21-24)
  rel-deg 2   A u + b + (a1 + a2) omega1 h_dot + a1 a2 h omega2 >= 0             :83-90,105-115  (DU, KB, Quad2D; a1 = a2 = .5)
              A = dh_dot_dx g, b = dh_dot_dx f                                   :141-146
  rel-deg 1   A u + b + alpha h omega1 >= 0, objective without the omega2 term   :65-70,99-104 (C3BF; alpha = .5)
  bounds      |u0| <= a_max, |u1| <= w_max | beta_max; Quad2D f_min <= u <= f_max   :88-89,96-97,110-113
  no obstacle A = b = h = h_dot = 0                                              :133-137

Parity: pinned by uniqueness of the minimiser of a strictly convex QP (exact active-set enumeration
below, cross-checked with scipy SLSQP in tests/test_oracle_od.py); no reference run exists.
"""
import itertools

import numpy as np

from . import robots as R
from .cbf_qp import input_bounds

STATUS_OPTIMAL, STATUS_INFEASIBLE = 0, 1


def default_param(model):
    """optimal_decay_cbf_qp.py:17-50."""
    if model in R.REL_DEG2:
        return dict(alpha1=0.5, alpha2=0.5, omega1=1.0, p_sb1=1e4, omega2=1.0, p_sb2=1e4)
    return dict(alpha=0.5, omega1=1.0, p_sb1=1e4)


def solve_diag_qp(D, r, G, c, tol=1e-9):
    """min sum_i D_i (x_i - r_i)^2  s.t.  G x + c >= 0, by enumeration of active sets (|S| <= n)."""
    D, r, G, c = (np.asarray(a, dtype=np.float64) for a in (D, r, G, c))
    n, m = len(D), len(c)
    if not (np.all(np.isfinite(G)) and np.all(np.isfinite(c)) and np.all(np.isfinite(r))):
        return None, STATUS_INFEASIBLE
    best, best_cost = None, np.inf
    for size in range(0, min(n, m) + 1):
        for S in itertools.combinations(range(m), size):
            S = list(S)
            if size == 0:
                x = r.copy()
                lam = np.zeros(0)
            else:
                GS = G[S]
                K = 0.5 * GS @ (GS / D).T
                if abs(np.linalg.det(K)) < 1e-14 * max(1.0, np.abs(K).max()) ** size:
                    continue
                lam = np.linalg.solve(K, -(GS @ r + c[S]))
                x = r + 0.5 * (GS / D).T @ lam
            if np.any(lam < -1e-9 * (1 + np.abs(lam).max() if size else 1)):
                continue
            res = G @ x + c
            if np.any(res < -tol * np.maximum(1.0, np.abs(G) @ np.abs(x) + np.abs(c))):
                continue
            cost = float(np.sum(D * (x - r) ** 2))
            if cost < best_cost:
                best, best_cost = x, cost
    if best is None:
        return None, STATUS_INFEASIBLE
    return best, STATUS_OPTIMAL


def solve(model, X, u_ref, obs, spec, param=None):
    """Returns dict(u, omega (2,), status, h).  ``obs`` is one 7-wide row or None."""
    P = dict(default_param(model))
    if param:
        P.update(param)
    X = np.asarray(X, dtype=np.float64).reshape(-1)
    u_ref = np.asarray(u_ref, dtype=np.float64).reshape(2)
    rel2 = model in R.REL_DEG2
    if obs is None:
        A = np.zeros(2); b = 0.0; h = 0.0; hdot = 0.0
    else:
        obs = np.asarray(obs, dtype=np.float64)
        fx, gx = R.f(model, X, spec), R.g(model, X, spec)
        if rel2:
            h, hdot, dhd = R.agent_barrier(model, X, obs, spec["radius"])
            A, b = dhd @ gx, float(dhd @ fx)
        else:
            h, dh = R.agent_barrier(model, X, obs, spec["radius"])
            hdot = 0.0
            A, b = dh @ gx, float(dh @ fx)
    lo, hi = input_bounds(model, spec)
    if rel2:
        e1 = (P["alpha1"] + P["alpha2"]) * hdot
        e2 = P["alpha1"] * P["alpha2"] * h
        D = np.array([1.0, 1.0, P["p_sb1"], P["p_sb2"]])
        r = np.array([u_ref[0], u_ref[1], P["omega1"], P["omega2"]])
        G = np.array([[A[0], A[1], e1, e2], [1, 0, 0, 0], [-1, 0, 0, 0], [0, 1, 0, 0], [0, -1, 0, 0]], dtype=np.float64)
    else:
        e1 = P["alpha"] * h
        D = np.array([1.0, 1.0, P["p_sb1"]])
        r = np.array([u_ref[0], u_ref[1], P["omega1"]])
        G = np.array([[A[0], A[1], e1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]], dtype=np.float64)
    c = np.array([b, -lo[0], hi[0], -lo[1], hi[1]])
    x, st = solve_diag_qp(D, r, G, c)
    if x is None:
        return dict(u=None, omega=None, status=st, h=h)
    om = np.array([x[2], x[3] if rel2 else 1.0])
    return dict(u=x[:2], omega=om, status=st, h=h)
