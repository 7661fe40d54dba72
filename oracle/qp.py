"""Exact solver for the tiny strictly-convex QPs behind the CBF-QP controller.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Problem (position_control/cbf_qp.py:47-66):

    minimise ||u - u_ref||^2   s.t.   G u + c >= 0        (u in R^n, n = 2)

where the rows of ``G, c`` are the CBF rows ``A1 u + b1 >= 0`` followed by the
input box written as four half-planes.  The reference hands this to GUROBI
through cvxpy (cbf_qp.py:190); that stack is not installable here.  The
objective is strictly convex, so the minimiser is unique and any exact method
gives the reference's answer up to the solver's 1e-6-class tolerances.  This
oracle enumerates active sets (none / one row / two rows), keeps the feasible
candidate with the smallest cost and reports infeasibility if there is none.
It is deliberately a different algorithm from the HIP kernel (which walks the
constraints incrementally), so agreement is a real check.
"""
import numpy as np

STATUS_OPTIMAL = 0
STATUS_INFEASIBLE = 1

FEAS_TOL = 1e-9


def box_rows(lo, hi):
    """Half-planes ``u_i - lo_i >= 0`` and ``hi_i - u_i >= 0`` for i = 0, 1."""
    G = np.array([[1.0, 0.0], [-1.0, 0.0], [0.0, 1.0], [0.0, -1.0]])
    c = np.array([-lo[0], hi[0], -lo[1], hi[1]], dtype=np.float64)
    return G, c


def _feasible(G, c, u, tol):
    r = G @ u + c
    scale = np.maximum(1.0, np.abs(G) @ np.abs(u) + np.abs(c))
    return bool(np.all(r >= -tol * scale))


def solve_qp2(G, c, u_ref, tol=FEAS_TOL):
    """Exact 2-variable projection QP by active-set enumeration.

    Returns ``(u, status)``; ``u`` is None when infeasible (cvxpy leaves
    ``u.value`` None in that case, tracking.py:627-634 only checks status).
    Non-finite rows make the problem infeasible (GUROBI would reject NaN data).
    """
    G = np.asarray(G, dtype=np.float64).reshape(-1, 2)
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    u_ref = np.asarray(u_ref, dtype=np.float64).reshape(2)
    if not (np.all(np.isfinite(G)) and np.all(np.isfinite(c)) and np.all(np.isfinite(u_ref))):
        return None, STATUS_INFEASIBLE
    m = G.shape[0]
    best, best_cost = None, np.inf

    def consider(u):
        nonlocal best, best_cost
        cost = float((u - u_ref) @ (u - u_ref))
        if cost < best_cost and _feasible(G, c, u, tol):
            best, best_cost = u, cost

    consider(u_ref.copy())
    n2 = np.einsum("ij,ij->i", G, G)
    for i in range(m):
        if n2[i] <= 0.0:
            continue
        lam = (G[i] @ u_ref + c[i]) / n2[i]
        consider(u_ref - lam * G[i])
    for i in range(m):
        for j in range(i + 1, m):
            det = G[i, 0] * G[j, 1] - G[i, 1] * G[j, 0]
            if abs(det) <= 1e-14 * np.sqrt(n2[i] * n2[j]):
                continue
            # G_i u = -c_i ; G_j u = -c_j
            u = np.array([(-c[i] * G[j, 1] + c[j] * G[i, 1]) / det,
                          (-c[j] * G[i, 0] + c[i] * G[j, 0]) / det])
            consider(u)
    if best is None:
        return None, STATUS_INFEASIBLE
    return best, STATUS_OPTIMAL


def feasibility_margin(G, c):
    """max_u min_i (G_i u + c_i)/||G_i|| over the 2-D plane (Chebyshev-centre LP).

    Used by the tests to set aside cases whose feasible/infeasible status sits
    within solver tolerance of flipping.  Solved with
    scipy.optimize.linprog; rows with ||G_i|| = 0 contribute ``c_i`` directly.
    """
    from scipy.optimize import linprog

    G = np.asarray(G, dtype=np.float64)
    if G.ndim != 2:
        G = G.reshape(-1, 2)
    n = G.shape[1]                                          # 2 for the planar models, 3 for Manipulator2D
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    nrm = np.sqrt(np.einsum("ij,ij->i", G, G))
    zero = nrm <= 0
    if np.any(c[zero] < 0):
        return float(np.min(c[zero]))
    Gn, cn, nn = G[~zero], c[~zero], nrm[~zero]
    # maximise t  s.t.  G u + c >= t*||G_i||   <=>  -G u + ||G_i|| t <= c
    A_ub = np.hstack([-Gn, nn[:, None]])
    res = linprog(c=[0.0] * n + [-1.0], A_ub=A_ub, b_ub=cn, bounds=[(None, None)] * (n + 1), method="highs")
    if res.status != 0:
        return float("nan")
    return float(res.x[n])


# ---------------------------------------------------------------------------
# n-variable projection QP (Manipulator2D: n = 3, up to 150 CBF rows + 6 box rows, cbf_qp.py:94-104)
# ---------------------------------------------------------------------------
def _enumerate_qpn(G, c, u_ref, tol):
    """Exact minimiser of ||u - u_ref||^2 s.t. G u + c >= 0 by enumerating every active set of size <= n.
    For each set S the equality-constrained minimiser is u_ref - G_S^T (G_S G_S^T)^-1 (G_S u_ref + c_S); the feasible
    candidate of smallest cost is the answer (strict convexity => unique).  Exponential in n: small row counts only."""
    from itertools import combinations
    m, n = G.shape
    best, best_cost = None, np.inf
    if _feasible(G, c, u_ref, tol):
        return u_ref.copy()
    for q in range(1, n + 1):
        for S in combinations(range(m), q):
            GS = G[list(S)]
            gram = GS @ GS.T
            if np.linalg.cond(gram) > 1e12:
                continue
            lam = np.linalg.solve(gram, GS @ u_ref + c[list(S)])
            u = u_ref - GS.T @ lam
            cost = float((u - u_ref) @ (u - u_ref))
            if cost < best_cost and _feasible(G, c, u, tol):
                best, best_cost = u, cost
    return best


def solve_qpn(G, c, u_ref, tol=FEAS_TOL, max_rounds=400):
    """Exact n-variable projection QP with many rows: constraint generation around the enumerator.

    Keep a working subset W of rows (initially none); solve the QP restricted to W exactly by enumeration; if its
    minimiser satisfies every row of the full problem it is the full minimiser (the restricted problem is a
    relaxation); otherwise add the most violated (normalised) row and repeat.  If a restricted problem is infeasible
    so is the full one.  W only ever needs the rows that are active somewhere along the way, a handful for n = 3.
    Deliberately a different algorithm from the HIP kernel (dual active set with rank-one steps)."""
    G = np.asarray(G, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    u_ref = np.asarray(u_ref, dtype=np.float64).reshape(-1)
    m, n = G.shape
    if not (np.all(np.isfinite(G)) and np.all(np.isfinite(c)) and np.all(np.isfinite(u_ref))):
        return None, STATUS_INFEASIBLE
    nrm = np.sqrt(np.einsum("ij,ij->i", G, G))
    if np.any((nrm == 0) & (c < -tol * np.maximum(1.0, np.abs(c)))):
        return None, STATUS_INFEASIBLE
    W = []
    u = u_ref.copy()
    for _ in range(max_rounds):
        r = G @ u + c
        scale = np.maximum(1.0, np.abs(G) @ np.abs(u) + np.abs(c))
        viol = r < -tol * scale
        if not np.any(viol):
            return u, STATUS_OPTIMAL
        score = np.where(viol & (nrm > 0), r / np.where(nrm > 0, nrm, 1.0), np.inf)
        score[W] = np.inf
        p = int(np.argmin(score))
        if not np.isfinite(score[p]):
            return None, STATUS_INFEASIBLE          # only rows already in W are violated: W itself is infeasible
        W.append(p)
        u = _enumerate_qpn(G[W], c[W], u_ref, tol)
        if u is None:
            return None, STATUS_INFEASIBLE
    raise RuntimeError("solve_qpn: constraint generation did not terminate")
