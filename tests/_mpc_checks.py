"""Solver-independent optimality checks for the restated MPC-CBF NLP (shared by CPU and GPU tests).

Neither check uses the interior-point method of oracle/mpc_cbf.py or of the kernel:

* ``kkt_residual``: first-order conditions at a reported point z -- multipliers of the active inequalities by
  non-negative least squares (scipy.optimize.nnls), residual |grad f - J_A' lam|_inf relative to |grad f|_inf.
* ``slsqp_from``: scipy SLSQP on the same single-shooting functions from a given start.
"""
import numpy as np
from scipy.optimize import minimize, nnls


def kkt_residual(evaluate, z, active_tol=1e-6):
    """evaluate(z, level) -> dict(f, g, grad, J).  Returns (relative stationarity residual, min g, n_active)."""
    ev = evaluate(z, 1)
    g, J, grad = ev["g"], ev["J"], ev["grad"]
    scale = np.maximum(1.0, np.abs(J).max(axis=1))
    act = g <= active_tol * scale
    if not act.any():
        return float(np.abs(grad).max() / max(1.0, np.abs(grad).max())) if np.abs(grad).max() > 0 else 0.0, float(g.min()), 0
    A = J[act].T                                             # (n, n_act)
    col = np.maximum(np.linalg.norm(A, axis=0), 1e-300)
    lam, _ = nnls(A / col, grad, maxiter=50 * A.shape[1])
    res = grad - (A / col) @ lam
    return float(np.abs(res).max() / max(1.0, np.abs(grad).max())), float(g.min()), int(act.sum())


def slsqp_from(evaluate, z0, ftol=1e-13, maxiter=200):
    fun = lambda z: evaluate(z, 0)["f"]
    con = lambda z: evaluate(z, 0)["g"]
    jac = lambda z: evaluate(z, 1)["grad"]
    cjac = lambda z: evaluate(z, 1)["J"]
    return minimize(fun, z0, jac=jac, constraints=[{"type": "ineq", "fun": con, "jac": cjac}], method="SLSQP",
                    options={"ftol": ftol, "maxiter": maxiter})
