"""Float64 statement of the optimal-decay MPC-CBF NLP (position_control/optimal_decay_mpc_cbf.py) for the two models of its accept
list whose DT barrier steps the state with the robot's own step(): KinematicBicycle2D and Quad2D (:19).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **Parity unpinned and oracle-only**: do-mpc / casadi / IPOPT are absent and the
reference copy is stale (five 5-wide obstacle slots) -- as for DynamicUnicycle2D (oracle/od_mpc_cbf.py) the
obstacle rows are the 7-wide ones of MPCCBF.  What is restated:

  model       x+ = x + (f(x) + g(x) u) dt                                                    :135-141
  decay vars  omega1_k, omega2_k, two extra inputs per stage (here rho1_k, rho2_k)            :123-124
  cost        sum (x_k - goal)' Q (x_k - goal), stage and terminal                            :147-148,174-176
              + sum_k sum_i R_i u_{k,i}^2  (an expression r-term, not do-mpc's delta-u penalty) :178-179,185
              + sum_k p_sb1 (rho1_k - omega1)^2 + p_sb2 (rho2_k - omega2)^2                   :181-186
  weights     KB: Q = diag(50, 50, 1, 1), R = (0.5, 50); Quad2D: Q = diag(25, 25, 50, 10, 10, 50), R = (0.5, 0.5); N = 10   :37-42,23
              omega = 1, p_sb = 10                                                             :88-91
  CBF         dd_h + (a1 rho1 + a2 rho2) d_h + a1 a2 rho1 rho2 h >= 0 per stage and obstacle  :291-297
              h_k, d_h, dd_h from the robot's agent_barrier_dt: x1 = step(x_k, u_k), x2 = step(x1, u_k)
              (kinematic_bicycle2D.py:175-199, quad2D.py:179-206); DT gains KB 0.05, Quad2D 0.15   :66-68,72-74
  bounds      KB |v_k| <= v_max, |a| <= a_max, |beta| <= beta_max; Quad2D f_min <= u <= f_max; rho free   :211-224

With s_k = a1 rho1_k + a2 rho2_k and q_k = a1 a2 rho1_k rho2_k a row is  h(c_k) + (s_k - 2) h(b_k) + (1 - s_k + q_k) h(a_k):
the MPC-CBF row of oracle/mpc_gn.py with stage-dependent weights, so the problem functions are that module's evaluate() (per-
stage weights, R u^2 term) plus the rho-columns below.  Solver: oracle/od_mpc_cbf.py: solve (decay blocks eliminated per stage).
"""
import numpy as np

from . import mpc_cbf as M
from . import mpc_gn as G
from . import od_mpc_cbf as O


def kb_model(spec=None, dt=0.05):
    m = G.kb_model(spec, dt)
    m.update(R=np.array([0.5, 50.0]), alpha1=0.05, alpha2=0.05)          # optimal_decay_mpc_cbf.py:37-39,66-68
    return m


def quad2d_model(spec=None, dt=0.05):
    return G.quad2d_model(spec, dt)                                        # same Q, R and gains as MPCCBF (:40-42,72-74)


def params(model, N=10, **over):
    P = G.params(model, N)
    P.update(omega1=1.0, omega2=1.0, p_sb1=10.0, p_sb2=10.0, rterm="u")
    P["slack_reset"] = 0            # the optimal-decay solver of KinematicBicycle2D / Quad2D has no slack reset (VTOL2D switches it on: od_mpc_vtol.py)
    P.update(over)
    return P


def stage_weights(rho, P):
    a1, a2 = P["alpha1"], P["alpha2"]
    s = a1 * rho[:, 0] + a2 * rho[:, 1]
    q = a1 * a2 * rho[:, 0] * rho[:, 1]
    return np.stack([1.0 - s + q, s - 2.0, np.ones_like(s)], axis=1)


def evaluate(x0, zz, u_prev, goal, obs, P, lam=None, level=2):
    """Problem functions at zz = (z | rho); levels and row order of oracle.mpc_gn.evaluate."""
    N = P["N"]
    n = int(P.get("nu", 2)) * N
    a1, a2 = P["alpha1"], P["alpha2"]
    pen = np.array([P["p_sb1"], P["p_sb2"]]); ref = np.array([P["omega1"], P["omega2"]])
    z, rho = zz[:n], zz[n:].reshape(N, 2)
    K = obs.shape[0]
    Pb = dict(P, stage_w=stage_weights(rho, P))
    lam_b = None if lam is None else lam
    b = G.evaluate(x0, z, u_prev, goal, obs, Pb, lam_b, level)
    out = dict(f=b["f"] + float(np.sum(pen * (rho - ref) ** 2)), g=b["g"], X=b["X"], pts=b["pts"])
    if level == 0:
        return out
    hv = b["hv"]                                                            # (N, 3, K): h at a_k, b_k, c_k
    m = b["g"].shape[0]
    J = np.zeros((m, n + 2 * N))
    J[:, :n] = b["J"]
    A = np.zeros((N, K, 2))                                                 # d row / d rho_i = a_i (h_b - h_a) + a1 a2 rho_other h_a
    for k in range(N):
        A[k, :, 0] = a1 * (hv[k, 1] - hv[k, 0]) + a1 * a2 * rho[k, 1] * hv[k, 0]
        A[k, :, 1] = a2 * (hv[k, 1] - hv[k, 0]) + a1 * a2 * rho[k, 0] * hv[k, 0]
        J[k * K:(k + 1) * K, n + 2 * k] = A[k, :, 0]
        J[k * K:(k + 1) * K, n + 2 * k + 1] = A[k, :, 1]
    grad = np.concatenate([b["grad"], (2.0 * pen * (rho - ref)).reshape(-1)])
    out.update(grad=grad, J=J)
    if level == 1:
        return out
    lamv = np.zeros(m) if lam is None else lam
    lc = lamv[: N * K].reshape(N, K)
    W = np.zeros((n + 2 * N, n + 2 * N))
    W[:n, :n] = b["W"]
    JP = b["JP"]                                                            # (N, 3, K, n): gradient of h at each point in z
    for k in range(N):
        for i, (ai, other) in enumerate(((a1, rho[k, 1]), (a2, rho[k, 0]))):
            # d2 row / d rho_i dz = a_i (grad h_b - grad h_a) + a1 a2 rho_other grad h_a
            cross = ai * (JP[k, 1] - JP[k, 0]) + a1 * a2 * other * JP[k, 0]     # (K, n)
            W[:n, n + 2 * k + i] = -lc[k] @ cross
            W[n + 2 * k + i, :n] = W[:n, n + 2 * k + i]
            W[n + 2 * k + i, n + 2 * k + i] = 2.0 * pen[i]
        mix = -a1 * a2 * float(lc[k] @ hv[k, 0])                              # d2 row / d rho_1 d rho_2 = a1 a2 h_a
        W[n + 2 * k, n + 2 * k + 1] = mix
        W[n + 2 * k + 1, n + 2 * k] = mix
    out.update(W=W)
    return out


def solve(model, x0, u_prev, goal, obs, N=10, params_over=None, return_info=False, linear_algebra="schur"):
    """Returns u_0 (2,), rho_0 (2,), status, iterations [, info]."""
    P = params(model, N, **(params_over or {}))
    P["a_max"], P["w_max"] = 0.0, 0.0                                      # (unused: the box comes from u_lo / u_hi)
    return O.solve(x0, u_prev, goal, obs, params=P, return_info=return_info, linear_algebra=linear_algebra, evaluate_fn=evaluate)
