"""Float64 statement of the optimal-decay MPC-CBF NLP for VTOL2D -- the last model of the reference's accept list
(position_control/optimal_decay_mpc_cbf.py:19) -- as a model for oracle/od_mpc_gn.evaluate and oracle/od_mpc_cbf.solve.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned and oracle-only** (extension label, as every optimal-decay class):
do-mpc / casadi / IPOPT are absent and the reference copy is stale (five 5-wide obstacle slots).  What is restated:

  model       the tilt-rotor of robots/vtol2D.py:118-311 with the prediction x+ = x + (f + g u) dt          optimal_decay_mpc_cbf.py:135-141
  horizon     30, Q = diag(10, 10, 250, 10, 10, 50), R = (0.5, 0.5, 0.5, 50000)                              :44-47
  decay vars  omega1_k, omega2_k, two extra inputs per stage                                                   :123-124
  cost        sum (x_k - goal)' Q (x_k - goal) + sum_k R u_k^2 (an expression, not the delta-u penalty)      :147-148,173-179
              + sum_k p_sb1 (omega1_k - 1)^2 + p_sb2 (omega2_k - 1)^2, p_sb = 10                              :175-176,88-91
  CBF         dd_h + (a1 omega1 + a2 omega2) d_h + a1 a2 omega1 omega2 h >= 0, a1 = a2 = 0.35                 :83-86,288-296
              through step o step against K discs (vtol2D.py:475-497), obstacle rows 7 wide as in MPCCBF
  bounds      throttles in [0, 1], |elevator| <= 0.5, |x_dot| <= v_max, z_dot >= -descent_speed_max, |theta| <= pitch_max   :216-226
Solver: oracle/od_mpc_cbf.py: solve (decay blocks eliminated per stage, no restoration phase) with the exact Hessian of the aero
model and the slack reset of the line search that VTOL2D needs (oracle/mpc_vtol.py: params).
"""
import numpy as np

from . import mpc_gn as G
from . import mpc_vtol as V
from . import od_mpc_cbf as O
from . import od_mpc_gn as OG


def vtol_model(spec=None, dt=0.05):
    m = V.vtol_model(spec, dt)
    m.update(alpha1=0.35, alpha2=0.35)                                       # optimal_decay_mpc_cbf.py:83-86
    return m


def params(N=30, spec=None, dt=0.05, **over):
    P = G.params(vtol_model(spec, dt), N, exact_hessian=True)
    P.update(omega1=1.0, omega2=1.0, p_sb1=10.0, p_sb2=10.0, rterm="u", slack_reset=2)
    P["a_max"], P["w_max"] = 0.0, 0.0                                        # (unused: the box comes from u_lo / u_hi)
    P.update(over)
    return P


def solve(x0, u_prev, goal, obs, N=30, spec=None, dt=0.05, params_over=None, return_info=False, linear_algebra="schur"):
    """Returns u_0 (4,), rho_0 (2,), status, iterations [, info]."""
    P = params(N, spec, dt, **(params_over or {}))
    return O.solve(x0, u_prev, goal, obs, params=P, return_info=return_info, linear_algebra=linear_algebra, evaluate_fn=OG.evaluate)
