"""VTOL2D MPC-CBF: float64 statement of the problem functions, as a model for oracle/mpc_gn.evaluate.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The model functions are pinned on the reference's own code
(tests/golden/mpc_functions.npz: f, g, x_next, step, agent_barrier_dt, MPCCBF tables for VTOL2D); the solver is this repo's
interior point (parity unpinned: IPOPT absent).  The kernel that serves the model is csrc/mpc_vtol_wave.hip (DESIGN.md kernel 11), held to
this module problem by problem.  The first NLP of the reference's own example (examples/test_vtol.py) has no feasible point
(tests/test_oracle_mpc_vtol.py), so what the reference applies there is IPOPT's restoration output.

  dynamics    f, g of robots/vtol2D.py:118-311 (body velocity :333-343, lift blending :348-372, lift / drag / moment :374-401,
              wind -> inertial :410-419, rotors :424-452); prediction x+ = x + (f + g u) dt (mpc_cbf.py:135-141); step() adds the
              pitch wrap (:299-307), which the position-only barrier never sees
  barrier     h = |p - p_obs|^2 - beta (R + r)^2, beta = 1.01, rel-degree 2 through step o step (vtol2D.py:475-497)
  MPCCBF      N = 30, Q = diag(10, 10, 250, 10, 10, 50), R = (.5, .5, .5, 50000), alpha1 = alpha2 = 0.05 (mpc_cbf.py:40-43,83-87);
              bounds: throttles in [0, 1], elevator +-0.5; |x_dot| <= v_max, z_dot >= -descent_speed_max, |theta| <= pitch_max
              3.14159 / 180 (:222-233)

Jacobians come from forward-mode automatic differentiation over the ten (x, u) directions (class Dual below): the aero model
(atan2, exp blending, products of trigonometric terms) is differentiated exactly, with no hand-derived formula to get wrong.
"""
import math

import numpy as np

from . import mpc_cbf as M

GRAVITY = 9.81                                              # vtol2D.py:113


def default_spec(**over):
    """VTOL2D.__init__ defaults (vtol2D.py:56-111) + the radius examples/test_tracking.py passes for this model."""
    s = dict(mass=11.0, inertia=1.135, S_wing=0.55, rho=1.2682, C_L0=0.23, C_Lalpha=5.61, M=50.0, alpha_0=math.radians(15.0),
             C_Ldelta_e=0.13, C_D0=0.043, C_Dalpha=0.03, C_Ddelta_e=0.0, C_m0=0.0135, C_malpha=-2.74, C_mdelta_e=-0.99,
             chord=0.18994, k_front=70.0, k_rear=70.0, k_pusher=60.0, ell_f=0.5, ell_r=0.5, throttle_min=0.0, throttle_max=1.0,
             elevator_min=-0.5, elevator_max=0.5, v_max=15.0, pitch_max=15.0, descent_speed_max=5.0, radius=0.6)
    s.update(over)
    return s


class Dual:
    """value + gradient (forward mode); only the operations the VTOL2D model needs."""
    __slots__ = ("v", "d")
    __array_ufunc__ = None

    def __init__(self, v, d):
        self.v, self.d = float(v), d

    @staticmethod
    def lift(a, n):
        return a if isinstance(a, Dual) else Dual(a, np.zeros(n))

    def _o(self, o):
        return o if isinstance(o, Dual) else Dual(o, np.zeros_like(self.d))

    def __add__(self, o): o = self._o(o); return Dual(self.v + o.v, self.d + o.d)
    __radd__ = __add__
    def __sub__(self, o): o = self._o(o); return Dual(self.v - o.v, self.d - o.d)
    def __rsub__(self, o): o = self._o(o); return Dual(o.v - self.v, o.d - self.d)
    def __mul__(self, o): o = self._o(o); return Dual(self.v * o.v, self.d * o.v + o.d * self.v)
    __rmul__ = __mul__
    def __truediv__(self, o): o = self._o(o); return Dual(self.v / o.v, (self.d * o.v - o.d * self.v) / (o.v * o.v))
    def __rtruediv__(self, o): return self._o(o) / self
    def __neg__(self): return Dual(-self.v, -self.d)
    def __pow__(self, p): return Dual(self.v ** p, p * self.v ** (p - 1) * self.d)


class Dual2:
    """value + gradient + Hessian (forward mode, second order) over n directions: the exact-Hessian terms of the dynamics."""
    __slots__ = ("v", "d", "H")
    __array_ufunc__ = None

    def __init__(self, v, d, H):
        self.v, self.d, self.H = float(v), d, H

    def _o(self, o):
        return o if isinstance(o, Dual2) else Dual2(o, np.zeros_like(self.d), np.zeros_like(self.H))

    def chain(self, f, f1, f2):
        return Dual2(f, f1 * self.d, f1 * self.H + f2 * np.outer(self.d, self.d))

    def __add__(self, o): o = self._o(o); return Dual2(self.v + o.v, self.d + o.d, self.H + o.H)
    __radd__ = __add__
    def __sub__(self, o): o = self._o(o); return Dual2(self.v - o.v, self.d - o.d, self.H - o.H)
    def __rsub__(self, o): return self._o(o) - self
    def __neg__(self): return Dual2(-self.v, -self.d, -self.H)

    def __mul__(self, o):
        o = self._o(o)
        return Dual2(self.v * o.v, self.v * o.d + o.v * self.d, self.v * o.H + o.v * self.H + np.outer(self.d, o.d) + np.outer(o.d, self.d))
    __rmul__ = __mul__

    def recip(self):
        r = 1.0 / self.v
        return self.chain(r, -r * r, 2.0 * r * r * r)

    def __truediv__(self, o): return self * self._o(o).recip()
    def __rtruediv__(self, o): return self._o(o) * self.recip()
    def __pow__(self, p): return self.chain(self.v ** p, p * self.v ** (p - 1), p * (p - 1) * self.v ** (p - 2))


def _sin(a):
    if isinstance(a, Dual2): return a.chain(math.sin(a.v), math.cos(a.v), -math.sin(a.v))
    return Dual(math.sin(a.v), math.cos(a.v) * a.d) if isinstance(a, Dual) else math.sin(a)


def _cos(a):
    if isinstance(a, Dual2): return a.chain(math.cos(a.v), -math.sin(a.v), -math.cos(a.v))
    return Dual(math.cos(a.v), -math.sin(a.v) * a.d) if isinstance(a, Dual) else math.cos(a)


def _exp(a):
    if isinstance(a, Dual2): e = math.exp(a.v); return a.chain(e, e, e)
    return Dual(math.exp(a.v), math.exp(a.v) * a.d) if isinstance(a, Dual) else math.exp(a)


def _sqrt(a):
    if isinstance(a, Dual2): r = math.sqrt(a.v); return a.chain(r, 0.5 / r, -0.25 / (r * a.v))
    return Dual(math.sqrt(a.v), a.d / (2.0 * math.sqrt(a.v))) if isinstance(a, Dual) else math.sqrt(a)


def _atan2(y, x):
    if isinstance(y, Dual2) or isinstance(x, Dual2):
        ref = y if isinstance(y, Dual2) else x
        y, x = ref._o(y), ref._o(x)
        r2 = x.v * x.v + y.v * y.v
        ty, tx = x.v / r2, -y.v / r2                            # d atan2 / dy, / dx
        tyy, txx, txy = -2.0 * x.v * y.v / (r2 * r2), 2.0 * x.v * y.v / (r2 * r2), (y.v * y.v - x.v * x.v) / (r2 * r2)
        H = ty * y.H + tx * x.H + tyy * np.outer(y.d, y.d) + txx * np.outer(x.d, x.d) + txy * (np.outer(y.d, x.d) + np.outer(x.d, y.d))
        return Dual2(math.atan2(y.v, x.v), ty * y.d + tx * x.d, H)
    if isinstance(y, Dual) or isinstance(x, Dual):
        n = (y.d if isinstance(y, Dual) else x.d).shape[0]
        y, x = Dual.lift(y, n), Dual.lift(x, n)
        r2 = x.v * x.v + y.v * y.v
        return Dual(math.atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / r2)
    return math.atan2(y, x)


def _lift_drag_moment(V, alpha, delta_e, s):
    """vtol2D.py:348-401."""
    sig_a = _exp(-s["M"] * (alpha - s["alpha_0"]))
    sig_b = _exp(s["M"] * (alpha + s["alpha_0"]))
    sigma = (1 + sig_a + sig_b) / ((1 + sig_a) * (1 + sig_b))
    CL_lin = s["C_L0"] + s["C_Lalpha"] * alpha
    CL_non = 2 * _sin(alpha) * _cos(alpha)
    CL = (1 - sigma) * CL_lin + sigma * CL_non + s["C_Ldelta_e"] * delta_e
    CD = s["C_D0"] + s["C_Dalpha"] * (alpha ** 2) + s["C_Ddelta_e"] * delta_e
    CM = s["C_m0"] + s["C_malpha"] * alpha + s["C_mdelta_e"] * delta_e
    qbar = 0.5 * s["rho"] * (V ** 2)
    return qbar * s["S_wing"] * CL, qbar * s["S_wing"] * CD, qbar * s["S_wing"] * CM * s["chord"]


def _wind_to_inertial(theta, alpha, fx_w, fz_w):
    h = theta + alpha
    c, sn = _cos(h), _sin(h)
    return c * fx_w - sn * fz_w, sn * fx_w + c * fz_w


def fg(x, s):
    """f (6,), g (6 x 4 as a list of columns' non-zero rows) at x -- entries are floats or Duals."""
    th, xd, zd, thd = x[2], x[3], x[4], x[5]
    c, sn = _cos(th), _sin(th)
    u_b, w_b = c * xd + sn * zd, -sn * xd + c * zd
    V = _sqrt(u_b * u_b + w_b * w_b)
    alpha = _atan2(-w_b, u_b)
    L0, D0, M0 = _lift_drag_moment(V, alpha, 0.0, s)
    fx, fz = _wind_to_inertial(th, alpha, -D0, L0)
    m, I = s["mass"], s["inertia"]
    f = [xd, zd, thd, fx / m, (fz - m * GRAVITY) / m, M0 / I]
    Le, De, Me = _lift_drag_moment(V, alpha, 1.0, s)
    ex, ez = _wind_to_inertial(th, alpha, -De, Le)
    g = [[(-sn * s["k_front"]) / m, (c * s["k_front"]) / m, (s["ell_f"] * s["k_front"]) / I],
         [(-sn * s["k_rear"]) / m, (c * s["k_rear"]) / m, (-s["ell_r"] * s["k_rear"]) / I],
         [(c * s["k_pusher"]) / m, (sn * s["k_pusher"]) / m, 0.0],
         [ex / m, ez / m, Me / I]]                             # rows 3, 4, 5 of each column
    return f, g


def f_g_numeric(x, s):
    f, gc = fg([float(v) for v in x], s)
    g = np.zeros((6, 4))
    for j in range(4):
        g[3:, j] = gc[j]
    return np.array(f, dtype=float), g


def vt_F(x, u, spec, dt, jac=False):
    """prediction x + (f + g u) dt, with the Jacobians A = dF/dx, B = dF/du when asked."""
    if not jac:
        f, g = f_g_numeric(x, spec)
        return np.asarray(x, dtype=float) + (f + g @ np.asarray(u, dtype=float)) * dt
    n = 10
    xs = [Dual(x[i], np.eye(n)[i]) for i in range(6)]
    us = [Dual(u[j], np.eye(n)[6 + j]) for j in range(4)]
    f, gc = fg(xs, spec)
    out = []
    for i in range(6):
        acc = Dual.lift(f[i], n)
        if i >= 3:
            for j in range(4):
                acc = acc + gc[j][i - 3] * us[j]
        out.append(xs[i] + acc * dt)
    xn = np.array([o.v for o in out])
    Jm = np.array([o.d for o in out])
    return xn, Jm[:, :6], Jm[:, 6:]


def vt_S(x, u, spec, dt, jac=False):
    """robot.step: the same Euler step + the pitch wrap (vtol2D.py:299-307); the wrap changes no derivative and no barrier value."""
    r = vt_F(x, u, spec, dt, jac)
    xn = (r[0] if jac else r).copy()
    xn[2] = ((xn[2] + math.pi) % (2.0 * math.pi)) - math.pi
    return (xn, r[1], r[2]) if jac else xn


def vt_H(x, u, spec, dt, c, step=False):
    """sum_i c_i * Hessian of F_i (or of step()'s i-th component: the pitch wrap has no derivative) in (x, u): 10 x 10."""
    n = 10
    Z = np.zeros((n, n))
    xs = [Dual2(x[i], np.eye(n)[i], Z) for i in range(6)]
    us = [Dual2(u[j], np.eye(n)[6 + j], Z) for j in range(4)]
    f, gc = fg(xs, spec)
    H = np.zeros((n, n))
    for i in range(3, 6):                                       # rows 0..2 of F are linear in (x, u)
        if c[i] == 0.0:
            continue
        acc = xs[0]._o(f[i])
        for j in range(4):
            acc = acc + gc[j][i - 3] * us[j]
        H += c[i] * dt * acc.H
    return H


def vtol_model(spec=None, dt=0.05):
    s = default_spec(**(spec or {}))
    pm = s["pitch_max"] * 3.14159 / 180                        # mpc_cbf.py:232-233
    return dict(name="VTOL2D", nx=6, nu=4, F=vt_F, S=vt_S, H=vt_H, spec=s, dt=dt, Q=np.array([10.0, 10.0, 250.0, 10.0, 10.0, 50.0]),
                R=np.array([0.5, 0.5, 0.5, 50000.0]), alpha1=0.05, alpha2=0.05, beta=1.01, radius=s["radius"],
                u_lo=np.array([s["throttle_min"]] * 3 + [s["elevator_min"]]), u_hi=np.array([s["throttle_max"]] * 3 + [s["elevator_max"]]),
                xb=[(3, -s["v_max"], s["v_max"]), (4, -s["descent_speed_max"], np.inf), (2, -pm, pm)], circles_only=True, exact=True)


def params(N=30, spec=None, dt=0.05, **over):
    """Solver parameters for VTOL2D: the exact Hessian (vt_H) and the slack reset of the line search are what make the shared interior
    point converge on this model (tools/exp_vtol.py: 18-26 iterations on feasible cruise / hover / climb probes; with either one
    missing, 100 iterations end at KKT errors of 1e-2 .. 1e+2)."""
    from . import mpc_gn as G
    # no slack reset inside the restoration for this model: with it the restoration converges -- to least-violation inputs such as
    # (1, 0, 0, -0.5) that pitch the aircraft past its limit when applied (tools/exp_vtol_closed_loop.py)
    P = G.params(vtol_model(spec, dt), N, exact_hessian=True, slack_reset=2, resto_slack_reset=False, resto_retry=0, resto_stall_iter=0, resto_gn=True)
    P.update(over)
    return P


def solve(x0, u_prev, goal, obs, N=30, spec=None, dt=0.05, params_over=None, return_info=False):
    from . import mpc_gn as G
    P = params(N, spec, dt, **(params_over or {}))
    return M.solve(x0, u_prev, goal, obs, params=P, return_info=return_info, evaluate_fn=G.evaluate)
