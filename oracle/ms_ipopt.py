"""The MPC-CBF NLP as do-mpc poses it -- MULTIPLE SHOOTING -- solved by a restatement of IPOPT's published algorithm.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).   **Parity unpinned**: do-mpc, casadi and IPOPT are absent from the image
(SURVEY 8c), so nothing here can be held to an execution of the reference's solver.  What this module is for: the other oracles
(oracle/mpc_cbf.py, mpc_gn.py, mpc_vtol.py) and every kernel solve the CONDENSED single-shooting problem in z = (u_0 .. u_{N-1})
from the rollout of u_prev with an l1-merit interior point.  The reference does not.  do-mpc (`state_discretization =
'discrete'`, `n_robust = 0`; position_control/mpc_cbf.py:162-174) hands IPOPT

    variables    x_0 .. x_N, u_0 .. u_{N-1}                       ((N + 1) nx + N nu)
    objective    sum_{k<N} l(x_k) + m(x_N) + sum_k (u_k - u_{k-1})' R (u_k - u_{k-1}),  l = m = (x - goal)' Q (x - goal)
                                                                  mpc_cbf.py:144,176-180;  u_{-1} = the input applied last
    equalities   x_0 = x0;   x_{k+1} = x_k + (f(x_k) + g(x_k) u_k) dt                    mpc_cbf.py:135-141
    inequalities -cbf_i(x_k, u_k) <= 0,  k = 0 .. N-1, i < num_obs                       mpc_cbf.py:295-325
    bounds       on every x_k (k = 0 .. N) and u_k                                       mpc_cbf.py:182-233
    start        x_k = x0 for EVERY k, u_k = u_prev                                      mpc_cbf.py:366-369 (set_initial_guess)

and IPOPT runs with its defaults (do-mpc only silences it).  Same stationary points as the condensed problem, different iterates,
different basins on a non-convex problem, different behaviour on infeasible starts.  This module restates that side:

  * `StageNLP`: the multiple-shooting problem for DynamicUnicycle2D (robots/dynamic_unicycle2D.py:42-78,188-238) and VTOL2D
    (robots/vtol2D.py:118-311,475-497), derivatives by second-order forward mode vectorised over the stages (class VD2);
    held to tests/golden/mpc_functions.npz like the other oracles (tests/test_oracle_ms.py).
  * `solve_nlp`: Waechter & Biegler, "On the implementation of an interior-point filter line-search algorithm for large-scale
    nonlinear programming", Math. Program. 106 (2006) -- the paper IPOPT's documentation cites as its algorithm -- with the
    option defaults of IPOPT 3.14's documentation: gradient-based scaling (max gradient 100), bound_push = bound_frac = 1e-2,
    bound_relax_factor 1e-8, bound multipliers 1, least-square equality multipliers (dropped above 1e3), mu_init 0.1 with the
    monotone update (kappa_mu .2, theta_mu 1.5, kappa_eps 10), tau_min .99, the filter line search (gamma_theta 1e-5, gamma_phi
    1e-8, delta 1, s_theta 1.1, s_phi 2.3, eta_phi 1e-8, theta_max / theta_min = 1e4 / 1e-4 max(1, theta_0), alpha_min_frac .05),
    second-order corrections (4, kappa_soc .99), inertia correction (Algorithm IC: 1e-4, x100 / x8 / 1/3, delta_c = 1e-8 mu^.25),
    kappa_sigma 1e10, kappa_d 1e-5, tol 1e-8 (+ dual_inf 1, constr_viol 1e-4, compl 1e-4 unscaled), acceptable 1e-6 x 15,
    max_iter 3000, and the feasibility restoration phase of section 3.3 (rho 1000, zeta = sqrt(mu), D_R = 1 / max(1, |x_R|),
    closed-form n / p start, leaves when the iterate is acceptable to the original filter with 10 % less infeasibility;
    multipliers 0 and bound multipliers kept (<= 1e3) on return; "converged to a point of local infeasibility" when it
    converges instead).
    NOT restated: the watchdog, the soft restoration phase, the tiny-step and iterative-refinement heuristics and MUMPS'
    pivoting (LAPACK's Bunch-Kaufman factorisation supplies the inertia here) -- the iterates of a real IPOPT run differ
    in the digits those choices touch.
  * Linear algebra: the slacks and the restoration's n / p are eliminated from the primal-dual system (each sits in one row
    with coefficient +-1 and a diagonal Hessian block), every constraint row is kept: a dense symmetric indefinite system of
    order n + m (792 for VTOL2D, 188 for config 3).
"""
import math

import numpy as np
from scipy.linalg import lapack

INF = np.inf


# ---- second-order forward mode, vectorised over a batch (the stages of the horizon) -----------------------------------
class VD2:
    """value (B,), gradient (B, n), Hessian (B, n, n)."""
    __slots__ = ("v", "d", "H")
    __array_ufunc__ = None

    def __init__(self, v, d, H):
        self.v, self.d, self.H = v, d, H

    @staticmethod
    def seed(vals):
        """vals (B, n) -> n independent variables."""
        B, n = vals.shape
        Z = np.zeros((B, n, n))
        eye = np.eye(n)
        return [VD2(vals[:, i].copy(), np.tile(eye[i], (B, 1)), Z) for i in range(n)]

    def _c(self, o):
        return VD2(np.full_like(self.v, o), np.zeros_like(self.d), np.zeros_like(self.H))

    def chain(self, f, f1, f2):
        return VD2(f, f1[:, None] * self.d, f1[:, None, None] * self.H + f2[:, None, None] * (self.d[:, :, None] * self.d[:, None, :]))

    def __add__(self, o):
        if isinstance(o, VD2):
            return VD2(self.v + o.v, self.d + o.d, self.H + o.H)
        return VD2(self.v + o, self.d, self.H)
    __radd__ = __add__

    def __sub__(self, o):
        if isinstance(o, VD2):
            return VD2(self.v - o.v, self.d - o.d, self.H - o.H)
        return VD2(self.v - o, self.d, self.H)

    def __rsub__(self, o):
        return VD2(o - self.v, -self.d, -self.H)

    def __neg__(self):
        return VD2(-self.v, -self.d, -self.H)

    def __mul__(self, o):
        if isinstance(o, VD2):
            x = self.d[:, :, None] * o.d[:, None, :]
            return VD2(self.v * o.v, self.v[:, None] * o.d + o.v[:, None] * self.d,
                       self.v[:, None, None] * o.H + o.v[:, None, None] * self.H + x + x.transpose(0, 2, 1))
        return VD2(self.v * o, self.d * o, self.H * o)
    __rmul__ = __mul__

    def recip(self):
        r = 1.0 / self.v
        return self.chain(r, -r * r, 2.0 * r * r * r)

    def __truediv__(self, o):
        if isinstance(o, VD2):
            return self * o.recip()
        return self * (1.0 / o)

    def __rtruediv__(self, o):
        return self.recip() * o

    def __pow__(self, p):
        return self.chain(self.v ** p, p * self.v ** (p - 1), p * (p - 1) * self.v ** (p - 2))


def _sin(a):
    return a.chain(np.sin(a.v), np.cos(a.v), -np.sin(a.v)) if isinstance(a, VD2) else np.sin(a)


def _cos(a):
    return a.chain(np.cos(a.v), -np.sin(a.v), -np.cos(a.v)) if isinstance(a, VD2) else np.cos(a)


def _exp(a):
    if isinstance(a, VD2):
        e = np.exp(a.v)
        return a.chain(e, e, e)
    return np.exp(a)


def _sqrt(a):
    if isinstance(a, VD2):
        r = np.sqrt(a.v)
        return a.chain(r, 0.5 / r, -0.25 / (r * a.v))
    return np.sqrt(a)


def _abs(a):
    if isinstance(a, VD2):
        sg = np.sign(a.v)
        return VD2(np.abs(a.v), sg[:, None] * a.d, sg[:, None, None] * a.H)
    return np.abs(a)


def _atan2(y, x):
    if isinstance(y, VD2) or isinstance(x, VD2):
        ref = y if isinstance(y, VD2) else x
        y = y if isinstance(y, VD2) else ref._c(y)
        x = x if isinstance(x, VD2) else ref._c(x)
        r2 = x.v * x.v + y.v * y.v
        ty, tx = x.v / r2, -y.v / r2
        tyy, txx, txy = -2.0 * x.v * y.v / (r2 * r2), 2.0 * x.v * y.v / (r2 * r2), (y.v * y.v - x.v * x.v) / (r2 * r2)
        oyy = y.d[:, :, None] * y.d[:, None, :]
        oxx = x.d[:, :, None] * x.d[:, None, :]
        oyx = y.d[:, :, None] * x.d[:, None, :]
        H = ty[:, None, None] * y.H + tx[:, None, None] * x.H + tyy[:, None, None] * oyy + txx[:, None, None] * oxx \
            + txy[:, None, None] * (oyx + oyx.transpose(0, 2, 1))
        return VD2(np.arctan2(y.v, x.v), ty[:, None] * y.d + tx[:, None] * x.d, H)
    return np.arctan2(y, x)


# ---- models: x+ = x + (f(x) + g(x) u) dt, written once for floats-arrays and for VD2 ---------------------------------------
def du_next(x, u, spec, dt):
    """DynamicUnicycle2D: f = [v cos th, v sin th, 0, 0], g u = [0, 0, w, a]   (dynamic_unicycle2D.py:42-73; U = [a, w])."""
    c, s = _cos(x[2]), _sin(x[2])
    return [x[0] + x[3] * c * dt, x[1] + x[3] * s * dt, x[2] + u[1] * dt, x[3] + u[0] * dt]


def _clip(a, lo, hi):
    """fmax(fmin(a, hi), lo) as casadi differentiates it: slope one inside, zero outside."""
    if isinstance(a, VD2):
        ins = ((a.v >= lo) & (a.v <= hi)).astype(float)
        return VD2(np.clip(a.v, lo, hi), ins[:, None] * a.d, ins[:, None, None] * a.H)
    return np.clip(a, lo, hi)


def si_next(x, u, spec, dt):
    """SingleIntegrator2D: x = (px, py), f = 0, g = I   (single_integrator2D.py:45-66; U = [vx, vy])."""
    return [x[0] + u[0] * dt, x[1] + u[1] * dt]


def uni_next(x, u, spec, dt):
    """Unicycle2D: x = (px, py, theta), f = 0, g u = [v cos th, v sin th, w]   (unicycle2D.py:42-68; U = [v, w])."""
    return [x[0] + u[0] * _cos(x[2]) * dt, x[1] + u[0] * _sin(x[2]) * dt, x[2] + u[1] * dt]


def kb_next(x, u, spec, dt):
    """KinematicBicycle2D: f = [v cos th, v sin th, 0, 0], g u = [-v sin th b, v cos th b, v b / L_r, a]   (kinematic_bicycle2D.py:67-110; U = [a, beta])."""
    c, s = _cos(x[2]), _sin(x[2])
    return [x[0] + (x[3] * c - x[3] * s * u[1]) * dt, x[1] + (x[3] * s + x[3] * c * u[1]) * dt, x[2] + (x[3] * u[1] / spec["rear_ax_dist"]) * dt, x[3] + u[0] * dt]


def kb_step(x, u, spec, dt):
    """robot.step as the DT barrier calls it (kinematic_bicycle2D.py:112-123,178-179): Euler, the heading wrap (touches no position), and the
    speed clipped to [v_min, v_max] -- the model's x_next (mpc_cbf.py:138) has neither."""
    xn = kb_next(x, u, spec, dt)
    xn[3] = _clip(xn[3], spec["v_min"], spec["v_max"])
    return xn


def di_next(x, u, spec, dt):
    """DoubleIntegrator2D: f = [vx, vy, 0, 0], g u = [0, 0, ax, ay]   (double_integrator2D.py:46-77)."""
    return [x[0] + x[2] * dt, x[1] + x[3] * dt, x[2] + u[0] * dt, x[3] + u[1] * dt]


def di_step(x, u, spec, dt):
    """robot.step as the DT barrier calls it (double_integrator2D.py:79-107,225-226): Euler, then the velocity rescaled to norm v_max where it
    is above it (casadi: if_else(v_mag > v_max, v_max / v_mag, 1)) -- the model's x_next (mpc_cbf.py:138) is the plain Euler step."""
    xn = di_next(x, u, spec, dt)
    vmag = _sqrt(xn[2] * xn[2] + xn[3] * xn[3])
    over = (vmag.v if isinstance(vmag, VD2) else vmag) > spec["v_max"]
    if isinstance(vmag, VD2):
        if over.any():
            sc = spec["v_max"] / vmag
            o = over.astype(float)
            mix = lambda a, b: VD2(o * a.v + (1 - o) * b.v, o[:, None] * a.d + (1 - o)[:, None] * b.d, o[:, None, None] * a.H + (1 - o)[:, None, None] * b.H)
            xn[2], xn[3] = mix(xn[2] * sc, xn[2]), mix(xn[3] * sc, xn[3])
    else:                                                                   # floats or arrays over the stages
        sc = np.where(over, spec["v_max"] / np.where(over, vmag, 1.0), 1.0)
        xn[2], xn[3] = xn[2] * sc, xn[3] * sc
    return xn


GRAVITY = 9.81                                                   # vtol2D.py:113


def _vt_ldm(V2, alpha, delta_e, s):
    """vtol2D.py:348-401 (lift blending, lift / drag / moment); V2 = V^2."""
    sig_a = _exp(-s["M"] * (alpha - s["alpha_0"]))
    sig_b = _exp(s["M"] * (alpha + s["alpha_0"]))
    sigma = (1.0 + sig_a + sig_b) / ((1.0 + sig_a) * (1.0 + sig_b))
    CL_lin = s["C_L0"] + s["C_Lalpha"] * alpha
    CL_non = 2.0 * _sin(alpha) * _cos(alpha)
    CL = (1.0 - sigma) * CL_lin + sigma * CL_non + s["C_Ldelta_e"] * delta_e
    CD = s["C_D0"] + s["C_Dalpha"] * (alpha * alpha) + s["C_Ddelta_e"] * delta_e
    CM = s["C_m0"] + s["C_malpha"] * alpha + s["C_mdelta_e"] * delta_e
    qS = (0.5 * s["rho"] * s["S_wing"]) * V2
    return qS * CL, qS * CD, qS * CM * s["chord"]


def vt_next(x, u, s, dt):
    """VTOL2D (vtol2D.py:118-311): body velocity :333-343, wind -> inertial :410-419, rotors :424-452."""
    th, xd, zd, thd = x[2], x[3], x[4], x[5]
    c, sn = _cos(th), _sin(th)
    u_b, w_b = c * xd + sn * zd, c * zd - sn * xd
    V2 = u_b * u_b + w_b * w_b                                  # (sqrt(.))^2 of the reference: equal to the last bit or two
    alpha = _atan2(-w_b, u_b)
    m, I = s["mass"], s["inertia"]
    L0, D0, M0 = _vt_ldm(V2, alpha, 0.0, s)
    Le, De, Me = _vt_ldm(V2, alpha, 1.0, s)
    hd = th + alpha
    ch, sh = _cos(hd), _sin(hd)
    fx, fz = -(ch * D0) - sh * L0, ch * L0 - sh * D0            # _wind_to_inertial(theta, alpha, -D, L)
    ex, ez = -(ch * De) - sh * Le, ch * Le - sh * De
    kf, kr, kp = s["k_front"], s["k_rear"], s["k_pusher"]
    ax = fx / m + (-(sn * kf) / m) * u[0] + (-(sn * kr) / m) * u[1] + ((c * kp) / m) * u[2] + (ex / m) * u[3]
    az = (fz - m * GRAVITY) / m + ((c * kf) / m) * u[0] + ((c * kr) / m) * u[1] + ((sn * kp) / m) * u[2] + (ez / m) * u[3]
    at = M0 / I + (s["ell_f"] * kf / I) * u[0] + (-s["ell_r"] * kr / I) * u[1] + (Me / I) * u[3]
    return [x[0] + xd * dt, x[1] + zd * dt, x[2] + thd * dt, x[3] + ax * dt, x[4] + az * dt, x[5] + at * dt]


def du_model(spec=None, dt=0.05):
    s = dict(v_max=1.0, a_max=1.0, w_max=0.5, radius=0.25)
    s.update(spec or {})
    return dict(name="DynamicUnicycle2D", nx=4, nu=2, next=du_next, spec=s, dt=dt, N=10, Q=np.array([50.0, 50.0, 0.01, 30.0]),
                R=np.array([0.5, 0.5]), alpha1=0.15, alpha2=0.15, beta=1.01, radius=s["radius"],
                u_lo=np.array([-s["a_max"], -s["w_max"]]), u_hi=np.array([s["a_max"], s["w_max"]]),
                x_lo=np.array([-INF, -INF, -INF, -s["v_max"]]), x_hi=np.array([INF, INF, INF, s["v_max"]]))


def si_model(spec=None, dt=0.05):
    """mpc_cbf.py:19-21 (Q, R), :49-51 (alpha = 0.05), :183-187 (input box), :312-315 (one-step rows), single_integrator2D.py:148 (beta = 1.01; the
    barrier has a superellipsoid branch: StageNLP._h serves it, the kernel does not)."""
    s = dict(v_max=1.0, radius=0.25)
    s.update(spec or {})
    return dict(name="SingleIntegrator2D", nx=2, nu=2, next=si_next, spec=s, dt=dt, N=10, Q=np.array([50.0, 50.0]), R=np.array([5.0, 5.0]),
                alpha=0.05, alpha1=0.0, alpha2=0.0, beta=1.01, radius=s["radius"],
                u_lo=np.array([-s["v_max"], -s["v_max"]]), u_hi=np.array([s["v_max"], s["v_max"]]), x_lo=np.full(2, -INF), x_hi=np.full(2, INF))


def uni_model(spec=None, dt=0.05):
    """mpc_cbf.py:22-24 (Q, R), :52-53 (alpha = 0.05), :188-192 (input box), :312-315 (d_h + alpha h_k >= 0: one step), unicycle2D.py:127 (beta = 1.01)."""
    s = dict(v_max=1.0, w_max=0.5, radius=0.25)
    s.update(spec or {})
    return dict(name="Unicycle2D", nx=3, nu=2, next=uni_next, spec=s, dt=dt, N=10, Q=np.array([50.0, 50.0, 0.01]), R=np.array([0.5, 0.5]),
                alpha=0.05, alpha1=0.0, alpha2=0.0, beta=1.01, radius=s["radius"], circles_only=True,
                u_lo=np.array([-s["v_max"], -s["w_max"]]), u_hi=np.array([s["v_max"], s["w_max"]]), x_lo=np.full(3, -INF), x_hi=np.full(3, INF))


def _where(mask, a, b):
    """a where mask else b, for arrays and VD2 alike (a piecewise expression as casadi's if_else / fmax differentiates it)."""
    if isinstance(a, VD2) or isinstance(b, VD2):
        ref = a if isinstance(a, VD2) else b
        a = a if isinstance(a, VD2) else ref._c(a)
        b = b if isinstance(b, VD2) else ref._c(b)
        m = np.asarray(mask, dtype=float)
        return VD2(m * a.v + (1 - m) * b.v, m[:, None] * a.d + (1 - m)[:, None] * b.d, m[:, None, None] * a.H + (1 - m)[:, None, None] * b.H)
    return np.where(mask, a, b)


def _val(a):
    return a.v if isinstance(a, VD2) else a


def _kb_rel(x, o):
    """p_rel, v_rel of the state against an obstacle row.  The obstacle's velocity is ZERO, as in the reference's MPC: set_cbf_constraint hands the
    barrier a 1 x 7 casadi slice whose `shape[0] > 3` test is False (oracle/mpc_kb_state.py: _rel; the goldens say the same)."""
    px, py = o[0] - x[0], o[1] - x[1]
    c, s = _cos(x[2]), _sin(x[2])
    vx, vy = 0.0 - x[3] * c, 0.0 - x[3] * s
    return px, py, vx, vy, px * px + py * py, _sqrt(vx * vx + vy * vy)


def h_c3bf(x, o, radius):
    """kinematic_bicycle2D_c3bf.py:83-109 (beta = 1.01): <p_rel, v_rel> + |p_rel| |v_rel| sqrt(max(|p_rel|^2 - ego^2, 0)) / |p_rel|; inside the
    inflated radius the root is held at zero with zero derivatives (oracle/mpc_kb_state.py)."""
    px, py, vx, vy, pm2, vm = _kb_rel(x, o)
    ego = (o[2] + radius) * 1.01
    pm = _sqrt(pm2)
    a = pm2 - ego * ego
    pos = _val(a) > 0.0
    root = _where(pos, _sqrt(_where(pos, a, 1.0)), 0.0)
    return px * vx + py * vy + pm * vm * root / pm


def h_dpcbf(x, o, radius):
    """kinematic_bicycle2D_dpcbf.py:91-136 (s = 1.05): line-of-sight frame, cos / sin of atan2(p_y, p_x) = p_x / |p|, p_y / |p|."""
    px, py, vx, vy, pm2, vm = _kb_rel(x, o)
    s = 1.05
    ego = (o[2] + radius) * s
    pm = _sqrt(pm2)
    cr, sr = px / pm, py / pm
    vn0, vn1 = cr * vx + sr * vy, cr * vy - sr * vx
    a = pm2 - ego * ego
    d = _where(_val(a) > 1e-6, a, 1e-6)
    kl, km = 0.1 * math.sqrt(s * s - 1.0) / ego, 0.5 * math.sqrt(s * s - 1.0) / ego
    rd = _sqrt(d)
    return vn0 + (kl * rd / vm) * vn1 * vn1 + km * rd


def kb_state_model(name, spec=None, dt=0.05):
    """KinematicBicycle2D_C3BF / _DPCBF: the bicycle's dynamics, weights and boxes (mpc_cbf.py:31-33,202-208) with ONE-step rows on a barrier of the
    full state, d_h + alpha h_k >= 0 with alpha = 0.15 (mpc_cbf.py:67-72,312-315); x_k+1 = robot.step (speed clip)."""
    m = kb_model(spec, dt)
    m.update(name=name, alpha=0.15, alpha1=0.0, alpha2=0.0, state_barrier=h_c3bf if name.endswith("C3BF") else h_dpcbf)
    return m


def kb_model(spec=None, dt=0.05):
    """mpc_cbf.py:31-33 (Q, R), :64-66 (alpha1 = alpha2 = 0.1), :202-208 (|v| <= v_max, input box), kinematic_bicycle2D.py:175 (beta = 1.1)."""
    s = dict(wheel_base=0.4, radius=0.3, rear_ax_dist=0.2, v_max=3.5, a_max=5.0, v_min=0.2)
    s["beta_max"] = math.atan((0.2 / 0.4) * math.tan(math.radians(32)))
    s.update(spec or {})
    return dict(name="KinematicBicycle2D", nx=4, nu=2, next=kb_next, row_next=kb_step, spec=s, dt=dt, N=10, Q=np.array([50.0, 50.0, 1.0, 1.0]),
                R=np.array([0.5, 5000.0]), alpha1=0.1, alpha2=0.1, beta=1.1, radius=s["radius"], circles_only=True,
                u_lo=np.array([-s["a_max"], -s["beta_max"]]), u_hi=np.array([s["a_max"], s["beta_max"]]),
                x_lo=np.array([-INF, -INF, -INF, -s["v_max"]]), x_hi=np.array([INF, INF, INF, s["v_max"]]))


def di_model(spec=None, dt=0.05):
    """mpc_cbf.py:28-30 (Q, R), :56-59 (alpha1 = alpha2 = 0.2), :196-200 (input box, no state bounds), double_integrator2D.py:222 (beta = 1.01)."""
    s = dict(a_max=1.0, v_max=1.0, radius=0.25)
    s.update(spec or {})
    s.setdefault("ax_max", s["a_max"]); s.setdefault("ay_max", s["a_max"])
    return dict(name="DoubleIntegrator2D", nx=4, nu=2, next=di_next, row_next=di_step, spec=s, dt=dt, N=10, Q=np.array([50.0, 50.0, 20.0, 20.0]),
                R=np.array([0.5, 0.5]), alpha1=0.2, alpha2=0.2, beta=1.01, radius=s["radius"],
                u_lo=np.array([-s["ax_max"], -s["ay_max"]]), u_hi=np.array([s["ax_max"], s["ay_max"]]),
                x_lo=np.full(4, -INF), x_hi=np.full(4, INF))


def vtol_model(spec=None, dt=0.05):
    from . import mpc_vtol as OV
    s = OV.default_spec(**(spec or {}))
    pm = s["pitch_max"] * 3.14159 / 180                        # mpc_cbf.py:232-233
    return dict(name="VTOL2D", nx=6, nu=4, next=vt_next, spec=s, dt=dt, N=30, Q=np.array([10.0, 10.0, 250.0, 10.0, 10.0, 50.0]),
                R=np.array([0.5, 0.5, 0.5, 50000.0]), alpha1=0.05, alpha2=0.05, beta=1.01, radius=s["radius"],
                u_lo=np.array([s["throttle_min"]] * 3 + [s["elevator_min"]]), u_hi=np.array([s["throttle_max"]] * 3 + [s["elevator_max"]]),
                x_lo=np.array([-INF, -INF, -pm, -s["v_max"], -s["descent_speed_max"], -INF]),
                x_hi=np.array([INF, INF, pm, s["v_max"], INF, INF]))


def vtol_od_model(spec=None, dt=0.05):
    """OptimalDecayMPCCBF with a VTOL2D robot (optimal_decay_mpc_cbf.py:19,44-47,83-91,123-124,173-186,288-296) in the multiple-shooting form:
    the two decay rates omega1_k, omega2_k are INPUTS 4 and 5 of a stage (free), the CBF row of the stage reads
    dd_h + (a1 omega1 + a2 omega2) d_h + a1 a2 omega1 omega2 h >= 0 with a1 = a2 = 0.35, cost + p_sb (omega - 1)^2, input term R u^2."""
    m = vtol_model(spec, dt)
    big = np.array([INF, INF])
    m.update(name="VTOL2D_OD", nu=6, nu_dyn=4, alpha1=0.35, alpha2=0.35, rterm="u", od=dict(omega_ref=np.array([1.0, 1.0]), p_sb=np.array([10.0, 10.0])),
             u_lo=np.concatenate([m["u_lo"], -big]), u_hi=np.concatenate([m["u_hi"], big]), R=np.concatenate([m["R"], [0.0, 0.0]]))
    return m


class StageNLP:
    """min f(w)  s.t.  c(w) = 0,  d(w) <= 0,  w_lo <= w <= w_hi   with w = [x_0, u_0, x_1, u_1, .., x_{N-1}, u_{N-1}, x_N]."""

    def __init__(self, model, x0, u_prev, goal, obs, N=None):
        self.mdl = model
        self.N = N = int(N or model["N"])
        self.nx, self.nu = nx, nu = model["nx"], model["nu"]
        self.nv = nx + nu
        self.x0 = np.asarray(x0, dtype=float).reshape(-1)[:nx].copy()
        self.u_prev = np.zeros(nu)
        up_ = np.asarray(u_prev, dtype=float).reshape(-1)[:nu]
        self.u_prev[: up_.shape[0]] = up_
        self.xg = np.zeros(nx)
        self.xg[:2] = np.asarray(goal, dtype=float).reshape(-1)[:2]     # goal padded with zeros (mpc_cbf.py:267)
        self.obs = np.asarray(obs, dtype=float)
        self.circles_only = model["name"] == "VTOL2D" or bool(model.get("circles_only"))     # vtol2D.py:482-490, kinematic_bicycle2D.py:181-189: no superellipsoid branch
        self.K = K = self.obs.shape[0]
        self.n = (N + 1) * nx + N * nu
        self.m_c = (N + 1) * nx
        self.m_d = N * K
        self.ix = np.array([k * self.nv + np.arange(nx) for k in range(N + 1)])          # (N+1, nx) indices of x_k
        self.iu = np.array([k * self.nv + nx + np.arange(nu) for k in range(N)])          # (N, nu)
        self.w_lo = np.full(self.n, -INF)
        self.w_hi = np.full(self.n, INF)
        self.w_lo[self.ix] = model["x_lo"]
        self.w_hi[self.ix] = model["x_hi"]
        self.w_lo[self.iu] = model["u_lo"]
        self.w_hi[self.iu] = model["u_hi"]
        self.d_lo = np.full(self.m_d, -INF)
        self.d_hi = np.zeros(self.m_d)
        g1, g2 = model["alpha1"] + model["alpha2"], model["alpha1"] * model["alpha2"]
        self.cw = (1.0 - g1 + g2, g1 - 2.0, 1.0)                        # weights of h(x), h(x1), h(x2): dd_h + g1 d_h + g2 h
        if model.get("alpha") is not None:                              # one-step rows d_h + alpha h (mpc_cbf.py:312-315): h(x1) - (1 - alpha) h(x)
            self.cw = (float(model["alpha"]) - 1.0, 1.0, 0.0)
        self.n_eval = 0

    def initial_guess(self):
        """set_initial_guess (mpc_cbf.py:366-369): every stage's state at x0, every input at the input applied last."""
        w = np.zeros(self.n)
        w[self.ix] = self.x0
        up = self.u_prev
        if self.mdl.get("od"):                                              # (the decay rates start at their references, as in oracle/od_mpc_cbf.py)
            up = np.concatenate([up[: self.mdl["nu_dyn"]], self.mdl["od"]["omega_ref"]])
        w[self.iu] = up
        return w

    def split(self, w):
        return w[self.ix], w[self.iu]

    def _h(self, px, pz):
        """h_j = |p - p_obs_j|^2 - beta (R + r_j)^2 for every obstacle (agent_barrier_dt: dynamic_unicycle2D.py:194-202, vtol2D.py:482-490)."""
        out = []
        R = self.mdl["radius"]
        for j in range(self.K):
            o = self.obs[j]
            if o[6] < 0.5 or self.circles_only:
                dmin = R + o[2]
                ex, ez = px - o[0], pz - o[1]
                out.append(ex * ex + ez * ez - self.mdl["beta"] * dmin * dmin)
                continue
            # superellipsoid (dynamic_unicycle2D.py:204-220): fabs, clamps a, b >= 1e-3, e >= 2
            a, b, e = max(abs(o[2]), 1e-3) + R, max(abs(o[3]), 1e-3) + R, max(abs(o[4]), 2.0)
            ct, st = math.cos(o[5]), math.sin(o[5])
            qx, qy = ct * (px - o[0]) + st * (pz - o[1]), ct * (pz - o[1]) - st * (px - o[0])
            out.append(_abs(qx) ** e / a ** e + _abs(qy) ** e / b ** e - 1.0)
        return out

    def _rows(self, x, u):
        """d_kj = -(dd_h + (a1 + a2) d_h + a1 a2 h) with x1 = step(x, u), x2 = step(x1, u) (mpc_cbf.py:304,316-321); the heading /
        pitch wrap inside step() touches no position."""
        spec, dt, nxt = self.mdl["spec"], self.mdl["dt"], self.mdl["next"]
        nd = self.mdl.get("nu_dyn", self.nu)
        x1 = nxt(x, u[:nd], spec, dt)
        if self.mdl.get("state_barrier"):                                   # one-step rows on a barrier of the whole state (the collision-cone bicycles)
            hf, a_ = self.mdl["state_barrier"], float(self.mdl["alpha"])
            S = self.mdl["row_next"](x, u[:nd], spec, dt)
            R = self.mdl["radius"]
            return x1, [-(hf(S, self.obs[j], R) - (1.0 - a_) * hf(x, self.obs[j], R)) for j in range(self.K)]
        if self.cw[2] == 0.0:                                               # one-step rows (robot.step = x_next but for the heading wrap, which touches no position)
            h0, h1 = self._h(x[0], x[1]), self._h(x1[0], x1[1])
            return x1, [-(self.cw[1] * h1[j] + self.cw[0] * h0[j]) for j in range(self.K)]
        if self.mdl.get("row_next"):                                        # robot.step differs from the model's x_next (the bicycle's speed clip)
            step = self.mdl["row_next"]
            x1r = step(x, u[:nd], spec, dt)
            x2 = step(x1r, u[:nd], spec, dt)
            h0, h1, h2 = self._h(x[0], x[1]), self._h(x1r[0], x1r[1]), self._h(x2[0], x2[1])
            w0, w1, w2 = self.cw
            return x1, [-(w2 * h2[j] + w1 * h1[j] + w0 * h0[j]) for j in range(self.K)]
        x2 = nxt(x1, u[:nd], spec, dt)
        h0, h1, h2 = self._h(x[0], x[1]), self._h(x1[0], x1[1]), self._h(x2[0], x2[1])
        w0, w1, w2 = self.cw
        if self.mdl.get("od"):                                              # optimal decay: the gains of the stage's rows scale with its decay rates
            a1, a2 = self.mdl["alpha1"], self.mdl["alpha2"]
            sk, qk = a1 * u[nd] + a2 * u[nd + 1], (a1 * a2) * (u[nd] * u[nd + 1])
            w0, w1 = 1.0 - sk + qk, sk - 2.0
        return x1, [-(w2 * h2[j] + w1 * h1[j] + w0 * h0[j]) for j in range(self.K)]

    def evaluate(self, w, level=2):
        """level 0: f, c, d.   level 2: + grad f, J_c, J_d and `hess(sigma_f, y_c, y_d)`."""
        self.n_eval += 1
        N, nx, nu, nv, K = self.N, self.nx, self.nu, self.nv, self.K
        X, U = self.split(w)
        Q, R = self.mdl["Q"], self.mdl["R"]
        e = X - self.xg
        up = np.vstack([self.u_prev[None, :], U])
        r_on_u = self.mdl.get("rterm") == "u"                              # optimal decay: R u^2 (an expression r-term), not the delta-u penalty
        du = U.copy() if r_on_u else up[1:] - up[:-1]
        f = float(np.sum(Q * e * e) + np.sum(R * du * du))
        od = self.mdl.get("od")
        if od:
            dom = U[:, self.mdl["nu_dyn"]:] - od["omega_ref"]
            f += float(np.sum(od["p_sb"] * dom * dom))
        out = dict(f=f)
        if level == 0:
            xs = [X[:N, i] for i in range(nx)]
            us = [U[:, i] for i in range(nu)]
            x1, rows = self._rows(xs, us)
            c = np.concatenate([X[0] - self.x0, (np.stack(x1, axis=1) - X[1:]).reshape(-1)])
            out.update(c=c, d=np.stack(rows, axis=1).reshape(-1))
            return out
        V = VD2.seed(np.hstack([X[:N], U]))
        x1, rows = self._rows(V[:nx], V[nx:])
        c = np.concatenate([X[0] - self.x0, (np.stack([a.v for a in x1], axis=1) - X[1:]).reshape(-1)])
        d = np.stack([r.v for r in rows], axis=1).reshape(-1)
        grad = np.zeros(self.n)
        grad[self.ix] = 2.0 * Q * e
        gu = 2.0 * R * du
        grad[self.iu] += gu
        if not r_on_u:
            grad[self.iu[:-1]] -= gu[1:]
        if od:
            grad[self.iu[:, self.mdl["nu_dyn"]:]] += 2.0 * od["p_sb"] * dom
        Jc = np.zeros((self.m_c, self.n))
        Jc[np.arange(nx), np.arange(nx)] = 1.0
        Jd = np.zeros((self.m_d, self.n))
        F1 = np.stack([a.d for a in x1], axis=1)                          # (N, nx, nv)
        D1 = np.stack([r.d for r in rows], axis=1)                        # (N, K, nv)
        for k in range(N):
            r0 = nx + k * nx
            Jc[r0:r0 + nx, k * nv:(k + 1) * nv] = F1[k]
            Jc[np.arange(r0, r0 + nx), (k + 1) * nv + np.arange(nx)] = -1.0
            Jd[k * K:(k + 1) * K, k * nv:(k + 1) * nv] = D1[k]
        FH = np.stack([a.H for a in x1], axis=1)                          # (N, nx, nv, nv)
        DH = np.stack([r.H for r in rows], axis=1)                        # (N, K, nv, nv)
        Hf = np.zeros((self.n, self.n))
        Hf[self.ix, self.ix] = 2.0 * Q
        for k in range(N):
            iu = self.iu[k]
            Hf[iu, iu] += 2.0 * R
            if od:
                io = iu[self.mdl["nu_dyn"]:]
                Hf[io, io] += 2.0 * od["p_sb"]
            if k + 1 < N and not r_on_u:
                Hf[iu, iu] += 2.0 * R
                Hf[iu, self.iu[k + 1]] -= 2.0 * R
                Hf[self.iu[k + 1], iu] -= 2.0 * R

        def hess(sigma_f, y_c, y_d):
            W = sigma_f * Hf
            yc = y_c[nx:].reshape(N, nx)
            yd = y_d.reshape(N, K)
            blk = np.einsum("ki,kiab->kab", yc, FH) + np.einsum("kj,kjab->kab", yd, DH)
            for k in range(N):
                W[k * nv:(k + 1) * nv, k * nv:(k + 1) * nv] += blk[k]
            return W

        out.update(c=c, d=d, grad=grad, Jc=Jc, Jd=Jd, hess=hess)
        return out

    def x_bounds(self):
        return self.w_lo, self.w_hi

    def u0(self, w):
        return w[self.iu[0]].copy()


# ---- the interior point -----------------------------------------------------------------------------------------------
OPTS = dict(tol=1e-8, max_iter=3000, dual_inf_tol=1.0, constr_viol_tol=1e-4, compl_inf_tol=1e-4,
            acceptable_tol=1e-6, acceptable_iter=15, acceptable_dual_inf_tol=1e10, acceptable_constr_viol_tol=1e-2, acceptable_compl_inf_tol=1e-2,
            nlp_scaling_max_gradient=100.0, nlp_scaling_min_value=1e-8, bound_relax_factor=1e-8,
            bound_push=1e-2, bound_frac=1e-2, bound_mult_init_val=1.0, constr_mult_init_max=1e3,
            mu_init=0.1, mu_linear_decrease_factor=0.2, mu_superlinear_decrease_power=1.5, barrier_tol_factor=10.0, tau_min=0.99,
            kappa_sigma=1e10, kappa_d=1e-5, s_max=100.0,
            theta_max_fact=1e4, theta_min_fact=1e-4, eta_phi=1e-8, delta=1.0, s_phi=2.3, s_theta=1.1, gamma_phi=1e-8, gamma_theta=1e-5,
            alpha_min_frac=0.05, alpha_red_factor=0.5, max_soc=4, kappa_soc=0.99, obj_max_inc=5.0,
            first_hessian_perturbation=1e-4, min_hessian_perturbation=1e-20, max_hessian_perturbation=1e20,
            perturb_inc_fact_first=100.0, perturb_inc_fact=8.0, perturb_dec_fact=1.0 / 3.0, jacobian_regularization_value=1e-8,
            jacobian_regularization_exponent=0.25,
            resto_penalty_parameter=1000.0, resto_proximity_weight=1.0, required_infeasibility_reduction=0.9,
            bound_mult_reset_threshold=1e3, constr_mult_reset_threshold=0.0, resto_failure_feasibility_threshold=1e-6,
            resto_theta_max_fact=1e8,
            # NOT IPOPT options (IPOPT's watchdog / tiny-step heuristics are not restated): stop a solve -- or its restoration phase -- whose accepted
            # step length stays below stall_alpha for stall_iter consecutive iterations (0: never) -- csrc/mpc_vtol_ms.hip runs with (60, 1e-4)
            stall_iter=0, stall_alpha=1e-4,
            # IPOPT's recalc_y as a rescue (an experiment of round 5, DESIGN.md (f) 2b'; the kernel does not run it): once a regular-phase solve has sat recalc_y_iter iterations at a barrier
            # parameter within 10 x its floor with everything but the dual infeasibility inside the 'acceptable' tolerances, the row multipliers are
            # re-estimated by least squares at every iterate from then on (0: never)
            recalc_y_iter=0,
            # NOT an IPOPT option either (round 6; csrc/mpc_vtol_ms.hip runs with 30): a regular-phase solve that has sat floor_iter consecutive iterations
            # at the smallest barrier parameter with everything but the dual infeasibility inside the 'acceptable' tolerances stops there (same class as
            # running out of iterations: status 2) -- the iterate is the optimum, what does not come down is the precision floor of the multiplier
            # recovery (DESIGN.md kernel 12), and IPOPT itself would spend the rest of its 3000 iterations there for the same input (0: never)
            floor_iter=0,
            filter_cap=0)                   # (experiments: emulate a kernel's fixed-size filters; 0 = unbounded, as IPOPT's)

EPS = np.finfo(float).eps


def compare_le(lhs, rhs, basval):
    """IPOPT's Compare_le: lhs <= rhs up to 10 eps |basval|."""
    return lhs - rhs <= 10.0 * EPS * abs(basval)


class _Problem:
    """min f(x) + rho_t' t   s.t.  g(x) + A_t t = 0,  x_L <= x <= x_U,  t_L <= t <= t_U;  every column of A_t is +-e_row.
    Regular phase: g = [c; d] scaled, t = the slacks of the d rows.  Restoration: + n, p >= 0 on every row."""


class _Regular(_Problem):
    def __init__(self, nlp, w_start, o):
        self.nlp = nlp
        self.stage = nlp if isinstance(nlp, StageNLP) else None
        self.n, self.m_c, self.m_d = nlp.n, nlp.m_c, nlp.m_d
        self.m = self.m_c + self.m_d
        ev = nlp.evaluate(w_start, 2)
        gmax, gmin = o["nlp_scaling_max_gradient"], o["nlp_scaling_min_value"]
        a = float(np.max(np.abs(ev["grad"]))) if self.n else 0.0
        self.df = max(gmin, gmax / a) if a > gmax else 1.0
        J = np.vstack([ev["Jc"], ev["Jd"]])
        ra = np.max(np.abs(J), axis=1) if self.m else np.zeros(0)
        self.dg = np.where(ra > gmax, np.maximum(gmin, gmax / np.maximum(ra, 1e-300)), 1.0)
        rl = o["bound_relax_factor"]
        lo, hi = nlp.x_bounds()
        self.x_L = np.where(np.isfinite(lo), lo - rl * np.maximum(1.0, np.abs(lo)), -INF)
        self.x_U = np.where(np.isfinite(hi), hi + rl * np.maximum(1.0, np.abs(hi)), INF)
        dl, dh = nlp.d_lo * self.dg[self.m_c:], nlp.d_hi * self.dg[self.m_c:]
        self.nt = self.m_d
        self.t_L = np.where(np.isfinite(dl), dl - rl * np.maximum(1.0, np.abs(dl)), -INF)
        self.t_U = np.where(np.isfinite(dh), dh + rl * np.maximum(1.0, np.abs(dh)), INF)
        self.t_row = self.m_c + np.arange(self.m_d)
        self.t_sig = -np.ones(self.m_d)
        self.t_rho = np.zeros(self.m_d)

    def evaluate(self, x, level, mu=None):
        ev = self.nlp.evaluate(x, level)
        out = dict(f=self.df * ev["f"], g=self.dg * np.concatenate([ev["c"], ev["d"]]))
        if level >= 2:
            out["grad"] = self.df * ev["grad"]
            out["J"] = self.dg[:, None] * np.vstack([ev["Jc"], ev["Jd"]])
            h, df, dg, mc = ev["hess"], self.df, self.dg, self.m_c
            out["hess"] = lambda y: h(df, dg[:mc] * y[:mc], dg[mc:] * y[mc:])
        return out


class _Resto(_Problem):
    """Section 3.3:  min rho sum(n + p) + zeta/2 |D_R (x - x_R)|^2   s.t.  g(x) + A_s s + n - p = 0,  n, p >= 0,  zeta = sqrt(mu)."""

    def __init__(self, reg, x_R, o):
        self.reg = reg
        self.stage = reg.stage
        self.n, self.m = reg.n, reg.m
        self.x_L, self.x_U = reg.x_L.copy(), reg.x_U.copy()
        m, ms = reg.m, reg.nt
        # rows that get n - p: all of them (IPOPT), or the inequality rows only (o["resto_elastic"] = "ineq": the dynamics rows stay
        # hard equalities inside the restoration -- the variant csrc/mpc_vtol_ms.hip runs, DESIGN.md kernel 12)
        self.el = np.arange(m) if o.get("resto_elastic", "all") == "all" else np.arange(reg.m_c, m)
        ne = len(self.el)
        self.ns = ms
        self.nt = ms + 2 * ne
        self.t_L = np.concatenate([reg.t_L, np.zeros(2 * ne)])
        self.t_U = np.concatenate([reg.t_U, np.full(2 * ne, INF)])
        self.t_row = np.concatenate([reg.t_row, self.el, self.el])
        self.t_sig = np.concatenate([reg.t_sig, np.ones(ne), -np.ones(ne)])
        self.rho = o["resto_penalty_parameter"]
        self.t_rho = np.concatenate([np.zeros(ms), np.full(2 * ne, self.rho)])
        self.x_R = x_R.copy()
        self.DR2 = (1.0 / np.maximum(1.0, np.abs(x_R))) ** 2
        self.eta = o["resto_proximity_weight"]

    def evaluate(self, x, level, mu=None):
        ev = self.reg.evaluate(x, level)
        zeta = self.eta * math.sqrt(mu)
        dx = x - self.x_R
        out = dict(f=0.5 * zeta * float(np.sum(self.DR2 * dx * dx)), g=ev["g"], f_orig=ev["f"])
        if level >= 2:
            out["grad"] = zeta * self.DR2 * dx
            out["J"] = ev["J"]
            DR2, hreg = self.DR2, ev["hess"]
            W0 = hreg(np.zeros(self.m))                                   # = scaled Hessian of the original objective, which does not enter:

            def hess(y):                                                    # zeta D_R^2 + sum_i y_i grad^2 g_i
                return hreg(y) - W0 + np.diag(zeta * DR2)
            out["hess"] = hess
        return out


# what csrc/mpc_vtol_ms.hip runs (DESIGN.md kernel 12): stage-wise Riccati linear algebra, no second-order corrections, restoration phase with
# elastic variables on the inequality rows only, the stall rule, the precision-floor stop.  KERNEL_PROFILE_NO_RESTO: the same without a restoration phase (status 4).
KERNEL_PROFILE = dict(linear_solver="riccati", max_soc=0, resto_elastic="ineq", stall_iter=60, stall_alpha=1e-4, floor_iter=30)
KERNEL_PROFILE_NO_RESTO = dict(linear_solver="riccati", max_soc=0, restoration="none", stall_iter=60, stall_alpha=1e-4, floor_iter=30)


def _dist(v, lo, hi):
    return v - lo, hi - v                                                  # +inf where there is no bound


def _ftb(tau, slack, dslack):
    """largest alpha in (0, 1] with slack + alpha dslack >= (1 - tau) slack."""
    neg = dslack < 0
    if not np.any(neg):
        return 1.0
    return min(1.0, float(np.min(-tau * slack[neg] / dslack[neg])))


class _Filter:
    def __init__(self, cap=0):
        self.e = []
        self.cap = cap          # > 0: a kernel's fixed-size filter -- a full filter overwrites its last entry (option filter_cap; experiments only)
        self.peak = 0

    def acceptable(self, phi, theta):
        for (p, t) in self.e:
            if not (compare_le(phi, p, p) or compare_le(theta, t, t)):
                return False
        return True

    def add(self, phi, theta):
        self.e = [(p, t) for (p, t) in self.e if not (p >= phi and t >= theta)]
        if self.cap > 0 and len(self.e) >= self.cap:
            self.e = self.e[:self.cap - 1]
        self.e.append((phi, theta))
        self.peak = max(self.peak, len(self.e))

    def clear(self):
        self.e = []


class _Algo:
    """One run of the algorithm on a _Problem; the restoration phase is another _Algo on the _Resto problem."""

    def __init__(self, prob, o, resto_of=None, trace=None):
        self.P, self.o, self.outer, self.trace = prob, o, resto_of, trace
        self.in_resto = resto_of is not None
        self.filter = _Filter(int(o.get("filter_cap", 0) or 0))
        self.delta_w_last = 0.0
        self.iters = 0

    # -- quantities of an iterate --------------------------------------------------------------------------------------
    def slacks(self, x, t):
        P = self.P
        return _dist(x, P.x_L, P.x_U) + _dist(t, P.t_L, P.t_U)

    def safe_slacks(self, x, t, mu):
        """IPOPT's CalculateSafeSlack: a slack below eps min(1, mu) is raised to slack_move max(1, |bound|), slack_move = eps^(3/4), by
        moving the bound (for good) -- "slack too small, adjusting variable bound"."""
        P = self.P
        s_min, move = EPS * min(1.0, mu), EPS ** 0.75
        for v, lo, hi in ((x, P.x_L, P.x_U), (t, P.t_L, P.t_U)):
            bad = np.isfinite(lo) & (v - lo < s_min)
            if np.any(bad):
                lo[bad] = v[bad] - np.maximum(v[bad] - lo[bad], move * np.maximum(1.0, np.abs(lo[bad])))
            bad = np.isfinite(hi) & (hi - v < s_min)
            if np.any(bad):
                hi[bad] = v[bad] + np.maximum(hi[bad] - v[bad], move * np.maximum(1.0, np.abs(hi[bad])))

    def barrier(self, f, x, t, mu):
        P, kd = self.P, self.o["kappa_d"]
        sxL, sxU, stL, stU = self.slacks(x, t)
        if min(np.min(sxL, initial=1.0), np.min(sxU, initial=1.0), np.min(stL, initial=1.0), np.min(stU, initial=1.0)) <= 0.0:
            return INF
        phi = f + float(P.t_rho @ t)
        for s_, o_ in ((sxL, sxU), (sxU, sxL), (stL, stU), (stU, stL)):
            fin = np.isfinite(s_)
            phi -= mu * float(np.sum(np.log(s_[fin])))
            one = fin & ~np.isfinite(o_)
            phi += kd * mu * float(np.sum(s_[one]))
        return phi

    def barrier_grad(self, grad, x, t, mu):
        P, kd = self.P, self.o["kappa_d"]
        sxL, sxU, stL, stU = self.slacks(x, t)
        gx = grad - mu / sxL + mu / sxU + kd * mu * ((np.isfinite(sxL) & ~np.isfinite(sxU)).astype(float) - (np.isfinite(sxU) & ~np.isfinite(sxL)).astype(float))
        gt = P.t_rho - mu / stL + mu / stU + kd * mu * ((np.isfinite(stL) & ~np.isfinite(stU)).astype(float) - (np.isfinite(stU) & ~np.isfinite(stL)).astype(float))
        return gx, gt

    def residual(self, g, t):
        P = self.P
        r = g.copy()
        np.add.at(r, P.t_row, P.t_sig * t)
        return r

    def errors(self, ev, x, t, y, z, mu):
        """E_mu (scaled, eq. (5)) and its parts."""
        P, smax = self.P, self.o["s_max"]
        zxL, zxU, ztL, ztU = z
        sxL, sxU, stL, stU = self.slacks(x, t)
        dual_x = ev["grad"] + ev["J"].T @ y - zxL + zxU
        dual_t = P.t_rho + P.t_sig * y[P.t_row] - ztL + ztU
        r = self.residual(ev["g"], t)
        comp = 0.0
        for s_, z_ in ((sxL, zxL), (sxU, zxU), (stL, ztL), (stU, ztU)):
            fin = np.isfinite(s_)
            if np.any(fin):
                comp = max(comp, float(np.max(np.abs(s_[fin] * z_[fin] - mu))))
        nb = sum(int(np.sum(np.isfinite(s_))) for s_ in (sxL, sxU, stL, stU))
        zsum = float(sum(np.sum(np.abs(z_)) for z_ in z))
        sd = max(smax, (float(np.sum(np.abs(y))) + zsum) / max(1, P.m + nb)) / smax
        sc = max(smax, zsum / max(1, nb)) / smax
        dinf = max(float(np.max(np.abs(dual_x))), float(np.max(np.abs(dual_t), initial=0.0)))
        pinf = float(np.max(np.abs(r), initial=0.0))
        return max(dinf / sd, pinf, comp / sc), dinf, pinf, comp, r

    # -- the primal-dual system ----------------------------------------------------------------------------------------
    def factor(self, W, J, sig_x, sig_t, mu):
        """Algorithm IC around the reduced system; returns a solver closure or None (-> restoration)."""
        P, o = self.P, self.o
        if o.get("linear_solver") == "riccati" and P.stage is not None:
            return self.factor_riccati(W, J, sig_x, sig_t, mu)
        n, m = P.n, P.m
        dw, dc = 0.0, 0.0
        K = np.zeros((n + m, n + m))
        stage = 0                                                          # 0: unperturbed, 1: delta_c only (singular), 2: delta_w > 0
        while True:
            q = 1.0 / (sig_t + dw)
            e = np.full(m, dc)
            np.add.at(e, P.t_row, q)
            K[:n, :n] = W
            K[np.arange(n), np.arange(n)] += sig_x + dw
            K[:n, n:] = J.T
            K[n:, :n] = J
            K[n:, n:] = 0.0
            K[n + np.arange(m), n + np.arange(m)] = -e
            ldu, piv, info = lapack.dsytrf(K, lower=1)
            singular = info > 0
            if not singular and self._n_negative(ldu, piv) == m:
                if dw > 0.0:
                    self.delta_w_last = dw
                self.last_delta = (dw, dc)
                break
            if stage == 0 and singular:
                stage, dc = 1, o["jacobian_regularization_value"] * mu ** o["jacobian_regularization_exponent"]
                continue
            if stage < 2:
                stage = 2
                dw = o["first_hessian_perturbation"] if self.delta_w_last == 0.0 else max(o["min_hessian_perturbation"], o["perturb_dec_fact"] * self.delta_w_last)
            else:
                if singular and dc == 0.0:
                    dc = o["jacobian_regularization_value"] * mu ** o["jacobian_regularization_exponent"]
                dw = dw * (o["perturb_inc_fact_first"] if self.delta_w_last == 0.0 else o["perturb_inc_fact"])
            if dw > o["max_hessian_perturbation"]:
                return None

        def solve(rhs_x, rhs_t, rhs_g):
            """H dx + J'dy = rhs_x;  (sig_t + dw) dt + sigma dy_row = rhs_t;  J dx + A_t dt - dc dy = rhs_g."""
            b = np.concatenate([rhs_x, rhs_g])
            np.subtract.at(b, n + P.t_row, P.t_sig * q * rhs_t)
            sol, info2 = lapack.dsytrs(ldu, piv, b, lower=1)
            res = b - self._kmul(W, J, sig_x + dw, e, sol)                  # one step of iterative refinement
            cor, _ = lapack.dsytrs(ldu, piv, res, lower=1)
            sol = sol + cor
            dx, dy = sol[:n], sol[n:]
            dt = q * (rhs_t - P.t_sig * dy[P.t_row])
            return dx, dt, dy
        return solve


    # -- the same system by a Riccati recursion over the stages: the linear algebra of the HIP kernel ------------------------
    def factor_riccati(self, W, J, sig_x, sig_t, mu):
        """The primal-dual system of a StageNLP solved stage by stage (csrc/mpc_vtol_ms.hip does exactly this):
          * the inequality rows of stage k are condensed into its Hessian block, H_k += Jd_k' diag(1 / e) Jd_k (e > 0 on every row that
            carries a slack);
          * the input-rate term couples u_k with u_{k+1}: the recursion runs on the augmented state xi_k = (x_k, v_k), v_k = u_{k-1},
            with xi_{k+1} = (A_k x_k + B_k u_k + c_k, u_k); value function 1/2 xi' P_k xi + p_k' xi;
          * regular phase: hard dynamics.  Restoration: every dynamics row carries n - p, which makes the row SOFT -- dx+ = a + c - E
            dy+ with E = diag(1 / (Sigma_n + dw) + 1 / (Sigma_p + dw)): the recursion then uses the parallel sum P~ = P - P E^1/2 (I +
            E^1/2 P E^1/2)^-1 E^1/2 P in place of P (section "restoration" of DESIGN.md kernel 12);
          * inertia: the full system has (n, m, 0) iff every 4 x 4 input block Quu_k and every I + E^1/2 P_k E^1/2 is positive
            definite (block elimination in stage order + Sylvester's law); on failure delta_w grows as in Algorithm IC (delta_c is
            never needed: the dynamics rows have full rank by construction)."""
        P, o = self.P, self.o
        S = P.stage
        N, nx, nu, nv, K = S.N, S.nx, S.nu, S.nv, S.K
        n, m, mc = P.n, P.m, (S.N + 1) * S.nx
        na = nx + nu
        Wd = [W[k * nv:(k + 1) * nv, k * nv:(k + 1) * nv] for k in range(N)]
        WN = W[N * nv:, N * nv:]
        cpl = [np.array([-W[S.iu[k][i], S.iu[k + 1][i]] for i in range(nu)]) for k in range(N - 1)]
        # the dynamics rows carry their scaling factor: row = dg (F - x+); the recursion runs on the unscaled rows (costate = dg dy)
        sc = np.ones(mc)
        for k in range(N):
            sc[nx + k * nx: nx + (k + 1) * nx] = -np.diagonal(J[nx + k * nx: nx + (k + 1) * nx, (k + 1) * nv:(k + 1) * nv + nx])
        Jc = [J[nx + k * nx: nx + (k + 1) * nx, k * nv:(k + 1) * nv] / sc[nx + k * nx: nx + (k + 1) * nx, None] for k in range(N)]       # [A_k B_k]
        Jd = [J[mc + k * K: mc + (k + 1) * K, k * nv:(k + 1) * nv] for k in range(N)]
        # optimal-decay stages: the decay rates (the last nr inputs, absent from the dynamics) leave the stage before the recursion
        nr = (nu - S.mdl["nu_dyn"]) if (S.mdl.get("od") is not None and o.get("od_elimination", "sequential") == "sequential") else 0
        if nr:
            from types import SimpleNamespace
            Sr = SimpleNamespace(N=N, nx=nx, nu=nu - nr, nv=nv - nr, K=K, n=n - N * nr)
            Jcr = [a[:, :nv - nr] for a in Jc]
            Jdr = [a[:, :nv - nr] for a in Jd]
            cplr = [c[:nu - nr] for c in cpl]
        dw = 0.0
        first = True
        while True:
            q = 1.0 / (sig_t + dw)
            e = np.zeros(m)
            np.add.at(e, P.t_row, q)
            e_c, e_d = (e[:mc] / (sc * sc)).reshape(N + 1, nx), e[mc:].reshape(N, K)
            soft = bool(np.any(e_c > 0.0))
            if nr:
                red = self._od_eliminate(S, Wd, Jd, sig_x, dw, e_d, nr)
                fac = None if red is None else self._riccati_backward(Sr, None, WN, cplr, Jcr, None, sig_x[N * nv:], dw, e_c, e_d, soft, Hs=red["R"])
            else:
                fac = self._riccati_backward(S, Wd, WN, cpl, Jc, Jd, sig_x, dw, e_c, e_d, soft)
            if fac is not None:
                if dw > 0.0:
                    self.delta_w_last = dw
                self.last_delta = (dw, 0.0)
                break
            if first:
                first = False
                dw = o["first_hessian_perturbation"] if self.delta_w_last == 0.0 else max(o["min_hessian_perturbation"], o["perturb_dec_fact"] * self.delta_w_last)
            else:
                dw = dw * (o["perturb_inc_fact_first"] if self.delta_w_last == 0.0 else o["perturb_inc_fact"])
            if dw > o["max_hessian_perturbation"]:
                return None

        def solve(rhs_x, rhs_t, rhs_g):
            b = rhs_g.copy()
            np.subtract.at(b, P.t_row, P.t_sig * q * rhs_t)
            b[:mc] /= sc
            if nr:
                nvr = nv - nr
                b_d = b[mc:].reshape(N, K)
                gs, ts = self._od_reduce_rhs(red, rhs_x, b_d, N, nv, nr)
                dxr, dy = self._riccati_solve(Sr, fac, Jcr, Jdr, e_c, e_d, soft, np.concatenate([np.zeros(N * nvr), rhs_x[N * nv:]]), b, gs=gs)
                dx = np.zeros(n)
                dx[N * nv:] = dxr[N * nvr:]
                for k in range(N):
                    v = dxr[k * nvr:(k + 1) * nvr]
                    dr = -(ts[k] + red["T"][k].T @ v)                      # d rho = -D^-1 (g_rho + M_rho,v d(x, u))
                    dx[k * nv:k * nv + nvr] = v
                    dx[k * nv + nvr:(k + 1) * nv] = dr
                    dy[mc + k * K: mc + (k + 1) * K] += (Jd[k][:, nvr:] @ dr) / e_d[k]
            else:
                dx, dy = self._riccati_solve(S, fac, Jc, Jd, e_c, e_d, soft, rhs_x, b)
            dy[:mc] /= sc
            dt = q * (rhs_t - P.t_sig * dy[P.t_row])
            return dx, dt, dy
        return solve

    @staticmethod
    def _od_eliminate(S, Wd, Jd, sig_x, dw, e_d, nr):
        """Schur complement of the nr trailing stage variables (decay rates) in H_k = W_k + Sigma + dw + Jd_k' E Jd_k, the way
        csrc/mpc_vtol_ms.hip takes it (eval2, OD): the block without the rows is eliminated first, every row then enters as a rank-one
        update of the complement,
            w = D^-1 b, r = c - T b, q = 1 / E + b . w:   H/rho += r r' / q,  T += r w' / q,  D^-1 -= w w' / q        (T = M_v,rho D^-1)
        with (c, b) the row's gradient in (x, u | rho).  An active row at mu ~ 1e-9 has E b^2 ~ 1e13 against 2 df p_sb ~ 1e-2: the
        assembled block keeps nothing of the small part (tools/micro/seq_schur.py).  Inertia: every q < 0 turns one negative eigenvalue of
        D positive (det (D + E b b') = det D (1 + E b' D^-1 b); a positive semidefinite update lowers no eigenvalue); D not positive
        definite at the end -> None (Algorithm IC raises delta_w)."""
        N, nv, K = S.N, S.nv, S.K
        nvr = nv - nr
        assert nr == 2
        Rs, Ts, D0, T0, steps = [], [], [], [], []
        for k in range(N):
            H0 = Wd[k] + np.diag(sig_x[k * nv:(k + 1) * nv] + dw)
            s11, s12, s22 = H0[nvr, nvr], H0[nvr, nvr + 1], H0[nvr + 1, nvr + 1]
            det = s11 * s22 - s12 * s12
            if not abs(det) > 0.0:
                return None
            nneg = (0 if s11 > 0.0 else 2) if det > 0.0 else 1
            Di = np.array([[s22, -s12], [-s12, s11]]) / det
            Mr = H0[nvr:, :nvr]
            T = Mr.T @ Di
            R = H0[:nvr, :nvr] - T @ Mr
            D0.append(Di.copy()); T0.append(T.copy())
            st = []
            for j in range(K):
                c, b = Jd[k][j, :nvr], Jd[k][j, nvr:]
                w = Di @ b
                q = e_d[k][j] + b @ w                                      # e_d = 1 / E
                if q < 0.0:
                    nneg -= 1
                if not abs(q) > 0.0:
                    return None
                r = c - T @ b
                st.append((w, q, r, b))
                R = R + np.outer(r, r) / q
                T = T + np.outer(r, w) / q
                Di = Di - np.outer(w, w) / q
            if nneg != 0:
                return None
            Rs.append(0.5 * (R + R.T)); Ts.append(T); steps.append(st)
        return dict(R=Rs, T=Ts, D0=D0, T0=T0, steps=steps)

    @staticmethod
    def _od_reduce_rhs(red, rhs_x, b_d, N, nv, nr):
        """gradient side of _od_eliminate: g_(x,u) and t = D^-1 g_rho after the rows  (sigma = (gamma / E - b . t) / q:  g += sigma r,  t += sigma w;
        the row's share of the gradient is gamma (c, b) with gamma / E = -b_d)."""
        nvr = nv - nr
        gs, ts = [], []
        for k in range(N):
            g0 = -rhs_x[k * nv:(k + 1) * nv]
            t = red["D0"][k] @ g0[nvr:]
            g = g0[:nvr] - red["T0"][k] @ g0[nvr:]
            for j, (w, q, r, b) in enumerate(red["steps"][k]):
                sig = (-b_d[k][j] - b @ t) / q
                g = g + sig * r
                t = t + sig * w
            gs.append(g); ts.append(t)
        return gs, ts

    @staticmethod
    def _riccati_backward(S, Wd, WN, cpl, Jc, Jd, sig_x, dw, e_c, e_d, soft, Hs=None):
        N, nx, nu, nv, K = S.N, S.nx, S.nu, S.nv, S.K
        if Hs is not None:                                                 # stage Hessians given (rows and regularisation inside): sig_x = the terminal block's
            sig_x = np.concatenate([np.zeros(N * nv), sig_x])
        Hs_in = Hs
        na = nx + nu
        Pm = np.zeros((N + 1, na, na))                                     # value functions (unmodified)
        Pt = np.zeros((N + 1, na, na))                                     # parallel sums (soft rows) = what the stage before sees
        Lq, Kg, Lm, Hs = [None] * N, [None] * N, [None] * (N + 1), [None] * N
        Pm[N][:nx, :nx] = WN + np.diag(sig_x[N * nv:] + dw)
        for k in range(N, -1, -1):
            if k < N:
                H = Hs_in[k] if Hs_in is not None else Wd[k] + np.diag(sig_x[k * nv:(k + 1) * nv] + dw) + Jd[k].T @ (Jd[k] / e_d[k][:, None])
                Hs[k] = H
                G = np.zeros((na, na + nu))                                # xi+ = G (xi; u):  [[A, 0, B], [0, 0, I]]
                G[:nx, :nx] = Jc[k][:, :nx]
                G[:nx, na:] = Jc[k][:, nx:]
                G[nx:, na:] = np.eye(nu)
                Sk = np.zeros((na + nu, na + nu))
                Sk[:nx, :nx] = H[:nx, :nx]; Sk[:nx, na:] = H[:nx, nx:]; Sk[na:, :nx] = H[nx:, :nx]; Sk[na:, na:] = H[nx:, nx:]
                if k >= 1:                                                  # -2 R (u_k, v_k) cross term (the diagonal 2 R sits on u_{k-1})
                    c = cpl[k - 1]
                    Sk[nx + np.arange(nu), na + np.arange(nu)] -= c
                    Sk[na + np.arange(nu), nx + np.arange(nu)] -= c
                Q = Sk + G.T @ Pt[k + 1] @ G
                Quu = Q[na:, na:]
                try:
                    L = np.linalg.cholesky(Quu)
                except np.linalg.LinAlgError:
                    return None
                if not np.all(np.isfinite(L)):
                    return None
                Kk = -np.linalg.solve(L.T, np.linalg.solve(L, Q[na:, :na]))
                Pk = Q[:na, :na] + Q[:na, na:] @ Kk
                Pm[k] = 0.5 * (Pk + Pk.T)
                Lq[k], Kg[k] = L, Kk
            # what stage k - 1 (or the initial condition) sees of stage k: soft rows -> parallel sum
            if soft:
                Eh = np.zeros(na)
                Eh[:nx] = np.sqrt(e_c[k])
                M = np.eye(na) + Eh[:, None] * Pm[k] * Eh[None, :]
                try:
                    Lk = np.linalg.cholesky(M)
                except np.linalg.LinAlgError:
                    return None
                Y = np.linalg.solve(Lk, Eh[:, None] * Pm[k])               # L^-1 E^1/2 P
                Pt[k] = Pm[k] - Y.T @ Y
                Pt[k] = 0.5 * (Pt[k] + Pt[k].T)
                Lm[k] = Lk
            else:
                Pt[k] = Pm[k]
        return dict(Pm=Pm, Pt=Pt, Lq=Lq, Kg=Kg, Hs=Hs)

    @staticmethod
    def _riccati_solve(S, fac, Jc, Jd, e_c, e_d, soft, rhs_x, b, gs=None):
        """H dx + J' dy = rhs_x,  J dx - e dy = b  for one right-hand side with the factors of _riccati_backward."""
        N, nx, nu, nv, K = S.N, S.nx, S.nu, S.nv, S.K
        na, mc = nx + nu, (S.N + 1) * S.nx
        Pm, Pt, Lq, Kg = fac["Pm"], fac["Pt"], fac["Lq"], fac["Kg"]
        b_c, b_d = b[:mc].reshape(N + 1, nx), b[mc:].reshape(N, K)
        pm = np.zeros((N + 1, na))                                         # linear terms of the value functions, and as the stage before sees them
        pt = np.zeros((N + 1, na))
        kff = [None] * N
        pm[N][:nx] = -rhs_x[N * nv:]
        E = np.zeros((N + 1, na))
        E[:, :nx] = e_c
        for k in range(N, -1, -1):
            if k < N:
                g = gs[k] if gs is not None else -rhs_x[k * nv:(k + 1) * nv] - Jd[k].T @ (b_d[k] / e_d[k])
                cb = np.zeros(na)
                cb[:nx] = -b_c[k + 1]                                       # defect of the dynamics row block k + 1
                G = np.zeros((na, na + nu))
                G[:nx, :nx] = Jc[k][:, :nx]; G[:nx, na:] = Jc[k][:, nx:]; G[nx:, na:] = np.eye(nu)
                sk = np.zeros(na + nu)
                sk[:nx] = g[:nx]; sk[na:] = g[nx:]
                qv = sk + G.T @ (Pt[k + 1] @ cb + pt[k + 1])
                kff[k] = -np.linalg.solve(Lq[k].T, np.linalg.solve(Lq[k], qv[na:]))
                Q_xu_K = None
                # p_k = q_x + Q_xu kff   (Q_xu = -K' Quu  =>  Q_xu kff = K' (-Quu kff) = K' q_u)
                pm[k] = qv[:na] + Kg[k].T @ qv[na:]
            pt[k] = pm[k] - Pt[k] @ (E[k] * pm[k]) if soft else pm[k]
        dx, dy = np.zeros(S.n), np.zeros(mc + N * K)
        xi = np.zeros(na)
        if soft:                                                            # row 0: dx_0 - e dy_0 = b_0, dy_0 = -(P_0 dx_0 + p_0)
            rhs0 = np.zeros(na)
            rhs0[:nx] = b_c[0]
            xi = rhs0 - E[0] * pm[0]
            xi = xi - E[0] * (Pt[0] @ xi)                                   # (I + E P)^-1 = I - E P~
            xi[nx:] = 0.0
        else:
            xi[:nx] = b_c[0]
        dy[:nx] = -(Pm[0] @ xi + pm[0])[:nx]
        for k in range(N):
            u = Kg[k] @ xi + kff[k]
            dx[k * nv:k * nv + nx] = xi[:nx]
            dx[k * nv + nx:(k + 1) * nv] = u
            dy[mc + k * K: mc + (k + 1) * K] = (Jd[k] @ np.concatenate([xi[:nx], u]) - b_d[k]) / e_d[k]
            a = np.zeros(na)
            a[:nx] = Jc[k] @ np.concatenate([xi[:nx], u]) - b_c[k + 1]
            a[nx:] = u
            if soft:
                a = a - E[k + 1] * pm[k + 1]
                a = a - E[k + 1] * (Pt[k + 1] @ a)
            xi = a
            dy[nx + k * nx: nx + (k + 1) * nx] = (Pm[k + 1] @ xi + pm[k + 1])[:nx]
        dx[N * nv:] = xi[:nx]
        return dx, dy

    @staticmethod
    def _kmul(W, J, dxx, e, v):
        n = W.shape[0]
        return np.concatenate([W @ v[:n] + dxx * v[:n] + J.T @ v[n:], J @ v[:n] - e * v[n:]])

    @staticmethod
    def _n_negative(ldu, piv):
        """negative eigenvalues of the block diagonal of LAPACK's Bunch-Kaufman factorisation (a 2 x 2 pivot has one of each sign)."""
        nn, k, neg = ldu.shape[0], 0, 0
        d = np.diagonal(ldu)
        while k < nn:
            if piv[k] > 0:
                if d[k] < 0:
                    neg += 1
                k += 1
            else:
                neg += 1
                k += 2
        return neg

    # -- initialisation ------------------------------------------------------------------------------------------------
    @staticmethod
    def push(v, lo, hi, k1, k2):
        v = v.copy()
        fl, fu = np.isfinite(lo), np.isfinite(hi)
        both = fl & fu
        lo_, hi_ = np.where(fl, lo, 0.0), np.where(fu, hi, 0.0)
        rng = np.where(both, hi_ - lo_, INF)
        pl = np.minimum(k1 * np.maximum(1.0, np.abs(lo_)), k2 * rng)
        pu = np.minimum(k1 * np.maximum(1.0, np.abs(hi_)), k2 * rng)
        v = np.where(fl, np.maximum(v, lo_ + pl), v)
        v = np.where(fu, np.minimum(v, hi_ - pu), v)
        return v

    def ls_multipliers(self, ev, z):
        """least-square estimate of y for given bound multipliers (section 3.6)."""
        P = self.P
        zxL, zxU, ztL, ztU = z
        sol = self.factor_ls(ev["J"])
        if sol is None:
            return np.zeros(P.m)
        _, _, y = sol(-(ev["grad"] - zxL + zxU), -(P.t_rho - ztL + ztU), np.zeros(P.m))
        return y

    def factor_ls(self, J):
        save = self.delta_w_last
        s = self.factor(np.zeros((self.P.n, self.P.n)), J, np.ones(self.P.n), np.ones(self.P.nt), 1.0)
        self.delta_w_last = save
        return s

    # -- main loop -----------------------------------------------------------------------------------------------------
    def run(self, x, t, y, z, mu, budget):
        """Returns (status, x, t, y, z, mu, iterations).  status: 'optimal', 'acceptable', 'max_iter', 'resto_failed',
        'local_infeasibility', 'resto_converged_feasible', 'orig_progress' (restoration only), 'error'."""
        P, o = self.P, self.o
        tau = max(o["tau_min"], 1.0 - mu)
        ev = P.evaluate(x, 2, mu)
        theta0 = float(np.sum(np.abs(self.residual(ev["g"], t))))
        fact = o["resto_theta_max_fact"] if self.in_resto else o["theta_max_fact"]
        self.theta_max = fact * max(1.0, theta0)
        self.theta_min = o["theta_min_fact"] * max(1.0, theta0)
        n_acc = 0
        first_iter = True
        status = "max_iter"
        while True:
            E0, dinf, pinf, comp, r = self.errors(ev, x, t, y, z, 0.0)
            if self.trace is not None:
                self.trace.append(dict(it=self.total_iters(), resto=self.in_resto, E0=E0, dinf=dinf, pinf=pinf, comp=comp, mu=mu, f=ev["f"],
                                       theta=float(np.sum(np.abs(r))), delta=getattr(self, "last_delta", (0, 0))[0], alpha=getattr(self, "last_alpha", 0.0)))
            # ---- convergence ---------------------------------------------------------------------------------------------
            if self.in_resto:
                st = self.outer_progress(x, t, first_iter)
                if st:
                    status = "orig_progress"
                    break
            df = getattr(P, "df", 1.0)
            dgs = getattr(P, "dg", None)
            un_pinf = float(np.max(np.abs(r / dgs), initial=0.0)) if dgs is not None else pinf
            if E0 <= o["tol"] and dinf / df <= o["dual_inf_tol"] and un_pinf <= o["constr_viol_tol"] and comp / df <= o["compl_inf_tol"]:
                status = "optimal"
                break
            if E0 <= o["acceptable_tol"] and dinf / df <= o["acceptable_dual_inf_tol"] and un_pinf <= o["acceptable_constr_viol_tol"] \
                    and comp / df <= o["acceptable_compl_inf_tol"]:
                n_acc += 1
                if n_acc >= o["acceptable_iter"]:
                    status = "acceptable"
                    break
            else:
                n_acc = 0
            if self.total_iters() >= budget:
                status = "max_iter"
                break
            if not self.in_resto and o["recalc_y_iter"] > 0 and not getattr(self, "recalc_y", False):
                sxL_, sxU_, stL_, stU_ = self.slacks(x, t)
                nb_ = sum(int(np.sum(np.isfinite(s_))) for s_ in (sxL_, sxU_, stL_, stU_))
                sc_ = max(o["s_max"], float(sum(np.sum(np.abs(z_)) for z_ in z)) / max(1, nb_)) / o["s_max"]
                mu_min_ = min(o["tol"], o["compl_inf_tol"]) / (o["barrier_tol_factor"] + 1.0)
                floor_ = (mu <= 10.0 * mu_min_ and n_acc == 0 and max(pinf, comp / sc_) <= o["acceptable_tol"] and un_pinf <= o["acceptable_constr_viol_tol"]
                          and comp <= o["acceptable_compl_inf_tol"] * df)
                self.n_floor = getattr(self, "n_floor", 0) + 1 if floor_ else 0
                if self.n_floor >= o["recalc_y_iter"]:
                    self.recalc_y = True
                    y = self.ls_multipliers(ev, z)
                    continue
            if not self.in_resto and o["floor_iter"] > 0:
                sxL_, sxU_, stL_, stU_ = self.slacks(x, t)
                nb_ = sum(int(np.sum(np.isfinite(s_))) for s_ in (sxL_, sxU_, stL_, stU_))
                sc_ = max(o["s_max"], float(sum(np.sum(np.abs(z_)) for z_ in z)) / max(1, nb_)) / o["s_max"]
                mu_min_ = min(o["tol"], o["compl_inf_tol"]) / (o["barrier_tol_factor"] + 1.0)
                at_floor = (mu <= 10.0 * mu_min_ and n_acc == 0 and max(pinf, comp / sc_) <= o["acceptable_tol"] and un_pinf <= o["acceptable_constr_viol_tol"]
                            and comp <= o["acceptable_compl_inf_tol"] * df)
                self.n_at_floor = getattr(self, "n_at_floor", 0) + 1 if at_floor else 0
                if self.n_at_floor >= o["floor_iter"]:
                    status = "max_iter"
                    break
            if o["stall_iter"] > 0 and getattr(self, "n_tiny", 0) >= o["stall_iter"]:
                status = "resto_failed" if self.in_resto else "max_iter"     # (the stall rule, in either phase: same class as running out of iterations)
                break
            # ---- barrier parameter -----------------------------------------------------------------------------------------
            mu_min = min(o["tol"], o["compl_inf_tol"]) / (o["barrier_tol_factor"] + 1.0)
            while True:
                Emu = self.errors(ev, x, t, y, z, mu)[0]
                if Emu > o["barrier_tol_factor"] * mu or mu <= mu_min:
                    break
                mu_new = max(mu_min, min(o["mu_linear_decrease_factor"] * mu, mu ** o["mu_superlinear_decrease_power"]))
                if mu_new == mu:
                    break
                mu = mu_new
                tau = max(o["tau_min"], 1.0 - mu)
                self.filter.clear()
                if getattr(P, "eta", None) is not None:                     # the restoration's objective depends on mu
                    ev = P.evaluate(x, 2, mu)
            first_iter = False
            # ---- search direction ------------------------------------------------------------------------------------------
            zxL, zxU, ztL, ztU = z
            sxL, sxU, stL, stU = self.slacks(x, t)
            sig_x = zxL / sxL + zxU / sxU
            sig_t = ztL / stL + ztU / stU
            W = ev["hess"](y)
            J = ev["J"]
            solver = self.factor(W, J, sig_x, sig_t, mu)
            gx, gt = self.barrier_grad(ev["grad"], x, t, mu)
            r = self.residual(ev["g"], t)
            need_resto = solver is None
            if not need_resto:
                rhs_x = -(gx + J.T @ y)
                rhs_t = -(gt + P.t_sig * y[P.t_row])
                dx, dt, dy = solver(rhs_x, rhs_t, -r)
                dz = (mu / sxL - zxL - zxL * dx / sxL, mu / sxU - zxU + zxU * dx / sxU,
                      mu / stL - ztL - ztL * dt / stL, mu / stU - ztU + ztU * dt / stU)
                dz = tuple(np.where(np.isfinite(s_), d_, 0.0) for s_, d_ in zip((sxL, sxU, stL, stU), dz))
                # ---- filter line search ----------------------------------------------------------------------------------
                a_max = min(_ftb(tau, sxL, dx), _ftb(tau, sxU, -dx), _ftb(tau, stL, dt), _ftb(tau, stU, -dt))
                a_z = min(_ftb(tau, zxL, dz[0]), _ftb(tau, zxU, dz[1]), _ftb(tau, ztL, dz[2]), _ftb(tau, ztU, dz[3]))
                theta = float(np.sum(np.abs(r)))
                phi = self.barrier(ev["f"], x, t, mu)
                gBD = float(gx @ dx + gt @ dt)
                acc = self.line_search(x, t, dx, dt, a_max, theta, phi, gBD, mu, tau, solver, r, rhs_x, rhs_t)
                need_resto = acc is None
            if need_resto:
                if self.in_resto:
                    status = "resto_failed"
                    break
                if o.get("restoration", "ipopt") == "none":               # the caller has another solver for these (csrc/mpc_vtol_ms.hip: the
                    status = "needs_resto"                                  # condensed wave kernel and its restoration)
                    break
                rs = self.restoration(x, t, y, z, mu, ev, budget)
                self.n_tiny = 0
                if rs[0] != "ok":
                    status, x, t = rs[0], rs[1], rs[2]
                    ev = P.evaluate(x, 2, mu)
                    break
                _, x, t, z = rs
                y = np.zeros(P.m)
                ev = P.evaluate(x, 2, mu)
                if o["constr_mult_reset_threshold"] > 0.0:
                    yl = self.ls_multipliers(ev, z)
                    if np.max(np.abs(yl), initial=0.0) <= o["constr_mult_reset_threshold"]:
                        y = yl
                continue
            alpha, x, t, ev_new = acc
            self.last_alpha = alpha
            self.n_tiny = getattr(self, "n_tiny", 0) + 1 if alpha < o["stall_alpha"] else 0
            self.safe_slacks(x, t, mu)
            y = y + alpha * dy
            z = tuple(z_ + a_z * d_ for z_, d_ in zip(z, dz))
            sl = self.slacks(x, t)
            ks = o["kappa_sigma"]
            z = tuple(np.where(np.isfinite(s_), np.maximum(np.minimum(z_, ks * mu / s_), mu / (ks * s_)), 0.0) for z_, s_ in zip(z, sl))
            ev = P.evaluate(x, 2, mu)
            self.iters += 1
            if getattr(self, "recalc_y", False) and not self.in_resto:
                y = self.ls_multipliers(ev, z)
        return status, x, t, y, z, mu

    def total_iters(self):
        return self.iters + (self.outer.total_iters() if self.outer is not None else 0)

    # -- line search ---------------------------------------------------------------------------------------------------
    def trial(self, x, t, mu):
        P = self.P
        self.safe_slacks(x, t, mu)                                          # IPOPT computes trial slacks through CalculateSafeSlack as well
        ev = P.evaluate(x, 0, mu)
        if not (np.all(np.isfinite(ev["g"])) and np.isfinite(ev["f"])):
            return None
        phi = self.barrier(ev["f"], x, t, mu)
        if not np.isfinite(phi):
            return None
        return phi, float(np.sum(np.abs(self.residual(ev["g"], t)))), ev

    def acceptable_to_iterate(self, phi_t, th_t, phi, theta):
        o = self.o
        if phi_t > phi:
            bas = max(1.0, math.log10(abs(phi))) if abs(phi) > 10.0 else 1.0
            if math.log10(phi_t - phi) > o["obj_max_inc"] + bas:
                return False
        return compare_le(th_t, (1.0 - o["gamma_theta"]) * theta, theta) or compare_le(phi_t - phi, -o["gamma_phi"] * theta, phi)

    def line_search(self, x, t, dx, dt, a_max, theta, phi, gBD, mu, tau, solver, r, rhs_x, rhs_t):
        o, P = self.o, self.P
        if gBD < 0.0:
            a_min = o["gamma_theta"]
            a_min = min(a_min, o["gamma_phi"] * theta / (-gBD))
            if theta <= self.theta_min:
                a_min = min(a_min, o["delta"] * theta ** o["s_theta"] / (-gBD) ** o["s_phi"])
        else:
            a_min = o["gamma_theta"]
        a_min *= o["alpha_min_frac"]

        def ftype(a):
            return gBD < 0.0 and a * (-gBD) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"]

        def armijo(a, phi_t):
            return compare_le(phi_t - phi, o["eta_phi"] * a * gBD, phi)

        def check(a, phi_t, th_t):
            if th_t > self.theta_max:
                return False
            if a > 0.0 and ftype(a) and theta <= self.theta_min:
                ok = armijo(a, phi_t)
            else:
                ok = self.acceptable_to_iterate(phi_t, th_t, phi, theta)
            return ok and self.filter.acceptable(phi_t, th_t)

        alpha = a_max
        first = True
        accepted = None
        while alpha > a_min or first:
            xt, tt = x + alpha * dx, t + alpha * dt
            tr = self.trial(xt, tt, mu)
            if tr is not None:
                phi_t, th_t, ev_t = tr
                if check(alpha, phi_t, th_t):
                    accepted = (alpha, xt, tt, ev_t, alpha, phi_t)
                    break
                if first and th_t >= theta and o["max_soc"] > 0:
                    # second-order correction (section 3.4 / A-5.5 .. A-5.9)
                    c_soc = alpha * r + self.residual(ev_t["g"], tt)
                    th_old = theta
                    th_soc_old = th_t
                    for _ in range(o["max_soc"]):
                        dxs, dts, _dy = solver(rhs_x, rhs_t, -c_soc)
                        sxL, sxU, stL, stU = self.slacks(x, t)
                        a_soc = min(_ftb(tau, sxL, dxs), _ftb(tau, sxU, -dxs), _ftb(tau, stL, dts), _ftb(tau, stU, -dts))
                        xs, ts = x + a_soc * dxs, t + a_soc * dts
                        trs = self.trial(xs, ts, mu)
                        if trs is None:
                            break
                        phi_s, th_s, ev_s = trs
                        if check(alpha, phi_s, th_s):
                            accepted = (a_soc, xs, ts, ev_s, alpha, phi_s)
                            break
                        if th_s > o["kappa_soc"] * th_soc_old:
                            break
                        th_soc_old = th_s
                        c_soc = a_soc * c_soc + self.residual(ev_s["g"], ts)
                    if accepted is not None:
                        break
            first = False
            alpha *= o["alpha_red_factor"]
        if accepted is None:
            return None
        a_step, xt, tt, ev_t, a_test, phi_t = accepted
        if not ftype(a_test) or not armijo(a_test, phi_t):
            self.filter.add(phi - o["gamma_phi"] * theta, (1.0 - o["gamma_theta"]) * theta)
        return a_step, xt, tt, ev_t

    # -- restoration ---------------------------------------------------------------------------------------------------
    def restoration(self, x, t, y, z, mu, ev, budget):
        P, o = self.P, self.o
        r = self.residual(ev["g"], t)
        if float(np.max(np.abs(r), initial=0.0)) <= o["resto_failure_feasibility_threshold"]:
            return ("resto_failed", x, t)                                   # "restoration phase is called at a point that is almost feasible"
        theta = float(np.sum(np.abs(r)))
        phi = self.barrier(ev["f"], x, t, mu)
        self.filter.add(phi - o["gamma_phi"] * theta, (1.0 - o["gamma_theta"]) * theta)
        RP = _Resto(P, x, o)
        ra = _Algo(RP, o, resto_of=self, trace=self.trace)
        ra.orig = dict(mu=mu, theta=theta, phi=phi, pinf=float(np.max(np.abs(r))))
        mu_r = max(mu, float(np.max(np.abs(r))))
        rho = RP.rho
        r_all = r
        r = r[RP.el]
        a = (mu_r - rho * r) / (2.0 * rho)
        nn = a + np.sqrt(a * a + mu_r * r / (2.0 * rho))                    # eq. (33): row + n - p = 0 with residual r = p - n ... sign below
        # the rows are  g + A_s s + n - p = 0:  p - n = r
        pp = r + nn
        tt = np.concatenate([t, nn, pp])
        zxL, zxU, ztL, ztU = z
        zr = (np.minimum(rho, zxL), np.minimum(rho, zxU),
              np.concatenate([np.minimum(rho, ztL), mu_r / nn, mu_r / pp]), np.concatenate([np.minimum(rho, ztU), np.zeros(2 * len(RP.el))]))
        zr = tuple(np.where(np.isfinite(s_), z_, 0.0) for z_, s_ in zip(zr, ra.slacks(x, tt)))
        st, xr, tr, yr, zrr, mur = ra.run(x.copy(), tt, np.zeros(P.m), zr, mu_r, budget)
        self.iters += ra.iters
        ra.outer = None
        ns = P.nt
        if st == "orig_progress":
            znew = (zrr[0], zrr[1], zrr[2][:ns], zrr[3][:ns])
            if max(float(np.max(zz, initial=0.0)) for zz in znew) > o["bound_mult_reset_threshold"]:
                sl = self.slacks(xr, tr[:ns])
                znew = tuple(np.where(np.isfinite(s_), 1.0, 0.0) for s_ in sl)
            return ("ok", xr, tr[:ns], znew)
        if st in ("optimal", "acceptable"):
            rr = self.residual(P.evaluate(xr, 0, mu)["g"], tr[:ns])
            feas = float(np.max(np.abs(rr), initial=0.0)) <= 1e2 * o["tol"]
            return ("resto_converged_feasible" if feas else "local_infeasibility", xr, tr[:ns])
        return (st if st == "max_iter" else "resto_failed", xr, tr[:ns])

    def outer_progress(self, x, tt, first_iter):
        """RestoFilterConvergenceCheck: leave the restoration when the (x, s) part is acceptable to the original filter and to the
        iterate the restoration started from, with the infeasibility reduced to kappa_resto of what it was."""
        if first_iter:
            return False
        out, o = self.outer, self.o
        ns = out.P.nt
        s = tt[:ns]
        ev = out.P.evaluate(x, 0, None)
        r = out.residual(ev["g"], s)
        pinf = float(np.max(np.abs(r), initial=0.0))
        if pinf > o["required_infeasibility_reduction"] * self.orig["pinf"]:
            return False
        phi_t = out.barrier(ev["f"], x, s, self.orig["mu"])
        th_t = float(np.sum(np.abs(r)))
        if not np.isfinite(phi_t):
            return False
        if not out.filter.acceptable(phi_t, th_t):
            return False
        return out.acceptable_to_iterate(phi_t, th_t, self.orig["phi"], self.orig["theta"])


STATUS_OF = dict(optimal=0, acceptable=0, local_infeasibility=1, max_iter=2, resto_failed=2, resto_converged_feasible=2, error=2, needs_resto=4)


def solve_nlp(nlp, w0, opts=None, trace=None):
    """IPOPT's algorithm (see the module docstring) on an NLP object with evaluate / x_bounds / d_lo / d_hi.  Returns a dict."""
    try:                                                                 # matrices of order < 1000: threaded BLAS only costs
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            return _solve_nlp(nlp, w0, opts, trace)
    except ImportError:
        return _solve_nlp(nlp, w0, opts, trace)


def _solve_nlp(nlp, w0, opts, trace):
    o = dict(OPTS)
    if opts:
        o.update(opts)
    w0 = np.asarray(w0, dtype=float)
    P = _Regular(nlp, w0, o)
    A = _Algo(P, o, trace=trace)
    x = A.push(w0, P.x_L, P.x_U, o["bound_push"], o["bound_frac"])
    ev = P.evaluate(x, 2)
    s = A.push(ev["g"][P.m_c:], P.t_L, P.t_U, o["bound_push"], o["bound_frac"])
    sl = A.slacks(x, s)
    z = tuple(np.where(np.isfinite(s_), o["bound_mult_init_val"], 0.0) for s_ in sl)
    y = A.ls_multipliers(ev, z)
    if np.max(np.abs(y), initial=0.0) > o["constr_mult_init_max"]:
        y = np.zeros(P.m)
    status, x, s, y, z, mu = A.run(x, s, y, z, o["mu_init"], o["max_iter"])
    ev = nlp.evaluate(x, 0)
    return dict(x=x, status=status, code=STATUS_OF.get(status, 2), iters=A.iters, f=ev["f"], c=ev["c"], d=ev["d"], y=y,
                mu=mu, obj_scale=P.df, con_scale=P.dg)


def solve(model, x0, u_prev, goal, obs, N=None, opts=None, return_info=False, trace=None):
    """One control step's NLP from do-mpc's starting point.  Returns u_0, status code (0 optimal / acceptable, 1 converged to a point
    of local infeasibility, 2 everything else), iterations [, info]."""
    nlp = StageNLP(model, x0, u_prev, goal, obs, N)
    r = solve_nlp(nlp, nlp.initial_guess(), opts, trace)
    u0 = nlp.u0(r["x"])
    if return_info:
        X, U = nlp.split(r["x"])
        r.update(X=X, U=U, n_eval=nlp.n_eval)
        return u0, r["code"], r["iters"], r
    return u0, r["code"], r["iters"]
