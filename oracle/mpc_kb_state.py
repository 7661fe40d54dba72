"""MPC-CBF for KinematicBicycle2D_C3BF / KinematicBicycle2D_DPCBF (position_control/mpc_cbf.py:31-33,68-73,205-211,312-315 over
dynamic_env/kinematic_bicycle2D_c3bf.py:77-118 and dynamic_env/kinematic_bicycle2D_dpcbf.py:86-142): float64 problem functions for
oracle.mpc_cbf.solve(evaluate_fn=...).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned** for the solver like oracle/mpc_cbf.py (do-mpc / casadi / IPOPT
absent); the barrier functions and the registered constraint are pinned on the reference's own agent_barrier_dt /
compute_cbf_constraint (tests/golden/mpc_functions.npz: hk, dh, cons of both models; tests/test_oracle_mpc_golden.py).

  prediction / cost / bounds   as KinematicBicycle2D (oracle/mpc_gn.py: kb_F, Q = diag(50, 50, 1, 1), R = (.5, 5000), |v_k| <= v_max,
                               |a| <= a_max, |beta| <= beta_max)
  CBF row (stage k, obstacle j)   d_h + alpha h_k = h(S(x_k, u_k)) - (1 - alpha) h(x_k) >= 0,  alpha = 0.15, S = the robot's step()
                               (Euler + speed clip), h a function of the FULL state (x, y, theta, v) and of the obstacle's position and
                               radius (obs columns 0..2; its velocity columns are NOT seen by the MPC, see _rel):
     C3BF   h = <p_rel, v_rel> + |p_rel| |v_rel| sqrt(max(|p_rel|^2 - ego^2, 0)) / |p_rel|,  ego = (r + R) 1.01
     DPCBF  h = v_n0 + lam v_n1^2 + mu in the line-of-sight frame,  d = max(|p_rel|^2 - ego^2, 1e-6), ego = (r + R) 1.05,
            lam = 0.1 sqrt(s^2 - 1) / ego sqrt(d) / |v_rel|,  mu = 0.5 sqrt(s^2 - 1) / ego sqrt(d),  s = 1.05

So a stage has TWO barrier points of dimension four (x_k and y1 = S(x_k, u_k)) where the rel-degree-2 models of oracle/mpc_gn.py have
three of dimension two.  First and second derivatives of h come from second-order forward-mode differentiation over the four state
components (class HD below; the HIP kernel carries the same arithmetic in registers), so there is no hand-derived formula to get
wrong.  Inside the inflated radius C3BF's square root is held at zero with zero derivatives (casadi would produce 0 * inf there; the
regime is a collision).
Rows: [CBF (stage major, obstacle minor) | v_max - v_k, v_k + v_max (k = 1..N) | u_hi - z | z - u_lo].
"""
import math

import numpy as np

from . import mpc_cbf as M
from . import mpc_gn as G

NS = 4


class HD:
    """value, gradient (4) and Hessian (4 x 4) with respect to the state (x, y, theta, v)."""
    __slots__ = ("v", "g", "H")
    __array_ufunc__ = None

    def __init__(self, v, g=None, H=None):
        self.v = float(v)
        self.g = np.zeros(NS) if g is None else g
        self.H = np.zeros((NS, NS)) if H is None else H

    @staticmethod
    def var(v, i):
        g = np.zeros(NS); g[i] = 1.0
        return HD(v, g)

    @staticmethod
    def lift(a):
        return a if isinstance(a, HD) else HD(a)

    def chain(self, f, f1, f2):
        """f(self) given f, f', f'' at self.v."""
        return HD(f, f1 * self.g, f1 * self.H + f2 * np.outer(self.g, self.g))

    def __add__(self, o):
        o = HD.lift(o)
        return HD(self.v + o.v, self.g + o.g, self.H + o.H)
    __radd__ = __add__

    def __sub__(self, o):
        o = HD.lift(o)
        return HD(self.v - o.v, self.g - o.g, self.H - o.H)

    def __rsub__(self, o):
        return HD.lift(o) - self

    def __neg__(self):
        return HD(-self.v, -self.g, -self.H)

    def __mul__(self, o):
        o = HD.lift(o)
        return HD(self.v * o.v, self.v * o.g + o.v * self.g,
                  self.v * o.H + o.v * self.H + np.outer(self.g, o.g) + np.outer(o.g, self.g))
    __rmul__ = __mul__

    def recip(self):
        r = 1.0 / self.v
        return self.chain(r, -r * r, 2.0 * r * r * r)

    def __truediv__(self, o):
        return self * HD.lift(o).recip()

    def __rtruediv__(self, o):
        return HD.lift(o) * self.recip()


def hd_sqrt(a):
    r = math.sqrt(a.v)
    return a.chain(r, 0.5 / r, -0.25 / (r * a.v))


def hd_sin(a):
    s, c = math.sin(a.v), math.cos(a.v)
    return a.chain(s, c, -s)


def hd_cos(a):
    s, c = math.sin(a.v), math.cos(a.v)
    return a.chain(c, -s, -c)


def _rel(x, obs):
    """p_rel, v_rel and their norms as HD values of the state x = (HD, HD, HD, HD).  The obstacle's velocity is ZERO here, as in the
    reference's MPC: set_cbf_constraint hands agent_barrier_dt the row `_obs[i, :]`, a 1 x 7 casadi slice (mpc_cbf.py:299-301), so
    the test `obs.shape[0] > 3` of both barriers (c3bf.py:88-93, dpcbf.py:95-100) is False and obs_vel = 0 -- the goldens, produced
    by running that code, say the same."""
    px, py = obs[0] - x[0], obs[1] - x[1]
    c, s = hd_cos(x[2]), hd_sin(x[2])
    vx, vy = 0.0 - x[3] * c, 0.0 - x[3] * s
    pm2 = px * px + py * py
    vm = hd_sqrt(vx * vx + vy * vy)
    return px, py, vx, vy, pm2, vm


def h_c3bf(x, obs, radius, beta=1.01):
    """kinematic_bicycle2D_c3bf.py:83-109."""
    px, py, vx, vy, pm2, vm = _rel(x, obs)
    ego = (obs[2] + radius) * beta
    pm = hd_sqrt(pm2)
    a = pm2 - ego * ego
    root = hd_sqrt(a) if a.v > 0.0 else HD(0.0)
    return px * vx + py * vy + pm * vm * root / pm


def h_dpcbf(x, obs, radius, s=1.05):
    """kinematic_bicycle2D_dpcbf.py:91-136.  cos / sin of atan2(p_y, p_x) are p_x / |p|, p_y / |p|."""
    px, py, vx, vy, pm2, vm = _rel(x, obs)
    ego = (obs[2] + radius) * s
    pm = hd_sqrt(pm2)
    cr, sr = px / pm, py / pm
    vn0, vn1 = cr * vx + sr * vy, cr * vy - sr * vx
    a = pm2 - ego * ego
    d = a if a.v > 1e-6 else HD(1e-6)
    kl, km = 0.1 * math.sqrt(s * s - 1.0) / ego, 0.5 * math.sqrt(s * s - 1.0) / ego
    rd = hd_sqrt(d)
    return vn0 + (kl * rd / vm) * vn1 * vn1 + km * rd


class HDV:
    """HD over a batch: value (P,), gradient (P, 4), Hessian (P, 4, 4).  Same arithmetic, operation for operation, as HD (held to it
    in tests/test_oracle_mpc_kb.py); it exists because evaluate() needs the barrier at 2 N K (point, obstacle) pairs per call and
    the scalar class costs a Python object per operation."""
    __slots__ = ("v", "g", "H")
    __array_ufunc__ = None

    def __init__(self, v, g=None, H=None):
        self.v = np.asarray(v, dtype=np.float64)
        n = self.v.shape[0]
        self.g = np.zeros((n, NS)) if g is None else g
        self.H = np.zeros((n, NS, NS)) if H is None else H

    @staticmethod
    def var(v, i):
        v = np.asarray(v, dtype=np.float64)
        g = np.zeros((v.shape[0], NS)); g[:, i] = 1.0
        return HDV(v, g)

    def lift(self, a):
        return a if isinstance(a, HDV) else HDV(np.full(self.v.shape, float(a)) if np.ndim(a) == 0 else a)

    def chain(self, f, f1, f2):
        return HDV(f, f1[:, None] * self.g, f1[:, None, None] * self.H + f2[:, None, None] * (self.g[:, :, None] * self.g[:, None, :]))

    def __add__(self, o):
        o = self.lift(o)
        return HDV(self.v + o.v, self.g + o.g, self.H + o.H)
    __radd__ = __add__

    def __sub__(self, o):
        o = self.lift(o)
        return HDV(self.v - o.v, self.g - o.g, self.H - o.H)

    def __rsub__(self, o):
        return self.lift(o) - self

    def __neg__(self):
        return HDV(-self.v, -self.g, -self.H)

    def __mul__(self, o):
        o = self.lift(o)
        return HDV(self.v * o.v, self.v[:, None] * o.g + o.v[:, None] * self.g,
                   self.v[:, None, None] * o.H + o.v[:, None, None] * self.H + self.g[:, :, None] * o.g[:, None, :]
                   + o.g[:, :, None] * self.g[:, None, :])
    __rmul__ = __mul__

    def recip(self):
        r = 1.0 / self.v
        return self.chain(r, -r * r, 2.0 * r * r * r)

    def __truediv__(self, o):
        return self * self.lift(o).recip()

    def __rtruediv__(self, o):
        return self.lift(o) * self.recip()

    def where(self, mask, other):
        """self where mask, else the constant `other` (zero derivatives)."""
        return HDV(np.where(mask, self.v, other), np.where(mask[:, None], self.g, 0.0), np.where(mask[:, None, None], self.H, 0.0))


def _v_sqrt(a):
    r = np.sqrt(a.v)
    return a.chain(r, 0.5 / r, -0.25 / (r * a.v))


def barrier_batch(pts, obs, P, derivs=True):
    """barrier() at the rows of pts (P, 4) against the rows of obs (P, 7): h (P,), grad (P, 4), Hessian (P, 4, 4)."""
    pts = np.asarray(pts, dtype=np.float64); obs = np.asarray(obs, dtype=np.float64)
    n = pts.shape[0]
    x = [HDV.var(pts[:, i], i) if derivs else HDV(pts[:, i]) for i in range(NS)]
    px, py = obs[:, 0] - x[0], obs[:, 1] - x[1]
    sn, cs = np.sin(x[2].v), np.cos(x[2].v)
    c, s = x[2].chain(cs, -sn, -cs), x[2].chain(sn, cs, -sn)
    vx, vy = 0.0 - x[3] * c, 0.0 - x[3] * s
    pm2 = px * px + py * py
    vm = _v_sqrt(vx * vx + vy * vy)
    pm = _v_sqrt(pm2)
    if P["model"]["kind"] == "c3bf":
        ego = (obs[:, 2] + P["radius"]) * 1.01
        a = pm2 - ego * ego
        pos = a.v > 0.0
        root = _v_sqrt(HDV(np.where(pos, a.v, 1.0), a.g, a.H)).where(pos, 0.0)
        r = px * vx + py * vy + pm * vm * root / pm
    else:
        sc = 1.05
        ego = (obs[:, 2] + P["radius"]) * sc
        cr, sr = px / pm, py / pm
        vn0, vn1 = cr * vx + sr * vy, cr * vy - sr * vx
        a = pm2 - ego * ego
        big = a.v > 1e-6
        d = a.where(big, 1e-6)
        kl, km = 0.1 * math.sqrt(sc * sc - 1.0) / ego, 0.5 * math.sqrt(sc * sc - 1.0) / ego
        rd = _v_sqrt(d)
        r = vn0 + (kl * rd / vm) * vn1 * vn1 + km * rd
    return r.v, (r.g if derivs else None), (r.H if derivs else None)


def barrier(xv, obs, P, derivs=True):
    """h, grad (4), Hessian (4 x 4) of the model's barrier at the state xv."""
    fn = h_c3bf if P["model"]["kind"] == "c3bf" else h_dpcbf
    if derivs:
        r = fn([HD.var(xv[i], i) for i in range(NS)], obs, P["radius"])
        return r.v, r.g, r.H
    r = fn([HD(xv[i]) for i in range(NS)], obs, P["radius"])
    return r.v, None, None


def _model(kind, spec=None, dt=0.05):
    m = G.kb_model(spec, dt)
    m.update(name="KinematicBicycle2D_C3BF" if kind == "c3bf" else "KinematicBicycle2D_DPCBF", kind=kind, alpha=0.15)
    m.pop("alpha1"); m.pop("alpha2")
    return m


def c3bf_model(spec=None, dt=0.05):
    return _model("c3bf", spec, dt)


def dpcbf_model(spec=None, dt=0.05):
    return _model("dpcbf", spec, dt)


def params(model, N=10, **over):
    P = dict(M.DEFAULTS, N=N, dt=model["dt"], nu=2, u_lo=model["u_lo"], u_hi=model["u_hi"], radius=model["radius"],
             alpha=model["alpha"], model=model)
    if "slack_reset" in model:
        P["slack_reset"] = model["slack_reset"]
    P.update(over)
    P.setdefault("row_noise", 1e-15)
    return P


def evaluate(x0, z, u_prev, goal, obs, P, lam=None, level=2):
    mdl = P["model"]
    N, nx, nu, dt, spec = P["N"], 4, 2, mdl["dt"], mdl["spec"]
    n = N * nu
    K = obs.shape[0]
    Q, Rw = mdl["Q"], mdl["R"]
    vmax = spec["v_max"]
    wp = (P["alpha"] - 1.0, 1.0)                              # d_h + alpha h_k = h(y1) - (1 - alpha) h(x_k)
    xg = np.zeros(nx); xg[:2] = np.asarray(goal, dtype=np.float64)[:2]
    U = z.reshape(N, nu)
    der = level >= 1
    X = np.zeros((N + 1, nx)); X[0] = np.asarray(x0, dtype=np.float64)[:nx]
    Phi = np.zeros((N + 1, nx, n))
    pts = np.zeros((N, 2, nx)); Gm = np.zeros((N, 2, nx, n))
    jac = [None] * N
    for k in range(N):
        if der:
            E = np.zeros((nu, n)); E[:, k * nu:(k + 1) * nu] = np.eye(nu)
            xn, A, B = G.kb_F(X[k], U[k], spec, dt, True)
            Phi[k + 1] = A @ Phi[k] + B @ E
            y1, S1x, S1u = G.kb_S(X[k], U[k], spec, dt, True)
            Gm[k, 0] = Phi[k]; Gm[k, 1] = S1x @ Phi[k] + S1u @ E
            jac[k] = (A, S1x)
        else:
            xn = G.kb_F(X[k], U[k], spec, dt)
            y1 = G.kb_S(X[k], U[k], spec, dt)
        X[k + 1] = xn
        pts[k, 0], pts[k, 1] = X[k], y1
    f = 0.0
    for k in range(1, N + 1):
        e = X[k] - xg
        f += float(Q @ (e * e))
    up = np.concatenate([np.asarray(u_prev, dtype=np.float64)[:nu], z])
    du = up[nu:] - up[:-nu]
    Rd = np.tile(Rw, N)
    f += float(np.sum(Rd * du * du))
    # the barrier at every (stage, point, obstacle) at once (barrier_batch: the scalar barrier(), vectorised)
    hb, gb, Hb = barrier_batch(np.repeat(pts.reshape(N * 2, nx), K, axis=0), np.tile(obs[:, :7], (N * 2, 1)), P, der)
    hv = hb.reshape(N, 2, K)
    dh = gb.reshape(N, 2, K, nx) if der else np.zeros((N, 2, K, nx))
    Hh = Hb.reshape(N, 2, K, nx, nx) if der else np.zeros((N, 2, K, nx, nx))
    m = N * K + 2 * N + 2 * n
    g = np.zeros(m)
    g[: N * K] = (wp[0] * hv[:, 0] + wp[1] * hv[:, 1]).reshape(-1)
    o = N * K
    for k in range(1, N + 1):
        g[o] = vmax - X[k, 3]; g[o + 1] = X[k, 3] + vmax
        o += 2
    hi_, lo_ = np.tile(mdl["u_hi"], N), np.tile(mdl["u_lo"], N)
    g[o:o + n] = hi_ - z
    g[o + n:] = z - lo_
    out = dict(f=float(f), g=g, X=X, pts=pts.reshape(-1, nx))
    if level == 0:
        return out
    grad = np.zeros(n)
    for k in range(1, N + 1):
        grad += Phi[k].T @ (2.0 * Q * (X[k] - xg))
    Dm = np.eye(n) - np.eye(n, k=-nu)
    grad += 2.0 * Dm.T @ (Rd * du)
    J = np.zeros((m, n))
    for k in range(N):
        for j in range(K):
            J[k * K + j] = sum(wp[p] * (dh[k, p, j] @ Gm[k, p]) for p in range(2))
    o = N * K
    for k in range(1, N + 1):
        J[o] = -Phi[k][3]; J[o + 1] = Phi[k][3]
        o += 2
    J[o:o + n] = -np.eye(n)
    J[o + n:] = np.eye(n)
    out.update(grad=grad, J=J)
    if level == 1:
        return out
    lam = np.zeros(m) if lam is None else lam
    lc = lam[: N * K].reshape(N, K)
    ls = lam[N * K:N * K + 2 * N].reshape(N, 2)
    W = 2.0 * Dm.T @ (Rd[:, None] * Dm)
    for k in range(1, N + 1):
        W += 2.0 * Phi[k].T @ (Q[:, None] * Phi[k])
    for k in range(N):
        for p in range(2):
            Om = -wp[p] * np.einsum("j,jab->ab", lc[k], Hh[k, p])
            W += Gm[k, p].T @ Om @ Gm[k, p]
    # second derivatives of the dynamics and of step(), weighted by the costates of the Lagrangian (oracle/mpc_gn.py: evaluate)
    pk = np.zeros(nx)
    for k in range(N, -1, -1):
        mu_k = np.zeros(nx)
        if k >= 1:
            mu_k = 2.0 * Q * (X[k] - xg)
            mu_k[3] += ls[k - 1, 0] - ls[k - 1, 1]
        if k == N:
            pk = mu_k
            continue
        A, S1x = jac[k]
        nu_ = [-wp[p] * (lc[k] @ dh[k, p]) for p in range(2)]
        Hk = G.kb_H(X[k], U[k], spec, dt, pk) + G.kb_H(X[k], U[k], spec, dt, nu_[1], True)
        E = np.zeros((nu, n)); E[:, k * nu:(k + 1) * nu] = np.eye(nu)
        V = np.vstack([Phi[k], E])
        W += V.T @ Hk @ V
        pk = mu_k + nu_[0] + S1x.T @ nu_[1] + A.T @ pk
    out.update(W=W)
    return out


def solve(model, x0, u_prev, goal, obs, N=10, params_over=None, return_info=False):
    P = params(model, N, **(params_over or {}))
    P["model"] = dict(model, circles_only=True)               # no superellipsoid branch, no steep-barrier scaling
    return M.solve(x0, u_prev, goal, obs, params=P, return_info=return_info, evaluate_fn=evaluate)
