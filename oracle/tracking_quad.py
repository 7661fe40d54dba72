"""Closed loop of the reference's LocalTrackingController for Quad2D and Quad3D (SURVEY 8f-1 over the 8f-3 models).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned on tests/golden/closed_loop_quads.npz (the reference's own
robot functions and control_step, tests/golden/make_golden_quads.py).  Follows:
  Quad2D   nominal_input / stop / has_stopped / step        robots/quad2D.py:83-158
  Quad3D   step (RK4 + wraps) / nominal_input / stop / has_stopped / rotate_to   robots/quad3D.py:100-257; A, B, B2 :70-98
  LocalTrackingController  X0 padding :80-93, set_waypoints / filter_waypoints :197-262 (Quad3D: 3-D waypoints, the third
           column is the z goal), update_goal :497-535 (Quad2D skips 'rotate'), get_nearest_unpassed_obs :345-403 (both
           models take every obstacle: angle_unpassed = 2 pi), control_step :559-668, is_in_fov robots/robot.py:854-872 with
           yaw = X[2] (Quad2D) / X[5] (Quad3D) robots/robot.py:449-452
  MPCCBF.solve_control_problem protocol                     position_control/mpc_cbf.py:366-402
  VTOL2D   X0 padding tracking.py:94-99 (cruise at 5 m/s), 'rotate' skipped (:512-513), is_in_fov always True (robot.py:858-860), zero
           nominal_input / stop (vtol2D.py:459-465), has_stopped on the planar speed (:467-469), obstacles inside the 1.2 pi cone about
           the pitch angle first and the nearest of all when the cone is empty (tracking.py:354-355,389-394), ground and pitch tests
           (:490-495: |theta| against robot_spec['pitch_max'] as given -- degrees), step = oracle/mpc_vtol.vt_S.  The reference's
           closed loop for this model cannot be run here (do-mpc / IPOPT absent): this part is a restatement, not pinned on a run.
"""
import math

import numpy as np

from . import mpc_cbf as M
from . import mpc_gn as OG
from . import mpc_lin as OL
from . import robots as R
from .tracking import is_collide

GRAV3 = 9.8                                                   # quad3D.py:67


def q2_nominal(X, goal, spec):
    """Quad2D.nominal_input (quad2D.py:88-143), default gains."""
    k_px, k_dx, k_pz, k_dz, k_pt, k_dt = 3.0, 0.5, 0.1, 0.5, 0.05, 0.05
    m, g = spec["mass"], 9.81
    r = spec["radius"]
    x, z, th, xd, zd, thd = X
    e_x, e_z = goal[0] - x, goal[1] - z
    xdd = k_px * e_x + k_dx * (-xd)
    zdd = k_pz * e_z + k_dz * (-zd)
    a_x, a_z = xdd, zdd + g
    T = m * np.sqrt(a_x ** 2 + a_z ** 2)
    th_d = -np.arctan2(a_x, a_z)
    e_th = th_d - th
    e_th = np.arctan2(np.sin(e_th), np.cos(e_th))
    tau = np.clip(k_pt * e_th + k_dt * (-thd), -1, 1)
    F_r = np.clip((T + tau / r) / 2.0, spec["f_min"], spec["f_max"])
    F_l = np.clip((T - tau / r) / 2.0, spec["f_min"], spec["f_max"])
    return np.array([F_r, F_l])


def q2_stop(X, spec):
    return q2_nominal(X, X[:2], spec)                         # quad2D.py:145-154


def q2_has_stopped(X, tol=0.05):
    return bool(np.linalg.norm(X[3:5]) < tol)                 # quad2D.py:156-158


def q2_step(X, U, dt, spec):
    return R.step(R.MODEL_QUAD2D, X, U, dt, spec)


def q3_matrices(spec):
    A, B = OL.quad3d_matrices(spec)[:2]
    L, nu = spec["L"], spec["nu"]
    B2 = np.array([[1, 1, 1, 1], [0, L, 0, -L], [L, 0, -L, 0], [nu, -nu, nu, -nu]], dtype=float)   # quad3D.py:84-89
    return A, B, B2


def q3_step(X, U, dt, spec):
    """Quad3D.step (quad3D.py:113-151): RK4 on the linear model, then the three angles wrapped."""
    A, B, _ = q3_matrices(spec)
    X = np.asarray(X, dtype=float)
    U = np.asarray(U, dtype=float)
    k1 = A @ X + B @ U
    k2 = A @ (X + dt / 2 * k1) + B @ U
    k3 = A @ (X + dt / 2 * k2) + B @ U
    k4 = A @ (X + dt * k3) + B @ U
    Xn = X + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    for i in (3, 4, 5):
        Xn[i] = R.angle_normalize(Xn[i])
    return Xn


def _q3_alloc(wrench, spec):
    _, _, B2 = q3_matrices(spec)
    return np.clip(np.linalg.pinv(B2) @ wrench, spec["u_min"], spec["u_max"])


def q3_nominal(X, goal, spec, k_p=1.0, k_d=2.0, k_ang=5.0):
    """Quad3D.nominal_input (quad3D.py:153-199)."""
    pe = np.asarray(goal[:3], dtype=float) - X[0:3]
    ve = -X[6:9]
    ax, ay, az = k_p * pe[0] + k_d * ve[0], k_p * pe[1] + k_d * ve[1], k_p * pe[2] + k_d * ve[2]
    th_d, ph_d, F = ax / GRAV3, -ay / GRAV3, spec["mass"] * az
    ty = spec["Iy"] * (k_ang * (th_d - X[3]) + k_d * (-X[9]))
    tx = spec["Ix"] * (k_ang * (ph_d - X[4]) + k_d * (-X[10]))
    tz = spec["Iz"] * (k_ang * (0 - X[5]) + k_d * (-X[11]))
    return _q3_alloc(np.array([F, ty, tx, tz]), spec)


def q3_stop(X, spec, k=1.0):
    """Quad3D.stop (quad3D.py:201-228)."""
    ax, ay, az = -k * X[6], -k * X[7], -k * X[8]
    th_d, ph_d, F = ax / GRAV3, -ay / GRAV3, spec["mass"] * az
    ty = spec["Iy"] * k * (th_d - X[3] - X[9] / k)
    tx = spec["Ix"] * k * (ph_d - X[4] - X[10] / k)
    tz = spec["Iz"] * k * (0 - X[5] - X[11] / k)
    return _q3_alloc(np.array([F, ty, tx, tz]), spec)


def q3_has_stopped(X, tol=0.05):
    return bool(np.linalg.norm(X[6:9]) < tol and np.linalg.norm(X[9:12]) < tol)   # quad3D.py:230-234


def q3_rotate_to(X, ang, spec, k=2.0):
    """Quad3D.rotate_to (quad3D.py:236-257): hover thrust m g (the linear model has no gravity term: the vehicle climbs)."""
    F = spec["mass"] * GRAV3
    ty = spec["Iy"] * k * (0 - X[3] - X[9] / k)
    tx = spec["Ix"] * k * (0 - X[4] - X[10] / k)
    tz = spec["Iz"] * k * (ang - X[5] - X[11] / k)
    return _q3_alloc(np.array([F, ty, tx, tz]), spec)


def default_spec(model):
    if model == "VTOL2D":
        from . import mpc_vtol as OV
        return OV.default_spec()
    if model == "Quad2D":
        return dict(R.default_spec(R.MODEL_QUAD2D))
    s = dict(mass=3.0, Ix=0.5, Iy=0.5, Iz=0.5, L=0.3, nu=0.1, u_max=10.0, u_min=-10.0, radius=0.25)   # quad3D.py:50-61
    return s


class QuadTrackingOracle:
    """Single-agent closed loop (mpc_cbf position controller = this repo's oracle NLP solver)."""

    def __init__(self, model, X0, spec=None, dt=0.05, obs=None, num_constraints=10, enable_rotation=True, horizon=10, solve_fn=None):
        assert model in ("Quad2D", "Quad3D", "VTOL2D")
        self.model, self.q3, self.vt = model, model == "Quad3D", model == "VTOL2D"
        if self.vt and horizon == 10:
            horizon = 30                                       # mpc_cbf.py:41
        self.spec = default_spec(model)
        self.spec.update(spec or {})
        self.spec.setdefault("exploration", False)
        X0 = np.asarray(X0, dtype=float).reshape(-1)
        if self.vt:                                           # tracking.py:94-99
            self.X = np.array([X0[0], X0[1], 0.0, 5.0, 0.0, 0.0]) if X0.shape[0] in (2, 3) else X0.copy()
        elif not self.q3:                                     # tracking.py:80-84
            self.X = np.array([X0[0], X0[1], 0, 0, 0, 0.0]) if X0.shape[0] in (2, 3) else X0.copy()
        else:                                                 # tracking.py:85-93
            X = np.zeros(12)
            if X0.shape[0] == 2:
                X[:2] = X0
            elif X0.shape[0] == 3:
                X[0], X[1], X[5] = X0
            elif X0.shape[0] == 4:
                X[0], X[1], X[2], X[5] = X0
            else:
                X = X0.copy()
            self.X = X
        self.dt, self.N = dt, horizon
        self.obs = np.zeros((0, 7)) if obs is None else np.array(obs, dtype=float)
        self.K = num_constraints
        self.enable_rotation = enable_rotation
        self.state_machine, self.goal, self.waypoints, self.current_goal_index = "idle", None, None, 0
        self.reached_threshold, self.rotation_threshold = self.spec.get("reached_threshold", 0.3), 0.1
        self.fov_angle = math.radians(float(self.spec.get("fov_angle", 70.0)))
        self.n_pos = 3 if self.q3 else 2
        self.u_prev = np.zeros(4 if (self.q3 or self.vt) else 2)
        self.solve_fn = solve_fn
        self.u_pos = None
        self.mdl = None if self.vt else (OL.quad3d_model(dict(self.spec), dt=dt) if self.q3 else OG.quad2d_model(dict(self.spec), dt=dt))

    @property
    def yaw(self):
        return self.X[5] if self.q3 else self.X[2]

    def set_waypoints(self, waypoints):
        wp = np.array(waypoints, dtype=float)
        if len(wp) >= 2:                                      # filter_waypoints, tracking.py:240-262
            pos = self.X[:3] if self.q3 else self.X[:2]
            aug = np.vstack((pos, wp[:, : self.n_pos]))
            dist = np.linalg.norm(np.diff(aug, axis=0), axis=1)
            wp = aug[np.concatenate(([False], dist >= self.reached_threshold))]
        self.waypoints, self.current_goal_index = wp, 0
        self.goal = self.update_goal()
        if self.goal is not None:
            ang = math.atan2(self.goal[1] - self.X[1], self.goal[0] - self.X[0])
            # is_in_fov (robots/robot.py:854-872): always True for Quad2D ("these dynamics do not have a stop() method")
            if self.q3 and abs(R.angle_normalize(ang - self.yaw)) > self.fov_angle / 2:
                if self.spec["exploration"]:
                    self.state_machine = "rotate"
                else:
                    self.state_machine, self.goal = "stop", None
            else:
                self.state_machine = "track"

    def update_goal(self):
        if self.state_machine == "rotate":
            rg = self.waypoints[self.current_goal_index]
            goal_angle = math.atan2(rg[1] - self.X[1], rg[0] - self.X[0])
            if not self.q3:                                   # Quad2D and VTOL2D skip 'rotate' (tracking.py:512-513)
                self.state_machine = "track"
            if not self.enable_rotation:
                self.state_machine = "track"
            if abs(self.yaw - goal_angle) > self.rotation_threshold:
                return rg[: self.n_pos]
            self.state_machine = "track"
        if self.current_goal_index >= len(self.waypoints):
            return None
        wp = self.waypoints[self.current_goal_index]
        if np.linalg.norm(self.X[:2] - wp[:2]) < self.reached_threshold:
            self.current_goal_index += 1
            if self.current_goal_index >= len(self.waypoints):
                self.state_machine = "idle"
                return None
        return np.array(self.waypoints[self.current_goal_index][0: self.n_pos])

    def nearest(self):
        """get_nearest_unpassed_obs with angle_unpassed = 2 pi: the K nearest centres, ties by index."""
        if len(self.obs) == 0:
            return None
        pool = self.obs
        if self.vt:                                           # angle_unpassed = 1.2 pi about the pitch angle; empty cone: everything
            ang = np.arctan2(pool[:, 1] - self.X[1], pool[:, 0] - self.X[0])
            front = np.abs(R.angle_normalize(ang - self.yaw)) <= 0.6 * np.pi
            if front.any():
                pool = pool[front]
        d = np.hypot(pool[:, 0] - self.X[0], pool[:, 1] - self.X[1])
        order = np.argsort(d, kind="stable")[: self.K]
        return pool[order]

    def _collide(self):
        if is_collide(self.X, self.obs, self.spec["radius"]):
            return True
        return self.vt and (self.X[1] < 0 or abs(self.X[2]) > self.spec["pitch_max"])     # tracking.py:490-495

    def control_step(self):
        has_stopped = q3_has_stopped(self.X) if self.q3 else q2_has_stopped(self.X)      # VTOL2D: the same planar-speed test (vtol2D.py:467-469)
        if self.state_machine == "stop":
            if has_stopped:
                self.state_machine = "rotate" if self.enable_rotation else "track"
                self.goal = self.update_goal()
        else:
            self.goal = self.update_goal()
        near = self.nearest()
        if self.vt:
            u_ref = np.zeros(4)
        elif self.state_machine == "rotate":
            ga = math.atan2(self.goal[1] - self.X[1], self.goal[0] - self.X[0])
            u_ref = q3_rotate_to(self.X, ga, self.spec) if self.q3 else np.array([0.0, 2.0 * R.angle_normalize(ga - self.X[2])])
        elif self.goal is None:
            u_ref = q3_stop(self.X, self.spec) if self.q3 else q2_stop(self.X, self.spec)
        else:
            u_ref = q3_nominal(self.X, self.goal, self.spec) if self.q3 else q2_nominal(self.X, self.goal, self.spec)
        if self.state_machine != "track":                     # mpc_cbf.py:379-381
            u = u_ref
        else:
            obs = np.tile(M.DUMMY_OBS, (self.K, 1))
            if near is not None:
                obs[: len(near)] = near[:, :7]
            if self.solve_fn is not None:
                u = self.solve_fn(self.X, self.u_prev, self.goal, obs)
            elif self.vt:
                from . import mpc_vtol as OV
                u = OV.solve(self.X, self.u_prev, self.goal[:2], obs, N=self.N, spec=dict(self.spec), dt=self.dt)[0]
            elif self.q3:
                u = OL.solve(self.mdl, self.X, self.u_prev, self.goal[:3], obs, N=self.N)[0]
            else:
                u = OG.solve(self.mdl, self.X, self.u_prev, self.goal[:2], obs, N=self.N)[0]
            u = np.asarray(u, dtype=float)
            self.u_prev = u.copy()
        if self._collide():
            return -2
        if self.vt:
            from . import mpc_vtol as OV
            self.X = OV.vt_S(self.X, u, self.spec, self.dt)
        else:
            self.X = q3_step(self.X, u, self.dt, self.spec) if self.q3 else q2_step(self.X, u, self.dt, self.spec)
        self.u_pos = np.asarray(u, dtype=float).reshape(-1)
        if self._collide():
            return -2
        if self.goal is None and self.state_machine != "stop":
            return -1
        return 0
