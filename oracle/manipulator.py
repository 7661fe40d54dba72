"""Float64 numpy restatement of the Manipulator2D path (3 joints, joint-velocity inputs).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: robots/manipulator2D.py (kinematic chain, link circles, one CBF row per link circle per obstacle) and the
Manipulator2D branches of position_control/cbf_qp.py (:34-35 alpha = 1.0, :94-104 three inputs with |u| <= w_max,
:130-151 row loop).  States are flat ``X = [theta1, theta2, theta3]``, inputs ``U = [omega1, omega2, omega3]``.
"""
import math

import numpy as np

from .qp import STATUS_INFEASIBLE, STATUS_OPTIMAL, solve_qpn

SCALE = 60.0                                              # manipulator2D.py:17
LINK_LENGTHS = np.array([80, 70, 50]) / SCALE             # manipulator2D.py:18
STEP_LEN = 10.0 / 60.0                                    # manipulator2D.py:133
BETA = 1.3                                                # manipulator2D.py:185 (agent_barrier default)


def default_spec():
    """manipulator2D.py:21-22 ; radius robots/robot.py:49."""
    return dict(w_max=2.0, Kp=3.0, radius=0.25, base_pos=(0.0, 0.0))


def link_steps():
    """Circles per link minus one: ``int(np.ceil(link_dist / step_len))`` evaluated in float64 (manipulator2D.py:143)."""
    return [int(np.ceil(L / STEP_LEN)) for L in LINK_LENGTHS]


def joint_positions(X, base=(0.0, 0.0)):
    """Base, joint 1, joint 2, end effector (manipulator2D.py:52-60)."""
    P = [np.array(base, dtype=np.float64)]
    ang = 0.0
    for i in range(3):
        ang += X[i]
        P.append(P[-1] + LINK_LENGTHS[i] * np.array([math.cos(ang), math.sin(ang)]))
    return P


def end_effector(X, base=(0.0, 0.0)):
    """manipulator2D.py:42-50 (accumulates x and y separately from the base)."""
    x, y = float(base[0]), float(base[1])
    ang = 0.0
    for i in range(3):
        ang += X[i]
        x += LINK_LENGTHS[i] * math.cos(ang)
        y += LINK_LENGTHS[i] * math.sin(ang)
    return np.array([x, y])


def jacobian(X):
    """End-effector Jacobian (2x3), manipulator2D.py:62-108: column i = sum over links k >= i of
    l_k (-sin, cos)(theta_1 + .. + theta_k)."""
    J = np.zeros((2, 3))
    for i in range(3):
        ang = 0.0
        for k in range(i):
            ang += X[k]
        jx = jy = 0.0
        for k in range(i, 3):
            ang += X[k]
            jx -= LINK_LENGTHS[k] * math.sin(ang)
            jy += LINK_LENGTHS[k] * math.cos(ang)
        J[0, i], J[1, i] = jx, jy
    return J


def f(X):
    return np.zeros(3)                                     # manipulator2D.py:26-30


def g(X):
    return np.eye(3)                                       # manipulator2D.py:32-36


def step(X, U, dt):
    return np.asarray(X, dtype=np.float64) + np.asarray(U, dtype=np.float64) * dt     # manipulator2D.py:38-41 (no wrap)


def nominal_input(X, goal, spec, base=(0.0, 0.0)):
    """Jacobian-transpose control clipped to w_max (manipulator2D.py:110-127)."""
    err = np.asarray(goal, dtype=np.float64)[:2] - end_effector(X, base)
    w = jacobian(X).T @ (spec["Kp"] * err)
    return np.clip(w, -spec["w_max"], spec["w_max"])


def link_circles(X, base=(0.0, 0.0)):
    """Centres and link index of the discretised links (manipulator2D.py:129-152): per link ``ns + 1`` points at
    ``p_start + (j / ns) (dx, dy)``."""
    out = []
    p_start = np.array(base, dtype=np.float64)
    ang = 0.0
    for i in range(3):
        ang += X[i]
        d = np.array([LINK_LENGTHS[i] * math.cos(ang), LINK_LENGTHS[i] * math.sin(ang)])
        p_end = p_start + d
        ns = int(np.ceil(LINK_LENGTHS[i] / STEP_LEN))
        for j in range(ns + 1):
            t = j / ns
            out.append((p_start + t * d, i))
        p_start = p_end
    return out


def point_jacobian(X, pt, link_idx, base=(0.0, 0.0)):
    """manipulator2D.py:154-182: J[:, k] = z x (pt - P_k) for k <= link_idx, zero beyond."""
    J = np.zeros((2, 3))
    P = [np.array(base, dtype=np.float64)]
    ang = 0.0
    for i in range(link_idx + 1):
        if i > 0:
            ang += X[i - 1]
            P.append(P[-1] + LINK_LENGTHS[i - 1] * np.array([math.cos(ang), math.sin(ang)]))
    for k in range(link_idx + 1):
        J[0, k] = -(pt[1] - P[k][1])
        J[1, k] = pt[0] - P[k][0]
    return J


def agent_barrier(X, obs, R, beta=BETA, base=(0.0, 0.0)):
    """manipulator2D.py:185-224: one (h, dh/dq) per link circle: h = |c - o|^2 - beta (R + r)^2, dh = 2 (c - o) J_c."""
    hs, dhs = [], []
    for c, li in link_circles(X, base):
        dx, dy = c[0] - obs[0], c[1] - obs[1]
        d_min = R + obs[2]
        hs.append(dx * dx + dy * dy - beta * d_min ** 2)
        dhs.append(2.0 * np.array([dx, dy]) @ point_jacobian(X, c, li, base))
    return hs, dhs


def assemble_rows(X, obs_list, spec, alpha=1.0, num_rows=150, dt=0.05, cbf_mode="cbf", base=(0.0, 0.0)):
    """cbf_qp.py:110-151: zeroed A1 (num_rows, 3) / b1; obstacles in order, circles in order, stop at num_rows."""
    A = np.zeros((num_rows, 3))
    b = np.zeros(num_rows)
    hv = np.full(num_rows, np.nan)
    row = 0
    for obs in obs_list:
        if obs is None:
            continue
        if row >= num_rows:
            break
        hs, dhs = agent_barrier(X, np.asarray(obs, dtype=np.float64), spec["radius"], base=base)
        for h, dh in zip(hs, dhs):
            if row >= num_rows:
                break
            A[row] = dh                                    # g = I, f = 0
            b[row] = h / dt if cbf_mode == "hard" else alpha * h
            hv[row] = h
            row += 1
    return A, b, hv


def solve(X, u_ref, obs_list, spec, alpha=1.0, num_rows=150, dt=0.05, cbf_mode="cbf", base=(0.0, 0.0)):
    """One Manipulator2D CBF-QP solve; dict(u, status, h, A, b).  ``obs_list is None`` -> u_ref (cbf_qp.py:113-118)."""
    u_ref = np.asarray(u_ref, dtype=np.float64).reshape(3)
    if obs_list is None:
        return dict(u=u_ref.copy(), status=STATUS_OPTIMAL, h=np.full(num_rows, np.nan), A=np.zeros((num_rows, 3)), b=np.zeros(num_rows))
    A, b, hv = assemble_rows(X, obs_list, spec, alpha, num_rows, dt, cbf_mode, base)
    w = spec["w_max"]
    Gb = np.vstack([np.eye(3), -np.eye(3)])
    u, status = solve_qpn(np.vstack([A, Gb]), np.concatenate([b, np.full(6, w)]), u_ref)
    return dict(u=u, status=status, h=hv, A=A, b=b)


class ArmTrackingOracle:
    """LocalTrackingController with a Manipulator2D robot, single arm (tracking.py:197-249 set_waypoints / filter_waypoints,
    :263-268 goal_reached on the end effector, :497-535 update_goal, :559-668 control_step).  The yaw of the robot object
    stays 0 (robots/robot.py:139-148) and its "position" is the fixed base (:354-356), which is what the obstacle ranking,
    the field-of-view test and the collision test see."""

    def __init__(self, q0, spec, base=(0.0, 0.0), dt=0.05, obs=None, num_constraints=150, enable_rotation=True, alpha=1.0):
        self.spec = dict(default_spec()); self.spec.update(spec)
        self.base = np.asarray(base, dtype=np.float64)
        self.dt = dt
        self.X = np.asarray(q0, dtype=np.float64).reshape(-1).copy()
        self.obs = np.zeros((0, 7)) if obs is None else np.array(obs, dtype=np.float64)
        self.num_constraints = num_constraints
        self.enable_rotation = enable_rotation
        self.alpha = alpha
        self.reached_threshold = self.spec.get("reached_threshold", 0.3)
        self.rotation_threshold = 0.1
        self.fov_angle = math.radians(float(self.spec.get("fov_angle", 70.0)))
        self.state_machine = "idle"
        self.current_goal_index = 0
        self.goal = None
        self.status = STATUS_OPTIMAL
        self.u_pos = None

    def set_waypoints(self, waypoints):
        wp = np.array(waypoints, dtype=np.float64)
        if len(wp) >= 2:                                    # filter_waypoints with the end effector as the robot position
            aug = np.vstack((end_effector(self.X, self.base), wp[:, :2]))
            dist = np.linalg.norm(np.diff(aug, axis=0), axis=1)
            wp = aug[np.concatenate(([False], dist >= self.reached_threshold))]
        self.waypoints = wp
        self.current_goal_index = 0
        self.goal = self.update_goal()
        if self.goal is not None:
            ang = math.atan2(self.goal[1] - self.base[1], self.goal[0] - self.base[0])
            wrapped = ((ang - 0.0 + math.pi) % (2.0 * math.pi)) - math.pi
            if abs(wrapped) <= self.fov_angle / 2:
                self.state_machine = "track"
            else:                                            # 'exploration' is False: stop first (tracking.py:221-224)
                self.state_machine = "stop"
                self.goal = None

    def update_goal(self):
        if self.state_machine == "rotate":
            rg = self.waypoints[self.current_goal_index]
            goal_angle = math.atan2(rg[1] - self.X[1], rg[0] - self.X[0])      # joint angles where positions are meant
            self.state_machine = "track"                     # Manipulator2D skips 'rotate' (tracking.py:512-513)
            if abs(0.0 - goal_angle) > self.rotation_threshold:
                return rg[:2]
        if self.current_goal_index >= len(self.waypoints):
            return None
        wp = self.waypoints[self.current_goal_index]
        if np.linalg.norm(end_effector(self.X, self.base) - wp[:2]) < self.reached_threshold:
            self.current_goal_index += 1
            if self.current_goal_index >= len(self.waypoints):
                self.state_machine = "idle"
                return None
        return np.array(self.waypoints[self.current_goal_index][0:2])

    def ranked_obstacles(self):
        if len(self.obs) == 0:
            return None
        d = np.linalg.norm(self.obs[:, :2] - self.base[None, :], axis=1)
        return self.obs[np.argsort(d)[: self.num_constraints]]

    def control_step(self):
        if self.state_machine == "stop":                    # has_stopped() is always True for the arm
            self.state_machine = "rotate" if self.enable_rotation else "track"
            self.goal = self.update_goal()
        else:
            self.goal = self.update_goal()
        near = self.ranked_obstacles()
        u_ref = np.zeros(3) if self.goal is None else nominal_input(self.X, self.goal, self.spec, self.base)
        r = solve(self.X, u_ref, None if near is None else list(near), self.spec, self.alpha, self.num_constraints, self.dt,
                  "cbf", self.base)
        self.status = r["status"]
        collide = bool(len(self.obs)) and bool(np.any(np.linalg.norm(self.obs[:, :2] - self.base[None, :], axis=1)
                                                      < self.obs[:, 2] + self.spec["radius"]))
        if self.status != STATUS_OPTIMAL or collide:
            return -2
        self.X = step(self.X, r["u"], self.dt)
        self.u_pos = r["u"]
        if self.goal is None and self.state_machine != "stop":
            return -1
        return 0
