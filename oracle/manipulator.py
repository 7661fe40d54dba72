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
