"""Float64 numpy restatement of the robot callbacks on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function cites the reference file:line it follows (paths relative to the
reference checkout).  States are flat float64 arrays ``X = [x, y, theta, v]``
and inputs ``U = [a, omega]`` (DynamicUnicycle2D) or ``U = [a, beta]``
(KinematicBicycle2D family); obstacles are 7-wide rows
``[x, y, r, vx, vy, -, flag]`` (circle, flag 0) or
``[ox, oy, a, b, e, theta, 1]`` (superellipsoid).
"""
import math

import numpy as np

MODEL_DU = 0          # DynamicUnicycle2D            (rel-deg 2 HOCBF)
MODEL_KB = 1          # KinematicBicycle2D           (rel-deg 2 HOCBF)
MODEL_KB_C3BF = 2     # KinematicBicycle2D_C3BF      (rel-deg 1 collision cone)
MODEL_KB_DPCBF = 3    # KinematicBicycle2D_DPCBF     (rel-deg 1 dynamic parabola)

MODEL_SI = 4          # SingleIntegrator2D           (rel-deg 1 distance barrier), X = [x, y, -, -], U = [vx, vy]
MODEL_DI = 5          # DoubleIntegrator2D           (rel-deg 2 HOCBF),            X = [x, y, vx, vy], U = [ax, ay]

MODEL_QUAD2D = 6      # Quad2D (rel-deg 2 HOCBF), X = [x, z, theta, vx, vz, theta_dot], U = [F_right, F_left]
GRAVITY = 9.81        # robots/quad2D.py:47
MODEL_UNI = 7         # Unicycle2D (rel-deg 1, distance barrier minus sigma(s)), X = [x, y, theta, -], U = [v, omega]
UNI_K1, UNI_K2 = 0.5, 1.8   # robots/unicycle2D.py:36-37

MODEL_NAMES = {
    "DynamicUnicycle2D": MODEL_DU,
    "KinematicBicycle2D": MODEL_KB,
    "KinematicBicycle2D_C3BF": MODEL_KB_C3BF,
    "KinematicBicycle2D_DPCBF": MODEL_KB_DPCBF,
    "SingleIntegrator2D": MODEL_SI,
    "DoubleIntegrator2D": MODEL_DI,
    "Quad2D": MODEL_QUAD2D,
    "Unicycle2D": MODEL_UNI,
}

REL_DEG2 = (MODEL_DU, MODEL_KB, MODEL_DI, MODEL_QUAD2D)


def angle_normalize(x):
    """robots/dynamic_unicycle2D.py:13-16: Python ``%`` wrap into [-pi, pi)."""
    return ((x + math.pi) % (2.0 * math.pi)) - math.pi


def default_spec(model):
    """Defaults the reference's robot classes ``setdefault`` into robot_spec.

    DU: robots/dynamic_unicycle2D.py:36-40 ; KB: robots/kinematic_bicycle2D.py:42-53 ;
    radius default robots/robot.py:49.
    """
    if model == MODEL_DU:
        return dict(a_max=0.5, w_max=0.5, v_max=1.0, radius=0.25)
    if model == MODEL_SI:                                  # robots/single_integrator2D.py:40-43
        return dict(v_max=1.0, w_max=0.5, radius=0.25)
    if model == MODEL_DI:                                  # robots/double_integrator2D.py:38-44
        return dict(a_max=1.0, v_max=1.0, w_max=0.5, radius=0.25)
    if model == MODEL_QUAD2D:                              # robots/quad2D.py:41-44
        return dict(mass=1.0, inertia=0.01, f_min=1.0, f_max=10.0, radius=0.25)
    if model == MODEL_UNI:                                 # robots/unicycle2D.py:39-40
        return dict(v_max=1.0, w_max=0.5, radius=0.25)
    rear, wb = 0.2, 0.4
    delta_max = np.deg2rad(32)
    return dict(a_max=5.0, v_max=3.5, v_min=0.2, radius=0.3, rear_ax_dist=rear,
                wheel_base=wb, delta_max=delta_max,
                beta_max=float(np.arctan((rear / wb) * np.tan(delta_max))))


# --------------------------------------------------------------------------
# dynamics
# --------------------------------------------------------------------------
def f(model, X, spec=None):
    """Drift. DU robots/dynamic_unicycle2D.py:42-54 ; KB robots/kinematic_bicycle2D.py:75-91 ;
    SI robots/single_integrator2D.py:45-55 (zero) ; DI robots/double_integrator2D.py:46-59 ([vx, vy, 0, 0])."""
    if model in (MODEL_SI, MODEL_UNI):                     # unicycle2D.py:42-50: zero drift
        return np.zeros(4)
    if model == MODEL_DI:
        return np.array([X[2], X[3], 0.0, 0.0])
    if model == MODEL_QUAD2D:                              # robots/quad2D.py:46-58
        return np.array([X[3], X[4], X[5], 0.0, -GRAVITY, 0.0])
    th, v = X[2], X[3]
    return np.array([v * math.cos(th), v * math.sin(th), 0.0, 0.0])


def g(model, X, spec=None):
    """Input matrix. DU :64-73 (constant) ; KB robots/kinematic_bicycle2D.py:93-111 (state dependent)."""
    if model == MODEL_DU:
        return np.array([[0.0, 0.0], [0.0, 0.0], [0.0, 1.0], [1.0, 0.0]])
    if model == MODEL_SI:                                  # single_integrator2D.py:57-65 (identity on x, y)
        return np.array([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0], [0.0, 0.0]])
    if model == MODEL_UNI:                                 # unicycle2D.py:52-62 (3 states, padded to 4)
        return np.array([[math.cos(X[2]), 0.0], [math.sin(X[2]), 0.0], [0.0, 1.0], [0.0, 0.0]])
    if model == MODEL_DI:                                  # double_integrator2D.py:69-79
        return np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    if model == MODEL_QUAD2D:                              # robots/quad2D.py:68-81
        m, I, r = spec["mass"], spec["inertia"], spec["radius"]
        sn, cs = math.sin(X[2]), math.cos(X[2])
        return np.array([[0, 0, 0, -sn / m, cs / m, r / I], [0, 0, 0, -sn / m, cs / m, -r / I]], dtype=np.float64).T
    th, v = X[2], X[3]
    L_r = spec["rear_ax_dist"]
    return np.array([[0.0, -v * math.sin(th)],
                     [0.0, v * math.cos(th)],
                     [0.0, v / L_r],
                     [1.0, 0.0]])


def df_dx(model, X):
    """Jacobian of f. DU :56-62 ; KB :67-73 (identical expressions)."""
    th, v = X[2], X[3]
    c, s = math.cos(th), math.sin(th)
    J = np.zeros((4, 4))
    J[0, 2], J[0, 3] = -v * s, c
    J[1, 2], J[1, 3] = v * c, s
    return J


def step(model, X, U, dt, spec=None):
    """Euler step + heading wrap. DU :75-78 ; KB :113-123 (also clips v to [v_min, v_max])."""
    Xn = np.asarray(X, dtype=np.float64) + (f(model, X, spec) + g(model, X, spec) @ np.asarray(U, dtype=np.float64)) * dt
    if model == MODEL_SI:                                  # single_integrator2D.py:67-69
        return Xn
    if model in (MODEL_QUAD2D, MODEL_UNI):                 # robots/quad2D.py:83-86, unicycle2D.py:64-67
        Xn[2] = angle_normalize(Xn[2])
        return Xn
    if model == MODEL_DI:                                  # double_integrator2D.py:81-108: speed saturation
        vm = math.sqrt(Xn[2] ** 2 + Xn[3] ** 2)
        if spec.get("v_max") is not None and vm > spec["v_max"]:
            Xn[2] *= spec["v_max"] / vm
            Xn[3] *= spec["v_max"] / vm
        return Xn
    Xn[2] = angle_normalize(Xn[2])
    if model != MODEL_DU:
        Xn[3] = min(max(Xn[3], spec["v_min"]), spec["v_max"])
    return Xn


def nominal_input(model, X, goal, spec, d_min=0.05):
    """Go-to-goal reference input.

    DU robots/dynamic_unicycle2D.py:80-104 (gains k_omega=2, k_a=1, k_v=1,
    overridable through ``nominal_k_*`` keys; output NOT clipped).
    KB robots/kinematic_bicycle2D.py:125-147; its own defaults (.5, 1.5, .5) are
    never used on the control_step path because BaseRobot.nominal_input
    (robots/robot.py:401-408) forwards k_omega=2, k_a=1, k_v=1 positionally.
    """
    if model in (MODEL_SI, MODEL_DI):
        # SI robots/single_integrator2D.py:75-93 ; DI robots/double_integrator2D.py:114-141
        # (BaseRobot forwards d_min, k_v(, k_a) = .05, 1(, 1): robots/robot.py:402-403,408-409)
        k_v = spec.get("nominal_k_v", 1.0) if model == MODEL_DI else 1.0
        k_a = spec.get("nominal_k_a", 1.0)
        pe = np.array([goal[0] - X[0], goal[1] - X[1]])
        pe = np.sign(pe) * np.maximum(np.abs(pe) - d_min, 0.0)
        v_des = k_v * pe
        vm = np.linalg.norm(v_des)
        if vm > spec["v_max"]:
            v_des = v_des * spec["v_max"] / vm
        if model == MODEL_SI:
            return v_des
        a = k_a * (v_des - np.array([X[2], X[3]]))
        am = np.linalg.norm(a)
        if am > spec["a_max"]:
            a = a * spec["a_max"] / am
        return a
    dxg, dyg = goal[0] - X[0], goal[1] - X[1]
    dist_raw = math.sqrt((X[0] - goal[0]) ** 2 + (X[1] - goal[1]) ** 2)
    theta_d = math.atan2(dyg, dxg)
    err = angle_normalize(theta_d - X[2])
    if model == MODEL_UNI:                                 # unicycle2D.py:69-85 with (d_min, k_omega, k_v) = (.05, 2, 1)
        distance = max(dist_raw - d_min, 0.05)             # forwarded by robots/robot.py:404-405; output not clipped
        v = 0.0 if abs(err) > math.radians(90) else 1.0 * distance * math.cos(err)
        return np.array([v, 2.0 * err])
    if model == MODEL_DU:
        k_omega = spec.get("nominal_k_omega", 2.0)
        k_a = spec.get("nominal_k_a", 1.0)
        k_v = spec.get("nominal_k_v", 1.0)
        distance = max(dist_raw - d_min, 0.0)
        omega = k_omega * err
        if abs(err) > math.radians(90):
            v = 0.0
        else:
            v = min(k_v * distance * math.cos(err), spec["v_max"])
        return np.array([k_a * (v - X[3]), omega])
    k_theta, k_a, k_v = 2.0, 1.0, 1.0
    distance = max(dist_raw - d_min, 0.05)
    delta = min(max(k_theta * err, -spec["delta_max"]), spec["delta_max"])
    beta = math.atan((spec["rear_ax_dist"] / spec["wheel_base"]) * math.tan(delta))
    v_cmd = k_v * distance * max(0.0, math.cos(err))
    v = min(max(v_cmd, spec["v_min"]), spec["v_max"])
    return np.array([k_a * (v - X[3]), beta])


def stop(model, X, spec):
    """DU :106-111 (brake with k_a) ; KB :149-150 (zeros) ; SI :103-106 (zeros) ; DI :151-157 (brake both axes)."""
    if model in (MODEL_SI, MODEL_UNI):                     # unicycle2D.py:87-88
        return np.array([0.0, 0.0])
    if model == MODEL_DI:
        k_a = spec.get("nominal_k_a", 1.0)
        return np.array([k_a * (0.0 - X[2]), k_a * (0.0 - X[3])])
    if model == MODEL_DU:
        return np.array([spec.get("nominal_k_a", 1.0) * (0.0 - X[3]), 0.0])
    return np.array([0.0, 0.0])


def has_stopped(model, X, tol=0.05):
    """DU :113-114 ; KB :152-153 ; Unicycle2D unicycle2D.py:90-92 (always)."""
    if model in (MODEL_UNI, MODEL_SI):                     # single_integrator2D.py:105-107: always
        return True
    if model == MODEL_DI:                                  # double_integrator2D.py:158-159: |(vx, vy)| < tol
        return math.hypot(X[2], X[3]) < tol
    return abs(X[3]) < tol


def rotate_to(model, X, theta_des, k=2.0):
    """DU :116-119 ; KB :155-158."""
    return np.array([0.0, k * angle_normalize(theta_des - X[2])])


# --------------------------------------------------------------------------
# continuous-time barriers
# --------------------------------------------------------------------------
def _hocbf_circle(X, obs, R, beta):
    """h, h_dot, d(h_dot)/dx for the distance barrier (rel-deg 2).

    DU robots/dynamic_unicycle2D.py:136-146 ; KB robots/kinematic_bicycle2D.py:160-173.
    """
    th, v = X[2], X[3]
    c, s = math.cos(th), math.sin(th)
    ex, ey = X[0] - obs[0], X[1] - obs[1]
    d_min = obs[2] + R
    h = math.sqrt(ex * ex + ey * ey) ** 2 - beta * d_min ** 2
    f0, f1 = v * c, v * s
    h_dot = 2.0 * (ex * f0 + ey * f1)
    dhd = np.array([2.0 * f0, 2.0 * f1,
                    2.0 * (ex * (-v * s) + ey * (v * c)),
                    2.0 * (ex * c + ey * s)])
    return h, h_dot, dhd


def _hocbf_superellipsoid(X, obs, R):
    """DU robots/dynamic_unicycle2D.py:148-183 (signed ``**`` as in numpy)."""
    th, v = X[2], X[3]
    c, s = math.cos(th), math.sin(th)
    ox, oy, a, b, e, tho = (np.float64(obs[i]) for i in range(6))
    ct, st = np.cos(tho), np.sin(tho)
    px = ct * (X[0] - ox) + st * (X[1] - oy)
    py = -st * (X[0] - ox) + ct * (X[1] - oy)
    Aa, Bb = a + R, b + R
    with np.errstate(all="ignore"):
        h = (px / Aa) ** e + (py / Bb) ** e - 1.0
        gx = e * px ** (e - 1) / Aa ** e          # d/dpx of (px/Aa)^e
        gy = e * py ** (e - 1) / Bb ** e
        dh_x = gx * ct - gy * st
        dh_y = gx * st + gy * ct
        h_dot = dh_x * v * c + dh_y * v * s
        ca = e * (e - 1) / Aa ** e * px ** (e - 2)
        cb = e * (e - 1) / Bb ** e * py ** (e - 2)
        hxx = ca * ct * ct + cb * st * st
        hxy = (ca - cb) * ct * st
        hyy = ca * st * st + cb * ct * ct
        dhd = np.array([hxx * v * c + hxy * v * s,
                        hxy * v * c + hyy * v * s,
                        dh_x * (-v * s) + dh_y * (v * c),
                        dh_x * c + dh_y * s], dtype=np.float64)
    return float(h), float(h_dot), dhd


def _superellipsoid_terms(X, obs, R):
    """h, dh/dp (2,), d2h/dp2 entries of the superellipsoid barrier (shared by SI / DI / DU formulas)."""
    ox, oy, a, b, e, tho = (np.float64(obs[i]) for i in range(6))
    ct, st = np.cos(tho), np.sin(tho)
    px = ct * (X[0] - ox) + st * (X[1] - oy)
    py = -st * (X[0] - ox) + ct * (X[1] - oy)
    Aa, Bb = a + R, b + R
    with np.errstate(all="ignore"):
        h = (px / Aa) ** e + (py / Bb) ** e - 1.0
        gx = e * px ** (e - 1) / Aa ** e
        gy = e * py ** (e - 1) / Bb ** e
        ca = e * (e - 1) / Aa ** e * px ** (e - 2)
        cb = e * (e - 1) / Bb ** e * py ** (e - 2)
    return (float(h), np.array([gx * ct - gy * st, gx * st + gy * ct]),
            ca * ct * ct + cb * st * st, (ca - cb) * ct * st, ca * st * st + cb * ct * ct)


def _uni_barrier(X, obs, R, beta=1.01):
    """robots/unicycle2D.py:107-128 (rel-deg 1): h = |p - o|^2 - beta d_min^2 - sigma(s), s = (p - o) . heading,
    sigma(s) = k2 (e^(k1-s) - 1) / (e^(k1-s) + 1); the obstacle flag is not looked at.  sigma is evaluated as
    k2 tanh((k1 - s) / 2), the same function without the overflow of e^(k1-s) for obstacles far behind the robot
    (the reference returns NaN there); the reference indexes the obstacle as a column (obs[2][0]), which the 1-D rows
    cbf_qp.py hands over do not support -- the goldens call it with a column."""
    ex, ey = X[0] - obs[0], X[1] - obs[1]
    c, sn = math.cos(X[2]), math.sin(X[2])
    d_min = obs[2] + R
    s = ex * c + ey * sn
    th = math.tanh(0.5 * (UNI_K1 - s))
    sig = UNI_K2 * th
    dsig = -0.5 * UNI_K2 * (1.0 - th * th)
    h = math.sqrt(ex * ex + ey * ey) ** 2 - beta * d_min ** 2 - sig
    return h, np.array([2.0 * ex - dsig * c, 2.0 * ey - dsig * sn, -dsig * (-sn * ex + c * ey), 0.0])


def _si_barrier(X, obs, R, beta=1.01):
    """robots/single_integrator2D.py:119-149 (rel-deg 1): h and dh/dx over the 2 position states (padded to 4)."""
    if obs[-1] == 0:
        ex, ey = X[0] - obs[0], X[1] - obs[1]
        d_min = obs[2] + R
        return math.sqrt(ex * ex + ey * ey) ** 2 - beta * d_min ** 2, np.array([2.0 * ex, 2.0 * ey, 0.0, 0.0])
    if obs[-1] == 1:
        h, dh, _, _, _ = _superellipsoid_terms(X, obs, R)
        return h, np.array([dh[0], dh[1], 0.0, 0.0])
    raise ValueError("SingleIntegrator2D: obstacle flag must be 0 or 1")


def _di_barrier(X, obs, R, beta=1.01):
    """robots/double_integrator2D.py:167-220 (rel-deg 2): h, h_dot, d(h_dot)/dx."""
    vx, vy = X[2], X[3]
    if obs[-1] == 0:
        ex, ey = X[0] - obs[0], X[1] - obs[1]
        d_min = obs[2] + R
        h = math.sqrt(ex * ex + ey * ey) ** 2 - beta * d_min ** 2
        return h, 2.0 * (ex * vx + ey * vy), np.array([2.0 * vx, 2.0 * vy, 2.0 * ex, 2.0 * ey])
    if obs[-1] == 1:
        h, dh, hxx, hxy, hyy = _superellipsoid_terms(X, obs, R)
        return h, float(dh[0] * vx + dh[1] * vy), np.array([hxx * vx + hxy * vy, hxy * vx + hyy * vy, dh[0], dh[1]])
    raise ValueError("DoubleIntegrator2D: obstacle flag must be 0 or 1")


def _c3bf(X, obs, R, beta=1.0):
    """dynamic_env/kinematic_bicycle2D_c3bf.py:15-75 (collision-cone CBF, rel-deg 1)."""
    th, v = X[2], X[3]
    c, s = math.cos(th), math.sin(th)
    ovx, ovy = obs[3], obs[4]
    ego = (obs[2] + R) * beta
    px, py = obs[0] - X[0], obs[1] - X[1]
    vx, vy = ovx - v * c, ovy - v * s
    pm = math.sqrt(px * px + py * py)
    vm = math.sqrt(vx * vx + vy * vy)
    eps = 1e-6
    sq = math.sqrt(max(pm ** 2 - ego ** 2, eps))
    cos_phi = sq / (pm + eps)
    h = (px * vx + py * vy) + pm * vm * cos_phi
    with np.errstate(all="ignore"):
        k = np.float64(sq + eps) / np.float64(vm)
        dh = np.array([-vx - vm * px / (sq + eps),
                       -vy - vm * py / (sq + eps),
                       v * s * px - v * c * py + k * (v * (ovx * s - ovy * c)),
                       -c * px - s * py + k * (v - (ovx * c + ovy * s))], dtype=np.float64)
    return h, dh


def _dpcbf(X, obs, R, s_margin=1.05, k_lambda=0.1, k_mu=0.5):
    """dynamic_env/kinematic_bicycle2D_dpcbf.py:16-84 (dynamic parabolic CBF, rel-deg 1).

    Note the reference's gradient uses the bare gains k_lambda / k_mu (no
    sqrt(s^2-1)/ego_dim factor) while h uses the scaled ones; restated as is.
    """
    th, v = X[2], X[3]
    c, s = math.cos(th), math.sin(th)
    ovx, ovy = obs[3], obs[4]
    ego = (obs[2] + R) * s_margin
    px, py = obs[0] - X[0], obs[1] - X[1]
    vx, vy = ovx - v * c, ovy - v * s
    pm = math.sqrt(px * px + py * py)
    vm = math.sqrt(vx * vx + vy * vy)
    rot = math.atan2(py, px)
    cr, sr = math.cos(rot), math.sin(rot)
    vnx = cr * vx + sr * vy
    vny = -sr * vx + cr * vy
    eps = 1e-6
    d_safe = max(pm ** 2 - ego ** 2, eps)
    sd = math.sqrt(d_safe)
    with np.errstate(all="ignore"):
        vm_ = np.float64(vm)
        pm2 = np.float64(pm) ** 2
        lam = k_lambda * sd / vm_ * math.sqrt(s_margin ** 2 - 1) / ego
        mu = k_mu * sd * math.sqrt(s_margin ** 2 - 1) / ego
        h = vnx + lam * vny ** 2 + mu
        dh = np.array([
            py * vny / pm2 - k_lambda * px * vny ** 2 / vm_ / sd
            - 2 * k_lambda * sd / vm_ * vny * py / pm2 * vnx - k_mu * px / sd,
            -px * vny / pm2 - k_lambda * py * vny ** 2 / vm_ / sd
            + 2 * k_lambda * sd / vm_ * vny * px / pm2 * vnx - k_mu * py / sd,
            -v * math.sin(rot - th)
            - k_lambda * sd * v * (ovx * s - ovy * c) * vny ** 2 / vm_ ** 3
            - 2 * k_lambda * sd * vny * v * math.cos(rot - th) / vm_,
            -math.cos(rot - th)
            - k_lambda * sd / vm_ ** 3 * (v - ovx * c - ovy * s) * vny ** 2
            - 2 * k_lambda * sd * vny * math.sin(rot - th) / vm_], dtype=np.float64)
    return float(h), dh


def agent_barrier(model, X, obs, R):
    """Dispatch mirroring robots/robot.py:435-436 -> <model>.agent_barrier.

    Returns ``(h, h_dot, dh_dot_dx)`` for rel-deg-2 models and ``(h, dh_dx)``
    for rel-deg-1 models.  An obstacle flag other than 0/1 for the DU model
    raises ValueError (the reference returns integer zeros and the caller's
    ``@`` fails, dynamic_unicycle2D.py:133-136).
    """
    if model == MODEL_DU:
        flag = obs[-1]
        if flag == 0:
            return _hocbf_circle(X, obs, R, 1.01)
        if flag == 1:
            return _hocbf_superellipsoid(X, obs, R)
        raise ValueError("DynamicUnicycle2D: obstacle flag must be 0 or 1")
    if model == MODEL_QUAD2D:                              # robots/quad2D.py:166-177 (circle only, no flag test)
        ex, ez = X[0] - obs[0], X[1] - obs[1]
        d_min = obs[2] + R
        h = math.sqrt(ex * ex + ez * ez) ** 2 - 1.01 * d_min ** 2
        return h, 2.0 * (ex * X[3] + ez * X[4]), np.array([2.0 * X[3], 2.0 * X[4], 0.0, 2.0 * ex, 2.0 * ez, 0.0])
    if model == MODEL_SI:
        return _si_barrier(X, obs, R)
    if model == MODEL_UNI:
        return _uni_barrier(X, obs, R)
    if model == MODEL_DI:
        return _di_barrier(X, obs, R)
    if model == MODEL_KB:
        return _hocbf_circle(X, obs, R, 1.1)
    if model == MODEL_KB_C3BF:
        return _c3bf(X, obs, R)
    if model == MODEL_KB_DPCBF:
        return _dpcbf(X, obs, R)
    raise ValueError("unknown model id %r" % (model,))
