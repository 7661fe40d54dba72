"""Host-side description of the reference's LINEAR robot models for the MPC-CBF kernel (csrc/mpc_lin.hip).

The reference's MPCCBF builds its prediction model symbolically from ``robot.f_casadi`` / ``robot.g_casadi``
(position_control/mpc_cbf.py:135-141) and its CBF rows from ``robot.agent_barrier_dt`` (:312-315), which steps the state
with the robot's own ``step``.  For the models whose f(x) = A x and g(x) = B are constant both are matrices:

  SingleIntegrator2D   robots/single_integrator2D.py:45-66     A = 0, B = I, step = Euler
  Quad3D               robots/quad3D.py:70-158                  12-state linearised quadrotor, step = RK4 (u held)

``linear_model(robot_spec, dt)`` returns Ae, Be (Euler prediction), As, Bs (the barrier's one-step map), the MPC weights
and gains of mpc_cbf.py for that model, and the input box.
"""
import numpy as np

GRAVITY_Q3D = 9.8                                             # quad3D.py:70
LINEAR_MODELS = ("SingleIntegrator2D", "Quad3D")


def quad3d_matrices(spec):
    """A, B = B1 B2 of quad3D.py:70-96."""
    m, Ix, Iy, Iz, L, nu_ = (float(spec[k]) for k in ("mass", "Ix", "Iy", "Iz", "L", "nu"))
    B2 = np.array([[1, 1, 1, 1], [0, L, 0, -L], [L, 0, -L, 0], [nu_, -nu_, nu_, -nu_]], dtype=np.float64)
    A = np.zeros((12, 12))
    for i in range(6):
        A[i, 6 + i] = 1.0
    A[6, 3] = GRAVITY_Q3D
    A[7, 4] = -GRAVITY_Q3D
    B1 = np.zeros((12, 4))
    B1[8, 0] = 1.0 / m
    B1[9, 1] = 1.0 / Iy
    B1[10, 2] = 1.0 / Ix
    B1[11, 3] = 1.0 / Iz
    return A, B1 @ B2


def rk4_maps(A, B, dt):
    """x+ = As x + Bs u for one RK4 step of x' = A x + B u with u held (quad3D.py:140-146)."""
    I = np.eye(A.shape[0])
    k1x, k1u = A, B
    k2x, k2u = A @ (I + dt / 2 * k1x), A @ (dt / 2 * k1u) + B
    k3x, k3u = A @ (I + dt / 2 * k2x), A @ (dt / 2 * k2u) + B
    k4x, k4u = A @ (I + dt * k3x), A @ (dt * k3u) + B
    return I + dt / 6 * (k1x + 2 * k2x + 2 * k3x + k4x), dt / 6 * (k1u + 2 * k2u + 2 * k3u + k4u)


def linear_model(robot_spec, dt):
    model = robot_spec["model"]
    if model == "SingleIntegrator2D":
        A, B = np.zeros((2, 2)), np.eye(2)
        Ae, Be = np.eye(2) + dt * A, dt * B
        v = float(robot_spec["v_max"])
        return dict(nx=2, nu=2, ng=2, Ae=Ae, Be=Be, As=Ae.copy(), Bs=Be.copy(),
                    Q=np.diag([50.0, 50.0]), R=np.array([5.0, 5.0]),           # mpc_cbf.py:19-21
                    cbf_param={"alpha": 0.05},                                 # mpc_cbf.py:48-50
                    u_lo=np.array([-v, -v]), u_hi=np.array([v, v]),            # mpc_cbf.py:183-187
                    circles_only=False)
    if model == "Quad3D":
        A, B = quad3d_matrices(robot_spec)
        As, Bs = rk4_maps(A, B, dt)
        lo, hi = float(robot_spec["u_min"]), float(robot_spec["u_max"])
        return dict(nx=12, nu=4, ng=3, Ae=np.eye(12) + dt * A, Be=dt * B, As=As, Bs=Bs,
                    Q=np.diag([30.0, 30, 5, 20, 20, 1, 10, 10, 10, 20, 20, 1]), R=np.array([1.0, 1, 1, 1]),   # mpc_cbf.py:37-39
                    cbf_param={"alpha": 0.15},                                 # mpc_cbf.py:77-78
                    u_lo=np.full(4, lo), u_hi=np.full(4, hi),                  # mpc_cbf.py:219-223
                    circles_only=True)                                         # quad3D.py:283-291
    raise NotImplementedError(model)
