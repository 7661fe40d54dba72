"""Host-side robot description consumed by the batched controllers.

The reference keeps per-model numpy callbacks in robots/<model>.py and a
dispatcher robots/robot.py:BaseRobot; on the batched path those callbacks run
inside the HIP kernels (csrc/sc_models.hpp), selected by model id.  What stays
on the host is the ``robot_spec`` dict and the three attributes the position
controllers read from the robot object: ``X``, ``dt``, ``robot_radius``.
A reference ``BaseRobot`` can be passed to CBFQP/MPCCBF directly; RobotHandle
is the minimal stand-in when the reference package is not importable.
"""
import math

import numpy as np

from .._lib import MODEL_IDS


VTOL2D_DEFAULTS = dict(mass=11.0, inertia=1.135, S_wing=0.55, rho=1.2682, C_L0=0.23, C_Lalpha=5.61, M=50.0, alpha_0=math.radians(15.0),
                       C_Ldelta_e=0.13, C_D0=0.043, C_Dalpha=0.03, C_Ddelta_e=0.0, C_m0=0.0135, C_malpha=-2.74, C_mdelta_e=-0.99,
                       chord=0.18994, k_front=70.0, k_rear=70.0, k_pusher=60.0, ell_f=0.5, ell_r=0.5, throttle_min=0.0, throttle_max=1.0,
                       elevator_min=-0.5, elevator_max=0.5, v_max=15.0, pitch_max=15.0, descent_speed_max=5.0, radius=0.6)


def complete_robot_spec(robot_spec):
    """Apply the defaults the reference's robot classes ``setdefault`` into robot_spec.

    SingleIntegrator2D / DoubleIntegrator2D: see below ; DynamicUnicycle2D:
    robots/dynamic_unicycle2D.py:34-40 ; KinematicBicycle2D
    family: robots/kinematic_bicycle2D.py:42-53 ; radius: robots/robot.py:49.
    Mutates and returns the dict, like the reference does.
    """
    model = robot_spec.setdefault("model", "DynamicUnicycle2D")
    if model == "Manipulator2D":                    # robots/manipulator2D.py:21-22 ; radius robots/robot.py:49
        robot_spec.setdefault("w_max", 2.0)
        robot_spec.setdefault("Kp", 3.0)
        robot_spec.setdefault("radius", 0.25)
        return robot_spec
    if model == "Quad3D":                           # robots/quad3D.py:50-59 (MPC-CBF only)
        for k, v in (("mass", 3.0), ("Ix", 0.5), ("Iy", 0.5), ("Iz", 0.5), ("L", 0.3), ("nu", 0.1), ("u_max", 10.0),
                     ("u_min", -10.0), ("radius", 0.25)):
            robot_spec.setdefault(k, v)
        return robot_spec
    if model == "VTOL2D":                           # robots/vtol2D.py:56-111 (MPC-CBF only)
        for k, v in VTOL2D_DEFAULTS.items():
            robot_spec.setdefault(k, v)
        return robot_spec
    if model not in MODEL_IDS:
        raise ValueError(f"model {model!r} is not supported by the batched engine (supported: {sorted(MODEL_IDS)})")
    if model == "SingleIntegrator2D":               # robots/single_integrator2D.py:40-43
        robot_spec.setdefault("v_max", 1.0)
        robot_spec.setdefault("w_max", 0.5)
        robot_spec.setdefault("radius", 0.25)
    elif model == "DoubleIntegrator2D":             # robots/double_integrator2D.py:38-44
        robot_spec.setdefault("a_max", 1.0)
        robot_spec.setdefault("v_max", 1.0)
        robot_spec.setdefault("ax_max", robot_spec["a_max"])
        robot_spec.setdefault("ay_max", robot_spec["a_max"])
        robot_spec.setdefault("w_max", 0.5)
        robot_spec.setdefault("radius", 0.25)
    elif model == "Unicycle2D":                     # robots/unicycle2D.py:39-40
        robot_spec.setdefault("v_max", 1.0)
        robot_spec.setdefault("w_max", 0.5)
        robot_spec.setdefault("radius", 0.25)
    elif model == "Quad2D":                         # robots/quad2D.py:41-44
        robot_spec.setdefault("mass", 1.0)
        robot_spec.setdefault("inertia", 0.01)
        robot_spec.setdefault("f_min", 1.0)
        robot_spec.setdefault("f_max", 10.0)
        robot_spec.setdefault("radius", 0.25)
    elif model == "DynamicUnicycle2D":
        robot_spec.setdefault("a_max", 0.5)
        robot_spec.setdefault("w_max", 0.5)
        robot_spec.setdefault("v_max", 1.0)
        robot_spec.setdefault("radius", 0.25)
    else:
        robot_spec.setdefault("wheel_base", 0.4)
        robot_spec.setdefault("body_width", 0.3)
        robot_spec.setdefault("radius", 0.3)
        robot_spec.setdefault("front_ax_dist", 0.2)
        robot_spec.setdefault("rear_ax_dist", 0.2)
        robot_spec.setdefault("v_max", 3.5)
        robot_spec.setdefault("a_max", 5.0)
        robot_spec.setdefault("delta_max", math.radians(32))
        robot_spec.setdefault("beta_max", math.atan((robot_spec["rear_ax_dist"] / robot_spec["wheel_base"])
                                                    * math.tan(robot_spec["delta_max"])))
        robot_spec.setdefault("v_min", 0.2)
    return robot_spec


class RobotHandle:
    """Minimal robot object: what CBFQP / MPCCBF read from ``robot`` (robots/robot.py:38-50)."""

    def __init__(self, X0, robot_spec, dt=0.05):
        self.robot_spec = complete_robot_spec(robot_spec)
        self.X = np.asarray(X0, dtype=np.float64).reshape(-1, 1)
        self.dt = float(dt)
        self.robot_radius = float(self.robot_spec["radius"])
