"""Float64 restatement of the control_step data flow around the solve.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows tracking.py (LocalTrackingController) for the pieces that feed or
consume the CBF-QP / MPC-CBF solve: waypoint filtering (:228-249), goal /
state machine (:497-535, :559-577), obstacle selection (:345-403), the
nominal-input choice (:589-604), the collision check (:445-495), the robot
step (:637) and the return code (:666-668); plus the moving-obstacle variant
dynamic_env/main.py:54-58,126-236.  Sensing / rendering are out of scope.
"""
import math

import numpy as np

from . import cbf_qp, robots as R
from .qp import STATUS_OPTIMAL


def get_nearest_unpassed_obs(model, all_obs, pos, yaw, obs_num):
    """tracking.py:345-403.  Returns (k<=obs_num, 7) array or None."""
    if all_obs is None or len(all_obs) == 0:
        return None
    all_obs = np.asarray(all_obs, dtype=np.float64)
    if all_obs.ndim == 1:
        all_obs = all_obs.reshape(1, -1)
    angle_unpassed = math.pi * 1.2 if model == R.MODEL_DU else math.pi * 2.0   # :352-357
    keep = []
    for o in all_obs:
        ang = math.atan2(o[1] - pos[1], o[0] - pos[0])
        if abs(R.angle_normalize(ang - yaw)) <= angle_unpassed / 2:
            keep.append(o)
    cand = np.array(keep) if len(keep) else all_obs
    d = np.linalg.norm(cand[:, :2] - np.asarray(pos)[None, :], axis=1)
    idx = np.argsort(d)[:obs_num]          # numpy default (quicksort/introsort) as the reference
    return cand[idx]


def is_collide(X, obs_table, radius):
    """Known-obstacle collision test, tracking.py:445-495 (circle and superellipsoid)."""
    if obs_table is None:
        return False
    for o in np.atleast_2d(obs_table):
        if o.shape[0] < 3:
            continue
        superell = o.shape[0] >= 7 and np.isclose(o[6], 1.0) and o[4] >= 2.0   # :428-443
        if not superell:
            if math.hypot(X[0] - o[0], X[1] - o[1]) < o[2] + radius:
                return True
        else:
            ct, st = math.cos(o[5]), math.sin(o[5])
            px = ct * (X[0] - o[0]) + st * (X[1] - o[1])
            py = -st * (X[0] - o[0]) + ct * (X[1] - o[1])
            with np.errstate(all="ignore"):
                h = np.float64(px / (o[2] + radius)) ** o[4] + np.float64(py / (o[3] + radius)) ** o[4] - 1
            if h <= 0:
                return True
    return False


class TrackingOracle:
    """Single-agent closed loop with the oracle solver behind the boundary."""

    def __init__(self, model, X0, spec, dt=0.05, obs=None, num_constraints=10,
                 enable_rotation=True, dyn_obs=False, solve_fn=None, cbf_param=None, yaw0=None):
        self.model = model
        # the integrators keep their heading outside the state (robots/robot.py:66-72: X0 = [x, y, (vx, vy,) yaw]); without
        # an attitude controller it stays at its initial value
        self.integrator = model in (R.MODEL_SI, R.MODEL_DI)
        if self.integrator and enable_rotation:
            raise ValueError("the integrators' rotate state needs the attitude controllers (out of scope)")
        self.yaw = float(yaw0) if yaw0 is not None else 0.0
        self.spec = dict(R.default_spec(model))
        self.spec.update(spec)
        self.spec.setdefault("exploration", False)
        self.dt = dt
        self.X = np.asarray(X0, dtype=np.float64).reshape(-1).copy()
        if self.X.shape[0] == 3:                       # tracking.py:66-68
            self.X = np.append(self.X, 0.0)
        self.obs = np.zeros((0, 7)) if obs is None else np.array(obs, dtype=np.float64)
        self.num_constraints = num_constraints
        self.enable_rotation = enable_rotation
        self.dyn_obs = dyn_obs
        self.state_machine = "idle"
        self.rotation_threshold = 0.1
        self.reached_threshold = self.spec.get("reached_threshold", 0.3)
        self.current_goal_index = 0
        self.goal = None
        self.waypoints = None
        self.fov_angle = math.radians(float(self.spec.get("fov_angle", 70.0)))
        self.cbf_param = cbf_param or cbf_qp.default_cbf_param(model)
        self.solve_fn = solve_fn          # (X, control_ref, obs) -> (u or None, status)
        self.status = STATUS_OPTIMAL
        self.u_pos = None
        self.nearest_multi_obs = None

    # -- waypoint / goal logic ------------------------------------------------
    def set_waypoints(self, waypoints):
        wp = np.array(waypoints, dtype=np.float64)
        if len(wp) >= 2:                                 # tracking.py:228-249
            aug = np.vstack((self.X[:2], wp[:, :2]))
            dist = np.linalg.norm(np.diff(aug, axis=0), axis=1)
            mask = np.concatenate(([False], dist >= self.reached_threshold))
            wp = aug[mask]
        self.waypoints = wp
        self.current_goal_index = 0
        self.goal = self.update_goal()
        if self.goal is not None:                        # tracking.py:214-226
            ang = math.atan2(self.goal[1] - self.X[1], self.goal[0] - self.X[0])
            yaw = self.yaw if self.integrator else self.X[2]
            in_fov = abs(R.angle_normalize(ang - yaw)) <= self.fov_angle / 2   # robot.py:854-872
            if not in_fov:
                if self.spec["exploration"]:
                    self.state_machine = "rotate"
                else:
                    self.state_machine = "stop"
                    self.goal = None
            else:
                self.state_machine = "track"

    def update_goal(self):
        """tracking.py:497-535."""
        if self.state_machine == "rotate":
            rg = self.waypoints[self.current_goal_index]
            goal_angle = math.atan2(rg[1] - self.X[1], rg[0] - self.X[0])
            if not self.enable_rotation:
                self.state_machine = "track"
            if abs(self.X[2] - goal_angle) > self.rotation_threshold:
                return rg[:2]
            self.state_machine = "track"
        if self.current_goal_index >= len(self.waypoints):
            return None
        wp = self.waypoints[self.current_goal_index]
        if np.linalg.norm(self.X[:2] - wp[:2]) < self.reached_threshold:
            self.current_goal_index += 1
            if self.current_goal_index >= len(self.waypoints):
                self.state_machine = "idle"
                return None
        return np.array(self.waypoints[self.current_goal_index][0:2])

    # -- one control step -----------------------------------------------------
    def control_step(self):
        """tracking.py:559-668 (static) / dynamic_env/main.py:126-236 (moving obstacles)."""
        m = self.model
        if self.state_machine == "stop":
            if R.has_stopped(m, self.X):
                self.state_machine = "rotate" if self.enable_rotation else "track"
                self.goal = self.update_goal()
        else:
            self.goal = self.update_goal()

        self.nearest_multi_obs = get_nearest_unpassed_obs(
            m, self.obs, self.X[:2], self.yaw if self.integrator else self.X[2], self.num_constraints)
        if self.dyn_obs and len(self.obs) and self.obs.shape[1] >= 5:   # main.py:54-58 (after selection)
            self.obs[:, 0] += self.obs[:, 3] * self.dt
            self.obs[:, 1] += self.obs[:, 4] * self.dt

        if self.state_machine == "rotate":
            ga = math.atan2(self.goal[1] - self.X[1], self.goal[0] - self.X[0])
            u_ref = R.rotate_to(m, self.X, ga)
        elif self.goal is None:
            u_ref = R.stop(m, self.X, self.spec)
        else:
            u_ref = R.nominal_input(m, self.X, self.goal, self.spec)
        control_ref = {"state_machine": self.state_machine, "u_ref": u_ref, "goal": self.goal}

        if self.solve_fn is not None:
            u, self.status = self.solve_fn(self.X, control_ref, self.nearest_multi_obs)
        else:
            obs_list = None if self.nearest_multi_obs is None else list(self.nearest_multi_obs)
            r = cbf_qp.solve(m, self.X, u_ref, obs_list, self.spec, self.cbf_param,
                             num_obs=self.num_constraints, dt=self.dt)
            u, self.status = r["u"], r["status"]

        collide = is_collide(self.X, self.obs, self.spec["radius"])
        if self.status != STATUS_OPTIMAL or collide:
            return -2
        self.X = R.step(m, self.X, u, self.dt, self.spec)
        self.u_pos = np.asarray(u, dtype=np.float64).reshape(-1)
        if is_collide(self.X, self.obs, self.spec["radius"]):
            return -2
        if self.goal is None and self.state_machine != "stop":
            return -1
        return 0
