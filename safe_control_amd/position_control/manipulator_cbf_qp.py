"""Manipulator2D CBF-QP backed by the gfx950 HIP kernel (csrc/manip_cbf_qp.hip).

The reference handles the 3-joint arm inside ``CBFQP`` itself (position_control/cbf_qp.py:34-35 alpha,
:94-104 three inputs with ``|u| <= w_max``, :130-151 one row per link circle per obstacle up to ``num_obs`` rows) on
top of robots/manipulator2D.py.  ``safe_control_amd.CBFQP(robot, robot_spec, num_obs)`` returns a ``ManipulatorCBFQP``
for ``robot_spec['model'] == 'Manipulator2D'``; ``BatchedManipulatorCBFQP`` is the same controller for B arms per
launch on device tensors.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec

SCALE = 60.0                                                  # manipulator2D.py:17
LINK_LENGTHS = tuple(np.array([80, 70, 50]) / SCALE)          # manipulator2D.py:18
STEP_LEN = 10.0 / 60.0                                        # manipulator2D.py:133
BETA = 1.3                                                    # manipulator2D.py:185
DEFAULT_NUM_ROWS = 150                                        # tracking.py:134-135


def link_steps(link_lengths=LINK_LENGTHS):
    """``int(np.ceil(link_dist / step_len))`` in float64, exactly as manipulator2D.py:143 evaluates it (8, 8, 6)."""
    return [int(np.ceil(L / STEP_LEN)) for L in link_lengths]


def make_params(robot_spec, alpha, dt, radius, io_dtype, num_rows, base_pos, link_lengths=LINK_LENGTHS, obs_shared=False):
    p = _lib.ManipCbfQpParams()
    p.io_dtype = io_dtype
    mode = robot_spec.get("cbf_mode", "cbf")                  # cbf_qp.py:120
    if mode not in _lib.CBF_MODE:
        raise ValueError(f"cbf_mode must be 'cbf' or 'hard', got {mode!r}")
    p.cbf_mode = _lib.CBF_MODE[mode]
    p.obs_shared = 1 if obs_shared else 0
    p.num_rows = int(num_rows)
    for i, s in enumerate(link_steps(link_lengths)):
        p.link_steps[i] = s
        p.link_lengths[i] = float(link_lengths[i])
    p.robot_radius = float(radius)
    p.dt = float(dt)
    p.alpha = float(alpha)
    p.w_max = float(robot_spec["w_max"])
    p.beta = BETA
    p.base_pos[0], p.base_pos[1] = float(base_pos[0]), float(base_pos[1])
    return p


def _pad_obstacle(ob):
    ob = np.asarray(ob, dtype=np.float64).reshape(-1)
    if ob.shape[0] < 3:
        raise ValueError(f"Invalid obstacle format: {ob}")
    out = np.zeros(7)
    out[: min(7, ob.shape[0])] = ob[:7]
    return out


class ManipulatorCBFQP:
    """Drop-in for position_control.cbf_qp.CBFQP with a Manipulator2D robot (single arm per call)."""

    def __init__(self, robot, robot_spec, num_obs=DEFAULT_NUM_ROWS, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.num_obs = int(num_obs)
        self.device = device
        self.cbf_param = {"alpha": 1.0}                       # cbf_qp.py:34-35
        if "cbf_alpha" in self.robot_spec:                    # cbf_qp.py:38-39
            self.cbf_param["alpha"] = float(self.robot_spec["cbf_alpha"])
        self.status = "optimal"
        self.setup_control_problem()

    def setup_control_problem(self):
        if not 1 <= self.num_obs <= _lib.MANIP_MAX_ROWS:
            raise ValueError(f"num_obs must be in [1, {_lib.MANIP_MAX_ROWS}] rows for Manipulator2D")
        self._lib = _lib.load()
        self._u = np.zeros(3, dtype=np.float64)
        self._status = np.zeros(1, dtype=np.int32)
        self.h = None

    def _geometry(self):
        inner = getattr(self.robot, "robot", None)            # BaseRobot.robot is the Manipulator2D instance
        base = getattr(inner, "base_pos", None)
        if base is None:
            base = self.robot_spec.get("base_pos", (0.0, 0.0))
        links = getattr(inner, "link_lengths", None)
        return np.asarray(base, dtype=np.float64).reshape(-1)[:2], (LINK_LENGTHS if links is None else tuple(float(v) for v in links))

    def solve_control_problem(self, robot_state, control_ref, obs_list):
        u_ref = np.asarray(control_ref["u_ref"], dtype=np.float64).reshape(-1)
        if obs_list is None:                                  # cbf_qp.py:113-118
            self.status = "optimal"
            return u_ref.reshape(-1, 1).copy()
        base, links = self._geometry()
        circles = sum(link_steps(links)) + 3
        rows = [_pad_obstacle(o) for o in obs_list if o is not None]
        rows = rows[: -(-self.num_obs // circles)]            # obstacles past the row cap are never read (cbf_qp.py:126-128)
        k = len(rows)
        K = max(k, 1)
        obs = np.zeros((K, 7), dtype=np.float64)
        if k:
            obs[:k] = np.asarray(rows)
        n_obs = np.array([k], dtype=np.int32)
        X = np.asarray(robot_state, dtype=np.float64).reshape(-1)[:3].copy()
        p = make_params(self.robot_spec, self.cbf_param["alpha"], self.robot.dt, self.robot.robot_radius,
                        _lib.DTYPE_F64, self.num_obs, base, links)
        h = np.zeros(self.num_obs, dtype=np.float64)
        rc = self._lib.sc_manip_cbfqp_solve_batch_host(
            C.byref(p), 1, K, X.ctypes.data, u_ref.ctypes.data, obs.ctypes.data, n_obs.ctypes.data,
            self._u.ctypes.data, self._status.ctypes.data, h.ctypes.data, int(self.device))
        _lib.check(rc, "sc_manip_cbfqp_solve_batch_host")
        st = int(self._status[0])
        self.status = _lib.STATUS_STRINGS[st]
        self.h = h[: min(self.num_obs, k * circles)]
        if st != _lib.STATUS_OPTIMAL:
            return None
        return self._u.reshape(-1, 1).copy()


class BatchedManipulatorCBFQP:
    """Manipulator2D CBF-QP for B arms per launch on device tensors.

    ``solve(X[B,3], u_ref[B,3], obs[B,K,7] | obs[K,7], n_obs[B]|None)`` -> ``u[B,3]`` (NaN where not optimal),
    ``status[B] int32``, ``h[B,num_rows]``.  Contiguous CUDA tensors of ``io_dtype``; the launch goes on the current
    torch stream and does not synchronise.
    """

    def __init__(self, robot_spec, dt=0.05, io_dtype="f64", num_rows=DEFAULT_NUM_ROWS, base_pos=(0.0, 0.0), alpha=None):
        self.robot_spec = complete_robot_spec(robot_spec)
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.num_rows = int(num_rows)
        if not 1 <= self.num_rows <= _lib.MANIP_MAX_ROWS:
            raise ValueError(f"num_rows must be in [1, {_lib.MANIP_MAX_ROWS}]")
        self.base_pos = tuple(base_pos)
        self.alpha = float(self.robot_spec.get("cbf_alpha", 1.0) if alpha is None else alpha)
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def solve(self, X, u_ref, obs, n_obs=None, want_h=True):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_ref", u_ref), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if X.shape != (B, 3) or u_ref.shape != (B, 3) or obs.shape[-1] != 7 or (not shared and obs.shape[0] != B):
            raise ValueError("expected X[B,3], u_ref[B,3], obs[B,K,7] or obs[K,7]")
        if n_obs is not None and not (n_obs.is_cuda and n_obs.dtype == torch.int32 and n_obs.shape == (B,)
                                      and n_obs.is_contiguous()):
            raise ValueError("n_obs must be a contiguous CUDA int32 tensor of shape [B]")
        u = torch.empty((B, 3), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        h = torch.empty((B, self.num_rows), dtype=dt_, device=X.device) if want_h else None
        p = make_params(self.robot_spec, self.alpha, self.dt, self.robot_spec["radius"], self.io_dtype, self.num_rows,
                        self.base_pos, obs_shared=shared)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        rc = self._lib.sc_manip_cbfqp_solve_batch(
            C.byref(p), B, K, X.data_ptr(), u_ref.data_ptr(), obs.data_ptr(),
            n_obs.data_ptr() if n_obs is not None else None, u.data_ptr(), status.data_ptr(),
            h.data_ptr() if h is not None else None, stream)
        _lib.check(rc, "sc_manip_cbfqp_solve_batch")
        return u, status, h


class BatchedManipulatorTracking:
    """Closed loop for B arms that share one known-obstacle table: ``control_step(n)`` runs n iterations of
    ``LocalTrackingController.control_step`` (tracking.py:559-668) with a Manipulator2D robot in ONE launch
    (csrc/manip_cbf_qp.hip: manip_rollout_kernel).  ``q0 [B,3]`` joint angles, ``obs [M,3|7]``, one waypoint list per arm
    or one shared list; return codes as the reference (0 running, -1 all waypoints reached, -2 infeasible / collision)."""

    def __init__(self, q0, robot_spec, base_pos=(0.0, 0.0), dt=0.05, obs=None, enable_rotation=True, io_dtype="f64",
                 device="cuda:0"):
        import torch
        self.torch = torch
        self.robot_spec = complete_robot_spec(robot_spec)
        self.dt = float(dt)
        self.base_pos = np.asarray(base_pos, dtype=np.float64).reshape(-1)[:2]
        self.enable_rotation = bool(enable_rotation)
        self.device = torch.device(device)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.tdtype = torch.float32 if io_dtype == "f32" else torch.float64
        self.num_constraints = int(self.robot_spec.get("num_constraints", DEFAULT_NUM_ROWS))       # tracking.py:134-138
        if not 1 <= self.num_constraints <= _lib.MANIP_MAX_ROWS:
            raise ValueError(f"num_constraints must be in [1, {_lib.MANIP_MAX_ROWS}]")
        self.reached_threshold = float(self.robot_spec.get("reached_threshold", 0.3))
        self.fov_angle = np.radians(float(self.robot_spec.get("fov_angle", 70.0)))
        self.alpha = float(self.robot_spec.get("cbf_alpha", 1.0))
        self._lib = _lib.load()
        q0 = np.asarray(q0, dtype=np.float64)
        if q0.ndim == 1:
            q0 = q0[None, :]
        self.B = q0.shape[0]
        self.X = torch.tensor(q0[:, :3], dtype=self.tdtype, device=self.device).contiguous()
        if obs is None or len(obs) == 0:
            tab = np.zeros((0, 7))
        else:
            tab = np.asarray(obs, dtype=np.float64)
            if tab.shape[1] < 7:
                tab = np.hstack([tab, np.zeros((tab.shape[0], 7 - tab.shape[1]))])
            # get_nearest_unpassed_obs ranks by the distance to robot.get_position() = the base (robots/robot.py:354-356)
            tab = tab[np.argsort(np.linalg.norm(tab[:, :2] - self.base_pos[None, :], axis=1))][:, :7]
        self.obs = torch.tensor(tab, dtype=self.tdtype, device=self.device).contiguous()
        z = lambda *shape, dt_=torch.int32: torch.zeros(shape, dtype=dt_, device=self.device)
        self.state_machine, self.current_goal_index, self.ret = z(self.B), z(self.B), z(self.B)
        self.ret_step = torch.full((self.B,), -1, dtype=torch.int32, device=self.device)
        self.goal = z(self.B, 3, dt_=self.tdtype)
        self.u_pos = z(self.B, 3, dt_=self.tdtype)
        self.waypoints = None
        self.steps_done = 0

    def _end_effector(self, q):
        ang = np.cumsum(q)
        return self.base_pos + np.array([np.sum(np.asarray(LINK_LENGTHS) * np.cos(ang)), np.sum(np.asarray(LINK_LENGTHS) * np.sin(ang))])

    def set_waypoints(self, waypoints):
        """set_waypoints / filter_waypoints / first update_goal (tracking.py:197-249, :497-535) per arm, on the host."""
        torch = self.torch
        Q = self.X.double().cpu().numpy()
        shared = (isinstance(waypoints, np.ndarray) and waypoints.ndim == 2) or \
            (isinstance(waypoints, (list, tuple)) and len(waypoints) > 0 and np.ndim(waypoints[0]) == 1)
        lists = [np.asarray(waypoints, dtype=np.float64)] * self.B if shared else [np.asarray(w, dtype=np.float64) for w in waypoints]
        filt = []
        for i in range(self.B):
            wp = lists[i]
            if len(wp) >= 2:
                aug = np.vstack((self._end_effector(Q[i]), wp[:, :2]))
                dist = np.linalg.norm(np.diff(aug, axis=0), axis=1)
                wp = aug[np.concatenate(([False], dist >= self.reached_threshold))]
            filt.append(np.asarray(wp, dtype=np.float64)[:, :2].reshape(-1, 2))
        W = max(1, max(len(w) for w in filt))
        wps = np.zeros((self.B, W, 2)); n_wp = np.zeros(self.B, dtype=np.int32)
        idx = np.zeros(self.B, dtype=np.int32); sm = np.zeros(self.B, dtype=np.int32); goal = np.zeros((self.B, 3))
        for i in range(self.B):
            w = filt[i]
            n_wp[i] = len(w); wps[i, : len(w)] = w
            g = None
            if len(w) > 0:
                if np.linalg.norm(self._end_effector(Q[i]) - w[0]) < self.reached_threshold:
                    idx[i] = 1
                if idx[i] < len(w):
                    g = w[idx[i]]
            if g is not None:                                   # is_in_fov from the base with yaw 0 (robots/robot.py:854-872)
                ang = np.arctan2(g[1] - self.base_pos[1], g[0] - self.base_pos[0])
                if abs(((ang + np.pi) % (2 * np.pi)) - np.pi) <= self.fov_angle / 2:
                    sm[i] = _lib.SM_TRACK; goal[i] = [g[0], g[1], 1.0]
                else:
                    sm[i] = _lib.SM_STOP
        self.waypoints = torch.tensor(wps, dtype=self.tdtype, device=self.device).contiguous()
        self.n_wp = torch.tensor(n_wp, dtype=torch.int32, device=self.device)
        self.current_goal_index = torch.tensor(idx, dtype=torch.int32, device=self.device)
        self.state_machine = torch.tensor(sm, dtype=torch.int32, device=self.device)
        self.goal = torch.tensor(goal, dtype=self.tdtype, device=self.device).contiguous()
        self.ret.zero_(); self.ret_step.fill_(-1)

    def control_step(self, n=1, record=False):
        torch = self.torch
        if self.waypoints is None:
            raise RuntimeError("call set_waypoints first")
        p = _lib.ManipTrackingParams()
        p.qp = make_params(self.robot_spec, self.alpha, self.dt, self.robot_spec["radius"], self.io_dtype, self.num_constraints,
                           self.base_pos, obs_shared=True)
        p.n_steps, p.max_waypoints, p.waypoints_shared = int(n), int(self.waypoints.shape[1]), 0
        p.step_offset = int(self.steps_done)
        p.enable_rotation = 1 if self.enable_rotation else 0
        p.Kp, p.reached_threshold, p.rotation_threshold = float(self.robot_spec["Kp"]), self.reached_threshold, 0.1
        tX = torch.empty((n, self.B, 3), dtype=self.tdtype, device=self.device) if record else None
        tU = torch.empty((n, self.B, 3), dtype=self.tdtype, device=self.device) if record else None
        M = int(self.obs.shape[0])
        rc = self._lib.sc_manip_tracking_rollout_batch(
            C.byref(p), self.B, M, self.X.data_ptr(), self.waypoints.data_ptr(), self.n_wp.data_ptr(),
            self.current_goal_index.data_ptr(), self.state_machine.data_ptr(), self.goal.data_ptr(),
            self.obs.data_ptr() if M else None, self.u_pos.data_ptr(), self.ret.data_ptr(), self.ret_step.data_ptr(),
            tX.data_ptr() if record else None, tU.data_ptr() if record else None, torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "sc_manip_tracking_rollout_batch")
        self.steps_done += n
        return (self.ret, tX, tU) if record else self.ret
