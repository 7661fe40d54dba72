"""Backup-CBF QP backed by the gfx950 HIP kernel (csrc/backup_cbf.hip), SURVEY 8f-4.

``BackupCBF`` keeps the surface of the reference class of the same name (position_control/backup_cbf_qp.py:33-826):
``__init__(robot, robot_spec, dt, backup_horizon, ax=None)``, ``set_backup_controller``, ``set_environment``,
``set_moving_obstacles``, ``set_nominal_controller`` / ``set_nominal_trajectory``, ``solve_control_problem(robot_state)``,
``is_using_backup()``, ``get_status()``.  The reference composes arbitrary Python callables (robot.step, a backup
controller, an environment, an obstacle predictor); the native path serves the composition the reference ships --
examples/evade/test_evade.py --algo backupcbf: DoubleIntegrator2D + EvadeBackupController + EvadeEnv with its
constant-speed bullet -- and raises for anything else (DriftingCar / LaneChangeController are not served).
``BatchedBackupCBF`` runs B agents per launch on device tensors, and the example's closed loop (``control_step``)
fused in one kernel.  The QP is solved exactly (the reference calls OSQP).  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib

ENV_KEYS = ("hallway_length", "half_width", "pocket_x_min", "pocket_x_max", "pocket_y_min", "pocket_y_max", "goal_x_min",
            "goal_x_max", "bullet_speed", "bullet_length", "bullet_width", "bullet_start_x")


def default_evade_env(**over):
    """EvadeEnv(...) as examples/evade/test_evade.py:60-72,278-288 builds it (envs/evade_env.py:51-83)."""
    c = dict(hallway_length=60.0, hallway_width=4.0, pocket_x=25.0, pocket_length=10.0, pocket_width=4.0, goal_length=5.0,
             bullet_speed=3.0, bullet_length=3.0, bullet_width=None, bullet_start_x=-10.0)
    c.update(over)
    half = c["hallway_width"] / 2
    return dict(hallway_length=c["hallway_length"], half_width=half, pocket_x_min=c["pocket_x"],
                pocket_x_max=c["pocket_x"] + c["pocket_length"], pocket_y_min=half, pocket_y_max=half + c["pocket_width"],
                goal_x_min=c["hallway_length"] - c["goal_length"], goal_x_max=c["hallway_length"], bullet_speed=c["bullet_speed"],
                bullet_length=c["bullet_length"], bullet_width=c["bullet_width"] if c["bullet_width"] else c["hallway_width"],
                bullet_start_x=c["bullet_start_x"])


def env_from_object(env):
    """Read the EvadeEnv attributes the reference's _h_safety / _h_terminal use (backup_cbf_qp.py:359-392,481-494)."""
    missing = [k for k in ("half_width", "pocket_x_min", "pocket_x_max", "pocket_y_max", "hallway_length", "get_pocket_bounds")
               if not hasattr(env, k)]
    if missing:
        raise NotImplementedError(f"the native Backup-CBF path serves EvadeEnv (envs/evade_env.py); missing {missing}")
    return dict(hallway_length=float(env.hallway_length), half_width=float(env.half_width), pocket_x_min=float(env.pocket_x_min),
                pocket_x_max=float(env.pocket_x_max), pocket_y_min=float(env.pocket_y_min), pocket_y_max=float(env.pocket_y_max),
                goal_x_min=float(env.goal_x_min), goal_x_max=float(env.goal_x_max), bullet_speed=float(env.bullet_speed),
                bullet_length=float(env.bullet_length), bullet_width=float(env.bullet_width), bullet_start_x=float(env.bullet_start_x))


def make_params(env, robot_spec, dt, backup_horizon, io_dtype, bullet_shared=True, alpha=1.0, alpha_terminal=2.0, kp=2.0, kd=2.0):
    p = _lib.BackupCbfParams()
    p.io_dtype = io_dtype
    p.n_steps = int(backup_horizon / dt)                          # backup_cbf_qp.py:55
    p.bullet_shared = 1 if bullet_shared else 0
    p.dt, p.backup_horizon, p.fd_eps = float(dt), float(backup_horizon), 1e-5
    p.robot_radius = float(robot_spec.get("radius", 0.5))         # :352
    p.a_max = float(robot_spec.get("a_max", 2.0))                 # :690
    p.v_max = float(robot_spec.get("v_max", 1.5))                 # :533
    p.safety_margin = float(robot_spec.get("safety_margin", 0.0))  # :98-101
    p.alpha, p.alpha_terminal = float(alpha), float(alpha_terminal)
    p.backup_kp, p.backup_kd = float(kp), float(kd)               # backup_controller.py:449-450 (2.0, 2.0)
    for k in ENV_KEYS:
        setattr(p, k, float(env[k]))
    return p


class BatchedBackupCBF:
    """B agents of the evade scenario per launch.  ``solve(X[B,4], u_nom[B,2] | None, bullet_x[B] | [1])`` ->
    ``u[B,2], status[B] (-1 no rows / 0 solved / 1 infeasible), using_backup[B], h_min[B]`` (``want_rows``: also
    ``n_rows[B]``, ``rows[B, N, 3]`` float64, the reference's kept rows in scaled inputs).  ``rollout(...)`` runs the
    example's closed loop on device state."""

    def __init__(self, robot_spec=None, env=None, dt=0.1, backup_horizon=12.0, io_dtype="f64"):
        spec = dict(model="DoubleIntegrator2D", radius=0.5, a_max=2.0, v_max=1.5, safety_margin=0.5)   # test_evade.py:75-88,299
        spec.update(robot_spec or {})
        if spec.get("model", "DoubleIntegrator2D") not in ("DoubleIntegrator2D", "double_integrator"):
            raise NotImplementedError("the native Backup-CBF path serves DoubleIntegrator2D (the evade scenario)")
        self.robot_spec = spec
        self.env = dict(env) if env is not None else default_evade_env()
        self.dt, self.backup_horizon = float(dt), float(backup_horizon)
        self.io_dtype = _lib.DTYPE_F32 if io_dtype in ("f32", "float32") else _lib.DTYPE_F64
        self.N = int(self.backup_horizon / self.dt)
        self.alpha, self.alpha_terminal = 1.0, 2.0                # backup_cbf_qp.py:93-94
        self.kp, self.kd = 2.0, 2.0                               # EvadeBackupController's PD gains (backup_controller.py:449-450)
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def _check(self, X, bullet_x, u_nom=None):
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("bullet_x", bullet_x), ("u_nom", u_nom)):
            if t is not None and not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        if X.shape != (B, 4) or (u_nom is not None and u_nom.shape != (B, 2)) or bullet_x.numel() not in (1, B):
            raise ValueError("expected X[B,4], u_nom[B,2] or None, bullet_x[B] or [1]")
        return B, bullet_x.numel() == 1 and B != 1

    def solve(self, X, u_nom, bullet_x, want_rows=False):
        import torch
        B, shared = self._check(X, bullet_x, u_nom)
        dev = X.device
        u = torch.empty((B, 2), dtype=self.torch_dtype, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        using = torch.empty((B,), dtype=torch.int32, device=dev)
        h_min = torch.empty((B,), dtype=self.torch_dtype, device=dev)
        n_rows = torch.zeros((B,), dtype=torch.int32, device=dev) if want_rows else None
        rows = torch.zeros((B, self.N, 3), dtype=torch.float64, device=dev) if want_rows else None
        p = make_params(self.env, self.robot_spec, self.dt, self.backup_horizon, self.io_dtype, bullet_shared=shared,
                        alpha=self.alpha, alpha_terminal=self.alpha_terminal, kp=self.kp, kd=self.kd)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self._lib.sc_backupcbf_solve_batch(
            C.byref(p), B, X.data_ptr(), u_nom.data_ptr() if u_nom is not None else None, bullet_x.data_ptr(), u.data_ptr(),
            status.data_ptr(), using.data_ptr(), h_min.data_ptr(), n_rows.data_ptr() if want_rows else None,
            rows.data_ptr() if want_rows else None, stream)
        _lib.check(rc, "sc_backupcbf_solve_batch")
        return (u, status, using, h_min, n_rows, rows) if want_rows else (u, status, using, h_min)

    def rollout(self, X, bullet_x, ret, ret_step, n_ctrl, step_offset=0):
        """n_ctrl steps of the example's loop in one launch; X, bullet_x ([B]), ret, ret_step are updated in place.
        Returns (u_last, status, using_backup, h_min) of the last step."""
        import torch
        B, shared = self._check(X, bullet_x)
        if shared:
            raise ValueError("rollout needs one bullet position per agent (bullet_x[B])")
        dev = X.device
        u = torch.empty((B, 2), dtype=self.torch_dtype, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        using = torch.empty((B,), dtype=torch.int32, device=dev)
        h_min = torch.empty((B,), dtype=self.torch_dtype, device=dev)
        p = make_params(self.env, self.robot_spec, self.dt, self.backup_horizon, self.io_dtype, bullet_shared=False,
                        alpha=self.alpha, alpha_terminal=self.alpha_terminal, kp=self.kp, kd=self.kd)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self._lib.sc_backupcbf_rollout_batch(C.byref(p), B, int(n_ctrl), int(step_offset), X.data_ptr(), bullet_x.data_ptr(),
                                                  u.data_ptr(), status.data_ptr(), using.data_ptr(), h_min.data_ptr(),
                                                  ret.data_ptr(), ret_step.data_ptr(), stream)
        _lib.check(rc, "sc_backupcbf_rollout_batch")
        return u, status, using, h_min


class BackupCBF:
    """Drop-in for position_control.backup_cbf_qp.BackupCBF on the evade scenario (one robot per call)."""

    def __init__(self, robot, robot_spec, dt=0.05, backup_horizon=2.0, ax=None, device=0):
        self.robot, self.robot_spec = robot, robot_spec
        self.dt, self.backup_horizon = dt, backup_horizon
        self.N = int(backup_horizon / dt)
        self.n_states, self.n_controls = 4, 2
        self.nominal_controller = self.backup_controller = self.backup_target = None
        self.env = self.moving_obstacles = None
        self.nominal_x_traj = self.nominal_u_traj = None
        self.alpha, self.alpha_terminal = 1.0, 2.0
        self.safety_margin = robot_spec.get("safety_margin", 0.0)
        self.Q_u = np.array([1.0, 1.0])
        self._using_backup = self._last_intervention = False
        self._last_h_min, self.global_min_h = 1.0, float("inf")
        self.curr_step = 0
        self._device = device
        self._batched = None

    def set_nominal_controller(self, nominal_controller):
        self.nominal_controller = nominal_controller

    def set_backup_controller(self, backup_controller, target=None):
        for attr in ("safe_center", "safe_bounds", "Kp", "Kd"):
            if not hasattr(backup_controller, attr):
                raise NotImplementedError("the native Backup-CBF path serves EvadeBackupController (backup_controller.py:420)")
        # What the kernel implements of that controller: PD gains Kp / Kd (passed through), the pocket and the goal zone OF THE
        # ENVIRONMENT, one clamp a_max for the QP scaling and the backup input.  A controller set up differently would get silently
        # different inputs, so it is refused instead.
        a_spec = float(self.robot_spec.get("a_max", 2.0))
        if abs(float(getattr(backup_controller, "a_max", a_spec)) - a_spec) > 1e-12:
            raise NotImplementedError("backup controller a_max differs from robot_spec['a_max']: the kernel clamps both with one value")
        if getattr(backup_controller, "goal_bounds", True) is None:
            raise NotImplementedError("goal_bounds=None: the kernel always treats the goal zone of the environment as safe")
        self.backup_controller, self.backup_target = backup_controller, target
        self._batched = None

    def set_environment(self, env):
        self.env = env
        self._batched = None

    def set_nominal_trajectory(self, nominal_x_traj, nominal_u_traj):
        # same transposition rule as the reference (:160-170)
        for name, tr in (("nominal_x_traj", nominal_x_traj), ("nominal_u_traj", nominal_u_traj)):
            if tr is not None:
                tr = np.asarray(tr)
                if tr.ndim == 2 and tr.shape[0] < tr.shape[1]:
                    tr = tr.T
                setattr(self, name, np.array(tr))

    def set_moving_obstacles(self, obstacles):
        self.moving_obstacles = obstacles

    def _nominal(self, state):                                     # :177-184
        if self.nominal_u_traj is not None and len(self.nominal_u_traj) > 0:
            return np.asarray(self.nominal_u_traj[0], dtype=np.float64).flatten()
        if self.nominal_controller is not None:
            return np.array(self.nominal_controller(state.reshape(-1, 1)), dtype=np.float64).flatten()
        return np.zeros(self.n_controls)

    def solve_control_problem(self, robot_state, friction=None):
        import torch
        if self.env is None or self.backup_controller is None:
            raise RuntimeError("set_environment() and set_backup_controller() first")
        if self._batched is None:
            self._batched = BatchedBackupCBF(dict(self.robot_spec), env_from_object(self.env), self.dt, self.backup_horizon)
            self._batched.alpha, self._batched.alpha_terminal = self.alpha, self.alpha_terminal
            self._batched.kp, self._batched.kd = float(self.backup_controller.Kp), float(self.backup_controller.Kd)
            pc = np.asarray(self.backup_controller.safe_center, dtype=np.float64).flatten()
            e = self._batched.env
            want = np.array([0.5 * (e["pocket_x_min"] + e["pocket_x_max"]), 0.5 * (e["pocket_y_min"] + e["pocket_y_max"])])
            if np.abs(pc[:2] - want).max() > 1e-9:
                raise NotImplementedError("safe_center is not the centre of the environment's pocket: the kernel steers to the pocket of EvadeEnv")
        x = np.asarray(robot_state, dtype=np.float64).flatten()
        # the bullet as the obstacle predictor reports it at t = 0 (get_bullet_state: centre = bullet_x + length / 6)
        ob = self.moving_obstacles(0.0) if callable(self.moving_obstacles) else self.moving_obstacles
        if ob is None:
            raise NotImplementedError("the native path expects the evade scenario's active bullet")
        bx = float(ob["x"]) - float(self.env.bullet_length) / 6
        u_nom = self._nominal(x)
        if u_nom.shape[0] == 1:                                     # a [1, 2] trajectory is transposed by the rule above (:166)
            u_nom = np.repeat(u_nom, 2)
        dev = torch.device("cuda", self._device)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        u, st, using, hmin = self._batched.solve(t(x.reshape(1, 4)), t(u_nom.reshape(1, 2)), t([bx]))
        self._using_backup = self._last_intervention = bool(using.item())
        self._last_h_min = float(hmin.item())
        self.global_min_h = min(self.global_min_h, self._last_h_min)
        self.curr_step += 1
        self.qp_status = int(st.item())
        return u.cpu().numpy().reshape(-1, 1)

    def is_using_backup(self):
        return self._using_backup

    def get_status(self):
        return {"using_backup": self._using_backup, "last_intervention": self._last_intervention,
                "backup_horizon": self.backup_horizon, "h_min": self._last_h_min, "global_min_h": self.global_min_h,
                "num_constraints": self.N}
