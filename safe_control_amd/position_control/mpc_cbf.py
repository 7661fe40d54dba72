"""MPC-CBF position controllers backed by the gfx950 HIP kernel (csrc/mpc_cbf.hip).

``MPCCBF`` keeps the plugin surface of the reference class of the same name
(position_control/mpc_cbf.py:6-402): ``__init__(robot, robot_spec,
show_mpc_traj=False, num_obs=5)``, ``setup_control_problem()``,
``update_tvp(goal, obs)``, ``solve_control_problem(robot_state, control_ref,
nearest_obs)``, attributes ``status``, ``cbf_param``, ``horizon``, ``Q``, ``R``,
``goal``, ``obs``.  ``BatchedMPCCBF`` solves B agents' NLPs in one launch, one
NLP per wavefront.

The reference's NLP is solved by IPOPT (do-mpc); here the same NLP is solved by
the kernel's own interior-point method from the same constant initial guess.
No CPU fallback: the HIP library must be present.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec

DUMMY_OBS = np.array([1000.0, 1000.0, 0.0, 0.0, 0.0, 0.0, 0.0])      # mpc_cbf.py:343,360


def default_mpc_weights(model):
    """Q, R of MPCCBF.__init__ (position_control/mpc_cbf.py:19-43)."""
    if model == "DynamicUnicycle2D":
        return np.diag([50.0, 50.0, 0.01, 30.0]), np.array([0.5, 0.5])
    if model == "Unicycle2D":
        return np.diag([50.0, 50.0, 0.01]), np.array([0.5, 0.5])
    if model == "SingleIntegrator2D":                       # mpc_cbf.py:19-21 (the multiple-shooting kernel; the condensed solve of this model is mpc_cbf_linear.py)
        return np.diag([50.0, 50.0]), np.array([5.0, 5.0])
    if model == "DoubleIntegrator2D":                       # mpc_cbf.py:28-30 (the multiple-shooting kernel; the condensed solve of this model is mpc_cbf_gn.py)
        return np.diag([50.0, 50.0, 20.0, 20.0]), np.array([0.5, 0.5])
    if model == "KinematicBicycle2D":                       # mpc_cbf.py:31-33 (likewise)
        return np.diag([50.0, 50.0, 1.0, 1.0]), np.array([0.5, 5000.0])
    raise NotImplementedError(f"MPC-CBF on the batched engine supports DynamicUnicycle2D and Unicycle2D, not {model}")


def default_mpc_cbf_param(model):
    """DT-CBF gains, position_control/mpc_cbf.py:49-88."""
    if model == "DynamicUnicycle2D":
        return {"alpha1": 0.15, "alpha2": 0.15}
    if model == "Unicycle2D":
        return {"alpha": 0.05}                              # mpc_cbf.py:52-53
    if model == "SingleIntegrator2D":
        return {"alpha": 0.05}                              # mpc_cbf.py:49-51
    if model == "DoubleIntegrator2D":
        return {"alpha1": 0.2, "alpha2": 0.2}               # mpc_cbf.py:56-59
    if model == "KinematicBicycle2D":
        return {"alpha1": 0.1, "alpha2": 0.1}               # mpc_cbf.py:64-66
    raise NotImplementedError(model)


def apply_mpc_overrides(cbf_param, robot_spec):
    """position_control/mpc_cbf.py:90-95."""
    for key, src in (("alpha", "mpc_cbf_alpha"), ("alpha1", "mpc_cbf_alpha1"), ("alpha2", "mpc_cbf_alpha2")):
        if src in robot_spec:
            cbf_param[key] = float(robot_spec[src])
    return cbf_param


def pad_obstacles(obs, num_obs):
    """update_tvp (mpc_cbf.py:338-364): 3-wide rows get zero tails, anything but 3/7 wide raises,
    missing rows become far-away dummies, extra rows are dropped."""
    out = np.tile(DUMMY_OBS, (num_obs, 1))
    if obs is None or len(obs) == 0:
        return out
    rows = []
    for ob in obs:
        ob = np.asarray(ob, dtype=np.float64).reshape(-1)
        if ob.shape[0] == 3:
            ob = np.concatenate([ob, [0.0, 0.0, 0.0, 0.0]])
        elif ob.shape[0] != 7:
            raise ValueError(f"Invalid obstacle format: {ob}")
        rows.append(ob)
    rows = np.array(rows)[:num_obs]
    out[: len(rows)] = rows
    return out


def make_params(robot_spec, cbf_param, Q, R, horizon, dt, radius, io_dtype, obs_shared=False,
                tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER, mu_init=0.1, mu_min=1e-9, acceptable_tol=1e-5, resto=None, slack_reset=0):
    p = _lib.MpcCbfParams()
    p.slack_reset = int(slack_reset)                         # 2: the config-5 extension (oracle/od_mpc_rd1.py)
    p.model_id = _lib.MODEL_IDS[robot_spec["model"]]
    p.io_dtype = io_dtype
    p.horizon = int(horizon)
    p.max_iter = int(max_iter)
    p.obs_shared = 1 if obs_shared else 0
    p.dt = float(dt)
    qd = np.diag(np.asarray(Q, dtype=np.float64)) if np.ndim(Q) == 2 else np.asarray(Q, dtype=np.float64)
    for i in range(4):
        p.Q[i] = float(qd[i]) if i < len(qd) else 0.0
    p.R[0], p.R[1] = float(R[0]), float(R[1])
    p.v_max = float(robot_spec["v_max"])
    if robot_spec["model"] == "SingleIntegrator2D":         # inputs [vx, vy] (mpc_cbf.py:183-187), one gain alpha
        p.alpha1, p.alpha2 = float(cbf_param["alpha"]), 0.0
        p.u_max[0], p.u_max[1] = float(robot_spec["v_max"]), float(robot_spec["v_max"])
    elif robot_spec["model"] == "Unicycle2D":               # inputs [v, omega] (mpc_cbf.py:188-192), one gain alpha
        p.alpha1, p.alpha2 = float(cbf_param["alpha"]), 0.0
        p.u_max[0], p.u_max[1] = float(robot_spec["v_max"]), float(robot_spec["w_max"])
    elif robot_spec["model"] == "KinematicBicycle2D":       # inputs [a, beta] (mpc_cbf.py:202-208); robot.step clips the speed to [v_min, v_max]
        p.alpha1, p.alpha2 = float(cbf_param["alpha1"]), float(cbf_param["alpha2"])
        p.u_max[0], p.u_max[1] = float(robot_spec["a_max"]), float(robot_spec["beta_max"])
        p.v_min, p.rear_ax_dist = float(robot_spec["v_min"]), float(robot_spec["rear_ax_dist"])
    elif robot_spec["model"] == "DoubleIntegrator2D":       # inputs [ax, ay] (mpc_cbf.py:196-200); v_max is the norm robot.step rescales the velocity to
        p.alpha1, p.alpha2 = float(cbf_param["alpha1"]), float(cbf_param["alpha2"])
        p.u_max[0], p.u_max[1] = float(robot_spec.get("ax_max", robot_spec["a_max"])), float(robot_spec.get("ay_max", robot_spec["a_max"]))
    else:
        p.alpha1, p.alpha2 = float(cbf_param["alpha1"]), float(cbf_param["alpha2"])
        p.u_max[0], p.u_max[1] = float(robot_spec["a_max"]), float(robot_spec["w_max"])
    p.robot_radius = float(radius)
    p.beta = 1.1 if robot_spec["model"] == "KinematicBicycle2D" else 1.01      # agent_barrier_dt defaults: kinematic_bicycle2D.py:175, dynamic_unicycle2D.py:188
    p.tol, p.mu_init, p.mu_min = float(tol), float(mu_init), float(mu_min)
    p.acceptable_tol = float(acceptable_tol)
    p.resto = resto if resto is not None else _lib.default_resto()     # feasibility restoration (sc_resto_params)
    return p


class MPCCBF:
    """Drop-in for position_control.mpc_cbf.MPCCBF (single agent per call).  ``robot_spec['mpc_formulation']`` (DynamicUnicycle2D, Unicycle2D):
    'multiple_shooting' (default since round 6: the NLP as do-mpc poses it under IPOPT's algorithm with its restoration phase,
    csrc/mpc_du_ms.hip, kernel 13 -- what the reference's solver returns also where the NLP has no feasible point) or 'condensed'
    (single shooting, csrc/mpc_cbf.hip); a DynamicUnicycle2D scene with superellipsoid rows runs on kernel 13's instantiation for them
    (csrc/mpc_du_ms_se.hip)."""

    def __new__(cls, robot, robot_spec, *args, **kwargs):
        # the reference serves every model from this one class; the linear models run on their own kernel
        if cls is MPCCBF and robot_spec.get("model") in ("SingleIntegrator2D", "Quad3D"):
            from .mpc_cbf_linear import LinearMPCCBF
            return LinearMPCCBF(robot, robot_spec, *args, **kwargs)
        if cls is MPCCBF and robot_spec.get("model") in ("DoubleIntegrator2D", "Quad2D", "KinematicBicycle2D", "KinematicBicycle2D_C3BF",
                                                         "KinematicBicycle2D_DPCBF"):
            from .mpc_cbf_gn import GnMPCCBF
            return GnMPCCBF(robot, robot_spec, *args, **kwargs)
        if cls is MPCCBF and robot_spec.get("model") == "VTOL2D":
            from .mpc_cbf_vtol import VtolMPCCBF
            return VtolMPCCBF(robot, robot_spec, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, robot, robot_spec, show_mpc_traj=False, num_obs=5, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.status = "optimal"                             # the reference hard-wires this (mpc_cbf.py:10)
        self.show_mpc_traj = show_mpc_traj
        self.num_obs = int(num_obs)
        self.device = device
        self.horizon = int(self.robot_spec.get("mpc_horizon", 10))       # mpc_cbf.py:15
        self.dt = robot.dt
        model = self.robot_spec["model"]
        self.Q, self.R = default_mpc_weights(model)
        self.n_controls, self.n_states = 2, (3 if model == "Unicycle2D" else 4)
        self.goal = np.array([0, 0])
        self.cbf_param = apply_mpc_overrides(default_mpc_cbf_param(model), self.robot_spec)
        self.obs = None
        self.setup_control_problem()

    def setup_control_problem(self):
        """The reference compiles a do-mpc/casadi NLP here (mpc_cbf.py:102-106)."""
        if not 1 <= self.horizon <= _lib.MPCCBF_MAX_HORIZON:
            raise ValueError(f"mpc_horizon must be in [1, {_lib.MPCCBF_MAX_HORIZON}]")
        self._lib = _lib.load()
        self.u_prev = np.zeros(2, dtype=np.float64)         # do-mpc's u0: last applied input, zeros at start
        self.z = np.zeros(2 * self.horizon, dtype=np.float64)
        self.iterations = 0
        self.solver_status = "optimal"
        self._ms = None
        want = self.robot_spec.get("mpc_formulation", "multiple_shooting")
        if self.robot_spec["model"] in ("DynamicUnicycle2D", "Unicycle2D") and want == "multiple_shooting" and self.horizon <= 62 and self.num_obs <= 16:
            from .mpc_cbf_ms import BatchedMSMPCCBF
            self._ms = BatchedMSMPCCBF(self.robot_spec, dt=self.dt, io_dtype="f64", horizon=self.horizon, cbf_param=self.cbf_param, check_circles=False)

    def update_tvp(self, goal, obs):
        self.goal = np.array(goal)
        self.obs = pad_obstacles(obs, self.num_obs)

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        goal = control_ref["goal"]
        self.update_tvp(goal, nearest_obs)
        if control_ref["state_machine"] != "track":         # mpc_cbf.py:379-381: pass the reference through
            return control_ref["u_ref"]
        X = np.zeros(4)                                     # the C-ABI takes [B,4] rows; Unicycle2D leaves X[3] unused
        xs = np.asarray(robot_state, dtype=np.float64).reshape(-1)[: self.n_states]
        X[: xs.shape[0]] = xs
        g = np.ascontiguousarray(np.asarray(self.goal, dtype=np.float64).reshape(-1)[:2])
        obs = np.ascontiguousarray(self.obs, dtype=np.float64)
        se = bool((obs[:, 6] >= 0.5).any())
        if self._ms is not None and (not se or self.robot_spec["model"] == "DynamicUnicycle2D"):      # (superellipsoid rows: csrc/mpc_du_ms_se.hip; Unicycle2D's barrier has no such branch)
            import torch
            self._ms.superellipsoids = se
            dev = torch.device("cuda", int(self.device))
            t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), dtype=torch.float64, device=dev)     # noqa: E731
            self._ms.cbf_param = self.cbf_param               # (users mutate cbf_param in place: README "online adaptive CBF")
            self._ms.robot_spec["radius"] = self.robot.robot_radius
            u, st, it, plan = self._ms.solve(t(X[None]), t(self.u_prev[None]), t(g[None]), t(obs[None]), want_plan=True)
            self.iterations = int(it[0].item())
            self.solver_status = _lib.STATUS_STRINGS[int(st[0].item())]
            self.z = plan[0, (self.horizon + 1) * 4:].cpu().numpy().copy()      # the planned inputs u_0 .. u_{N-1}
            self.u_prev = u[0].cpu().numpy().copy()
            return self.u_prev.reshape(-1, 1).copy()
        p = make_params(self.robot_spec, self.cbf_param, self.Q, self.R, self.horizon, self.dt,
                        self.robot.robot_radius, _lib.DTYPE_F64)
        u = np.zeros(2); st = np.zeros(1, dtype=np.int32); it = np.zeros(1, dtype=np.int32)
        rc = self._lib.sc_mpccbf_solve_batch_host(
            C.byref(p), 1, self.num_obs, X.ctypes.data, self.u_prev.ctypes.data, g.ctypes.data, obs.ctypes.data,
            u.ctypes.data, st.ctypes.data, it.ctypes.data, self.z.ctypes.data, int(self.device))
        _lib.check(rc, "sc_mpccbf_solve_batch_host")
        self.iterations = int(it[0])
        # the reference never reports MPC failures to control_step (status stays 'optimal', mpc_cbf.py:10,400);
        # the real outcome is kept in solver_status
        self.solver_status = _lib.STATUS_STRINGS[int(st[0])]
        self.u_prev = u.copy()
        return u.reshape(-1, 1).copy()


class BatchedMPCCBF(_lib.SlicedSolver):
    """MPC-CBF for B agents per launch on device tensors.

    ``solve(X[B,4], u_prev[B,2], goal[B,2], obs[B,K,7] | obs[K,7])`` ->
    ``u[B,2]``, ``status[B] int32``, ``iters[B] int32`` (and ``z[B,2N]`` if asked).
    Unicycle2D states are [x, y, theta] padded to 4 columns (the last is not read).
    ``iter_slices`` / ``classify_first`` / ``order``: continuation launches (include/safe_control_amd.h: sc_mpc_slices).
    """

    def __init__(self, robot_spec, dt=0.05, io_dtype="f32", horizon=None, cbf_param=None,
                 tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER, iter_slices=None, classify_first=True, order=True):
        self.init_slices(iter_slices, classify_first, order)
        self.robot_spec = complete_robot_spec(robot_spec)
        model = self.robot_spec["model"]
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = int(horizon if horizon is not None else self.robot_spec.get("mpc_horizon", 10))
        self.Q, self.R = default_mpc_weights(model)
        self.cbf_param = cbf_param or apply_mpc_overrides(default_mpc_cbf_param(model), self.robot_spec)
        self.tol, self.max_iter = tol, max_iter
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def solve(self, X, u_prev, goal, obs, want_z=False, out=None):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_prev", u_prev), ("goal", goal), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if X.shape != (B, 4) or u_prev.shape != (B, 2) or goal.shape != (B, 2) or obs.shape[-1] != 7 \
                or (not shared and obs.shape[0] != B):
            raise ValueError("expected X[B,4], u_prev[B,2], goal[B,2], obs[B,K,7] or obs[K,7]")
        if out is None:
            u = torch.empty((B, 2), dtype=dt_, device=X.device)
            status = torch.empty((B,), dtype=torch.int32, device=X.device)
            iters = torch.empty((B,), dtype=torch.int32, device=X.device)
            z = torch.empty((B, 2 * self.horizon), dtype=dt_, device=X.device) if want_z else None
        else:
            u, status, iters, z = out
        p = make_params(self.robot_spec, self.cbf_param, self.Q, self.R, self.horizon, self.dt,
                        self.robot_spec["radius"], self.io_dtype, obs_shared=shared, tol=self.tol,
                        max_iter=self.max_iter, resto=getattr(self, "resto", None))   # .resto: a _lib.RestoParams override
        stream = torch.cuda.current_stream(X.device).cuda_stream
        sl = self.slices_for(lambda: self._lib.sc_mpccbf_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(),
                u.data_ptr(), status.data_ptr(), iters.data_ptr(), z.data_ptr() if z is not None else None, stream)
        if sl is None:
            rc = self._lib.sc_mpccbf_solve_batch(C.byref(p), *args)
        else:
            rc = self._lib.sc_mpccbf_solve_batch_sliced(C.byref(p), C.byref(sl), *args)
        _lib.check(rc, "sc_mpccbf_solve_batch")
        return (u, status, iters, z) if want_z else (u, status, iters)
