"""MPC-CBF for VTOL2D on the gfx950 kernel csrc/mpc_vtol_wave.hip (one NLP per wavefront, one stage per lane, stage-wise Riccati Newton steps;
the one-NLP-per-lane kernel it was developed against, ``kernel = 1``, was retired in round 6 and is refused by the C-ABI).

``safe_control_amd.MPCCBF(robot, robot_spec, ...)`` returns a ``VtolMPCCBF`` for ``model == 'VTOL2D'`` (the reference serves
every model from the one MPCCBF class, position_control/mpc_cbf.py:7-100; VTOL2D: :40-43 weights, :83-87 gains, horizon 30,
:222-233 bounds); ``BatchedVtolMPCCBF`` solves B aircraft per launch on device tensors.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec
from .mpc_cbf import apply_mpc_overrides, pad_obstacles

Q_VTOL = [10.0, 10.0, 250.0, 10.0, 10.0, 50.0]                 # mpc_cbf.py:40-41
R_VTOL = [0.5, 0.5, 0.5, 50000.0]                              # mpc_cbf.py:42-43
CBF_VTOL = {"alpha1": 0.05, "alpha2": 0.05}                    # mpc_cbf.py:83-87
HORIZON_VTOL = 30


def make_params(robot_spec, cbf_param, horizon, dt, radius, io_dtype, obs_shared=False, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER, mu_init=0.1,
                mu_min=1e-9, acceptable_tol=1e-5, resto=None, slack_reset=2, kernel=0):
    p = _lib.MpcVtolParams()
    p.io_dtype, p.horizon, p.max_iter, p.obs_shared, p.acceptable_iter = io_dtype, int(horizon), int(max_iter), 1 if obs_shared else 0, 15
    p.slack_reset = int(slack_reset)
    p.kernel = int(kernel)                                 # 0 / 2: one NLP per wavefront (the kernel that serves the model); 1 (one NLP per lane) is retired and refused
    p.dt = float(dt)
    for i in range(6):
        p.Q[i] = Q_VTOL[i]
    lo = [robot_spec["throttle_min"]] * 3 + [robot_spec["elevator_min"]]
    hi = [robot_spec["throttle_max"]] * 3 + [robot_spec["elevator_max"]]
    for i in range(4):
        p.R[i], p.u_lo[i], p.u_hi[i] = R_VTOL[i], float(lo[i]), float(hi[i])
    p.alpha1, p.alpha2 = float(cbf_param["alpha1"]), float(cbf_param["alpha2"])
    p.v_max, p.descent_speed_max = float(robot_spec["v_max"]), float(robot_spec["descent_speed_max"])
    p.pitch_max = float(robot_spec["pitch_max"]) * 3.14159 / 180          # mpc_cbf.py:232-233 (its own pi)
    p.robot_radius, p.beta = float(radius), 1.01                          # vtol2D.py:475-497
    p.tol, p.acceptable_tol, p.mu_init, p.mu_min = float(tol), float(acceptable_tol), float(mu_init), float(mu_min)
    for i, k in enumerate(_lib.VTOL_AIRFRAME_KEYS):
        p.airframe[i] = float(robot_spec[k])
    p.resto = resto if resto is not None else _lib.default_resto(slack_reset=0, retry_max=0, stall_iter=0, gauss_newton=1)      # oracle/mpc_vtol.py: params
    return p


class VtolMPCCBF:
    """Drop-in for position_control.mpc_cbf.MPCCBF with a VTOL2D robot (horizon fixed at 30: mpc_cbf.py:41).  ``robot_spec['mpc_formulation']``:
    'multiple_shooting' (default since round 5: the NLP as do-mpc poses it under IPOPT's algorithm with its restoration phase,
    csrc/mpc_vtol_ms.hip -- the formulation that lands examples/test_vtol.py) or 'condensed' (single shooting, csrc/mpc_vtol_wave.hip)."""

    def __init__(self, robot, robot_spec, show_mpc_traj=False, num_obs=5, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.status = "optimal"                               # mpc_cbf.py:10
        self.show_mpc_traj = show_mpc_traj
        self.num_obs = int(num_obs)
        self.device = device
        self.horizon = HORIZON_VTOL
        self.dt = robot.dt
        self.Q, self.R = np.diag(Q_VTOL), np.array(R_VTOL)
        self.n_states, self.n_controls = 6, 4
        self.goal = np.array([0, 0])
        self.cbf_param = apply_mpc_overrides(dict(CBF_VTOL), self.robot_spec)
        self.obs = None
        self.setup_control_problem()

    def setup_control_problem(self):
        self._lib = _lib.load()
        self.u_prev = np.zeros(4)
        self.z = np.zeros(4 * self.horizon)
        self.iterations = 0
        self.solver_status = "optimal"
        self._ms = None
        if self.robot_spec.get("mpc_formulation", "multiple_shooting") != "condensed":
            from .mpc_cbf_vtol_ms import BatchedVtolMSMPCCBF
            self._ms = BatchedVtolMSMPCCBF(self.robot_spec, dt=self.dt, io_dtype="f64", cbf_param=self.cbf_param, fallback=False)

    def update_tvp(self, goal, obs):
        self.goal = np.array(goal)
        self.obs = pad_obstacles(obs, self.num_obs)

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        self.update_tvp(control_ref["goal"], nearest_obs)
        if control_ref["state_machine"] != "track":           # mpc_cbf.py:379-381
            return control_ref["u_ref"]
        X = np.zeros(6)
        xs = np.asarray(robot_state, dtype=np.float64).reshape(-1)[:6]
        X[: xs.shape[0]] = xs
        g = np.ascontiguousarray(np.asarray(self.goal, dtype=np.float64).reshape(-1)[:2])
        obs = np.ascontiguousarray(self.obs, dtype=np.float64)
        if self._ms is not None:
            import torch
            dev = torch.device("cuda", int(self.device))
            t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), dtype=torch.float64, device=dev)     # noqa: E731
            self._ms.cbf_param = self.cbf_param               # (users mutate cbf_param in place: README "online adaptive CBF")
            self._ms.robot_spec["radius"] = self.robot.robot_radius
            u, st, it, plan = self._ms.solve(t(X[None]), t(self.u_prev[None]), t(g[None]), t(obs[None]), want_plan=True)
            self.iterations = int(it[0].item())
            self.solver_status = _lib.STATUS_STRINGS[int(st[0].item())]
            self.z = plan[0, (self.horizon + 1) * 6:].cpu().numpy().copy()      # the planned inputs u_0 .. u_{N-1}
            self.u_prev = u[0].cpu().numpy().copy()
            return self.u_prev.reshape(-1, 1).copy()
        p = make_params(self.robot_spec, self.cbf_param, self.horizon, self.dt, self.robot.robot_radius, _lib.DTYPE_F64)
        u = np.zeros(4); st = np.zeros(1, dtype=np.int32); it = np.zeros(1, dtype=np.int32)
        rc = self._lib.sc_mpcvtol_solve_batch_host(
            C.byref(p), 1, self.num_obs, X.ctypes.data, self.u_prev.ctypes.data, g.ctypes.data, obs.ctypes.data,
            u.ctypes.data, st.ctypes.data, it.ctypes.data, self.z.ctypes.data, int(self.device))
        _lib.check(rc, "sc_mpcvtol_solve_batch_host")
        self.iterations = int(it[0])
        self.solver_status = _lib.STATUS_STRINGS[int(st[0])]
        self.u_prev = u.copy()
        return u.reshape(-1, 1).copy()


class BatchedVtolMPCCBF(_lib.SlicedSolver):
    """``solve(X[B,6], u_prev[B,4], goal[B,2], obs[B,K,7] | obs[K,7])`` -> ``u[B,4]``, ``status[B]``, ``iters[B]`` (and ``z[B,4N]`` if
    asked).  The default kernel (one NLP per wavefront, one stage per lane: csrc/mpc_vtol_wave.hip) keeps everything in registers
    and LDS and needs no workspace (``kernel = 1``, the one-NLP-per-lane kernel it was checked against, is retired: the C-ABI refuses it).
    ``iter_slices`` / ``classify_first`` / ``order``:
    continuation launches of the wave kernel (include/safe_control_amd.h: sc_mpc_slices)."""

    def __init__(self, robot_spec=None, dt=0.05, io_dtype="f64", cbf_param=None, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                 iter_slices=None, classify_first=True, order=True):
        self.init_slices(iter_slices, classify_first, order)
        self.robot_spec = complete_robot_spec(dict(robot_spec or {"model": "VTOL2D"}))
        if self.robot_spec["model"] != "VTOL2D":
            raise NotImplementedError("this controller serves VTOL2D")
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = HORIZON_VTOL
        self.Q, self.R = np.diag(Q_VTOL), np.array(R_VTOL)
        self.cbf_param = cbf_param or apply_mpc_overrides(dict(CBF_VTOL), self.robot_spec)
        self.tol, self.max_iter = tol, max_iter
        self.slack_reset = 2
        self.kernel = 0
        self._lib = _lib.load()
        self._ws = None

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def solve(self, X, u_prev, goal, obs, want_z=False):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_prev", u_prev), ("goal", goal), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if X.shape != (B, 6) or u_prev.shape != (B, 4) or goal.shape != (B, 2) or obs.shape[-1] != 7 or (not shared and obs.shape[0] != B):
            raise ValueError("expected X[B,6], u_prev[B,4], goal[B,2], obs[B,K,7] or obs[K,7]")
        u = torch.empty((B, 4), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        z = torch.empty((B, 4 * self.horizon), dtype=dt_, device=X.device) if want_z else None
        p = make_params(self.robot_spec, self.cbf_param, self.horizon, self.dt, self.robot_spec["radius"], self.io_dtype,
                        obs_shared=shared, tol=self.tol, max_iter=self.max_iter, resto=getattr(self, "resto", None),
                        slack_reset=self.slack_reset, kernel=self.kernel)
        need = int(self._lib.sc_mpcvtol_workspace_bytes(C.byref(p), B, K))
        if self._ws is None or self._ws.numel() < need or self._ws.device != X.device:
            self._ws = torch.empty((max(need, 8),), dtype=torch.uint8, device=X.device)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        # (continuation launches are the wave kernel's; the one-NLP-per-lane cross-check kernel runs one launch)
        sl = None if self.kernel == 1 else self.slices_for(lambda: self._lib.sc_mpcvtol_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(), u.data_ptr(),
                status.data_ptr(), iters.data_ptr(), z.data_ptr() if z is not None else None)
        if sl is None:
            rc = self._lib.sc_mpcvtol_solve_batch(C.byref(p), *args, self._ws.data_ptr(), need, stream)
        else:
            rc = self._lib.sc_mpcvtol_solve_batch_sliced(C.byref(p), C.byref(sl), *args, stream)
        _lib.check(rc, "sc_mpcvtol_solve_batch")
        return (u, status, iters, z) if want_z else (u, status, iters)


# ---- optimal decay (position_control/optimal_decay_mpc_cbf.py with a VTOL2D robot) ---------------------------------------------
OD_CBF_VTOL = {"alpha1": 0.35, "alpha2": 0.35, "omega1": 1.0, "p_sb1": 10.0, "omega2": 1.0, "p_sb2": 10.0}     # optimal_decay_mpc_cbf.py:83-91


def make_od_params(robot_spec, cbf_param, horizon, dt, radius, io_dtype, obs_shared=False, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER):
    p = _lib.OdMpcVtolParams()
    p.mpc = make_params(robot_spec, cbf_param, horizon, dt, radius, io_dtype, obs_shared=obs_shared, tol=tol, max_iter=max_iter)
    p.omega_ref[0], p.omega_ref[1] = float(cbf_param.get("omega1", 1.0)), float(cbf_param.get("omega2", 1.0))
    p.p_sb[0], p.p_sb[1] = float(cbf_param.get("p_sb1", 10.0)), float(cbf_param.get("p_sb2", 10.0))
    return p


class BatchedOptimalDecayVtolMPCCBF(_lib.SlicedSolver):
    """Optimal-decay MPC-CBF for B aircraft per launch (csrc/mpc_vtol_wave.hip, OD instantiation: the two decay variables of a stage
    are eliminated from the stage block before the Riccati recursion).

    ``solve(X[B,6], u_prev[B,4], goal[B,2], obs[B,K,7] | obs[K,7])`` -> ``u[B,4]``, ``rho[B,2N]`` (omega1_k, omega2_k per stage),
    ``status[B]``, ``iters[B]`` (and ``z[B,4N]`` if asked).  ``u_prev`` is taken for the signature's sake: the input term of this
    class is R u^2 (optimal_decay_mpc_cbf.py:173-174).  No CPU fallback."""

    def __init__(self, robot_spec=None, dt=0.05, io_dtype="f64", cbf_param=None, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                 iter_slices=None, classify_first=True, order=True):
        self.init_slices(iter_slices, classify_first, order)
        self.robot_spec = complete_robot_spec(dict(robot_spec or {"model": "VTOL2D"}))
        if self.robot_spec["model"] != "VTOL2D":
            raise NotImplementedError("this controller serves VTOL2D")
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = HORIZON_VTOL
        self.Q, self.R = np.diag(Q_VTOL), np.array(R_VTOL)
        self.cbf_param = cbf_param or dict(OD_CBF_VTOL)
        self.tol, self.max_iter = tol, max_iter
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def solve(self, X, u_prev, goal, obs, want_z=False):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_prev", u_prev), ("goal", goal), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if X.shape != (B, 6) or u_prev.shape != (B, 4) or goal.shape != (B, 2) or obs.shape[-1] != 7 or (not shared and obs.shape[0] != B):
            raise ValueError("expected X[B,6], u_prev[B,4], goal[B,2], obs[B,K,7] or obs[K,7]")
        u = torch.empty((B, 4), dtype=dt_, device=X.device)
        rho = torch.empty((B, 2 * self.horizon), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        z = torch.empty((B, 4 * self.horizon), dtype=dt_, device=X.device) if want_z else None
        p = make_od_params(self.robot_spec, self.cbf_param, self.horizon, self.dt, self.robot_spec["radius"], self.io_dtype,
                           obs_shared=shared, tol=self.tol, max_iter=self.max_iter)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        sl = self.slices_for(lambda: self._lib.sc_odmpcvtol_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(), u.data_ptr(), rho.data_ptr(),
                status.data_ptr(), iters.data_ptr(), z.data_ptr() if z is not None else None, stream)
        rc = self._lib.sc_odmpcvtol_solve_batch(C.byref(p), *args) if sl is None else self._lib.sc_odmpcvtol_solve_batch_sliced(C.byref(p), C.byref(sl), *args)
        _lib.check(rc, "sc_odmpcvtol_solve_batch")
        return (u, rho, status, iters, z) if want_z else (u, rho, status, iters)


class OptimalDecayVtolMPCCBF:
    """Drop-in for position_control.optimal_decay_mpc_cbf.OptimalDecayMPCCBF with a VTOL2D robot (single agent per call; the one
    NLP goes through the batched entry point on device ``device``).  Horizon 30, Q / R of :44-47, gains 0.35 (:83-86).
    The NLP is solved in the multiple-shooting form (the decay rates two more inputs of a stage: csrc/mpc_vtol_ms.hip, 99.95 % optimal in
    0.14 s per 4096 on the bench batch).  The condensed optimal-decay kernel (87 % optimal, one solve at the 3000-iteration budget, 1.1 s) is
    no longer selectable here: ``robot_spec['mpc_formulation'] = 'condensed'`` raises; ``BatchedOptimalDecayVtolMPCCBF`` remains as the
    restoration-less kernel behind ``BatchedOptimalDecayVtolMSMPCCBF(restoration=False, fallback=True)`` with a budget of 300 iterations."""

    def __init__(self, robot, robot_spec, num_obs=5, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.status = "optimal"                               # optimal_decay_mpc_cbf.py:21
        self.num_obs = int(num_obs)
        self.device = device
        self.horizon = HORIZON_VTOL                           # :44
        self.dt = robot.dt
        self.Q, self.R = np.diag(Q_VTOL), np.array(R_VTOL)
        self.n_states, self.n_controls = 6, 4
        self.cbf_param = dict(OD_CBF_VTOL)
        self.omega1 = None                                    # :92-93
        self.omega2 = None
        self.goal = np.array([0, 0])
        self.obs = None
        self.setup_control_problem()

    def setup_control_problem(self):
        if self.robot_spec.get("mpc_formulation", "multiple_shooting") == "condensed":
            raise ValueError("OptimalDecayMPCCBF for VTOL2D: the condensed kernel was withdrawn as a position controller in round 6 (87 % optimal, 1.1 s per 4096 "
                             "with one solve at the iteration budget); the multiple-shooting kernel serves the model")
        self.multiple_shooting = True
        from .mpc_cbf_vtol_ms import BatchedOptimalDecayVtolMSMPCCBF
        self._ctl = BatchedOptimalDecayVtolMSMPCCBF(self.robot_spec, dt=self.dt, io_dtype="f64", cbf_param=self.cbf_param, fallback=False)
        self.u_prev = np.zeros(4)
        self.z = np.zeros(4 * self.horizon)
        self.rho = np.ones(2 * self.horizon)
        self.iterations = 0
        self.solver_status = "optimal"

    def update_tvp(self, goal, obs):
        self.goal = np.array(goal)
        self.obs = pad_obstacles(obs, self.num_obs)

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        import torch
        self.update_tvp(control_ref["goal"], nearest_obs)
        if control_ref["state_machine"] != "track":           # optimal_decay_mpc_cbf.py:339-341
            return control_ref["u_ref"]
        X = np.zeros(6)
        xs = np.asarray(robot_state, dtype=np.float64).reshape(-1)[:6]
        X[: xs.shape[0]] = xs
        dev = torch.device("cuda", int(self.device))
        t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), dtype=torch.float64, device=dev)     # noqa: E731
        self._ctl.cbf_param = self.cbf_param                  # (users mutate cbf_param in place: README "online adaptive CBF")
        self._ctl.robot_spec["radius"] = self.robot.robot_radius
        u, rho, st, it, z = self._ctl.solve(t(X[None]), t(self.u_prev[None]), t(np.asarray(self.goal, dtype=np.float64).reshape(-1)[None, :2]),
                                            t(self.obs[None]), want_z=True)
        if self.multiple_shooting:
            z = z[:, (self.horizon + 1) * 6:]                 # (the plan holds the states first: keep the planned inputs)
        self.iterations = int(it[0].item())
        self.solver_status = _lib.STATUS_STRINGS[int(st[0].item())]
        self.rho, self.z = rho[0].cpu().numpy(), z[0].cpu().numpy()
        self.omega1, self.omega2 = float(self.rho[0]), float(self.rho[1])
        self.u_prev = u[0].cpu().numpy().copy()
        return self.u_prev.reshape(-1, 1).copy()
