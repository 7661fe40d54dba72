"""Optimal-decay MPC-CBF for KinematicBicycle2D and Quad2D on the gfx950 kernel csrc/mpc_gn.hip (OD instantiations).

position_control/optimal_decay_mpc_cbf.py:19 accepts DynamicUnicycle2D, KinematicBicycle2D, Quad2D, Quad3D and VTOL2D; the first is
served by csrc/mpc_cbf.hip (optimal_decay_mpc_cbf.py here), these two -- whose DT barrier steps the state with the robot's own
step() -- by the step()-barrier template.  Problem as the reference states it (:37-42 weights with R = (0.5, 50) for the bicycle,
:66-74 gains 0.05 / 0.15, :123-124 two omega inputs per stage, :178-186 R u^2 and the decay penalties, :291-297 the row); the
reference copy is stale and its solver absent, so parity is against oracle/od_mpc_gn.py only.  ``safe_control_amd.
OptimalDecayMPCCBF(robot, robot_spec)`` returns an ``OptimalDecayGnMPCCBF`` for these models.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec
from .mpc_cbf import pad_obstacles
from .mpc_cbf_gn import make_params, model_constants

OD_GN_MODELS = ("KinematicBicycle2D", "Quad2D")


def od_model_constants(robot_spec):
    """model_constants of MPCCBF with the optimal-decay class's own weights and gains (optimal_decay_mpc_cbf.py:37-42,66-74)."""
    mc = dict(model_constants(robot_spec))
    if robot_spec["model"] == "KinematicBicycle2D":
        mc.update(R=[0.5, 50.0], cbf_param={"alpha1": 0.05, "alpha2": 0.05})
    else:
        mc.update(cbf_param={"alpha1": 0.15, "alpha2": 0.15})
    mc["cbf_param"] = dict(mc["cbf_param"], omega1=1.0, p_sb1=10.0, omega2=1.0, p_sb2=10.0)      # :88-91
    return mc


def make_od_params(robot_spec, mc, cbf_param, horizon, dt, radius, io_dtype, obs_shared=False, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER):
    p = _lib.OdMpcGnParams()
    p.mpc = make_params(robot_spec, mc, cbf_param, horizon, dt, radius, io_dtype, obs_shared=obs_shared, tol=tol, max_iter=max_iter)
    p.omega_ref[0], p.omega_ref[1] = float(cbf_param.get("omega1", 1.0)), float(cbf_param.get("omega2", 1.0))
    p.p_sb[0], p.p_sb[1] = float(cbf_param.get("p_sb1", 10.0)), float(cbf_param.get("p_sb2", 10.0))
    return p


class BatchedOptimalDecayGnMPCCBF(_lib.SlicedSolver):
    """``solve(X[B,nx], u_prev[B,2], goal[B,2], obs[B,K,7] | obs[K,7])`` -> ``u[B,2]``, ``rho[B,2N]`` (omega1_k, omega2_k per
    stage), ``status[B]``, ``iters[B]`` (and ``z[B,2N]`` if asked); nx = 4 (KinematicBicycle2D) or 6 (Quad2D)."""

    def __init__(self, robot_spec, dt=0.05, io_dtype="f64", horizon=None, cbf_param=None, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                 iter_slices=None, classify_first=True, order=True):
        self.init_slices(iter_slices, classify_first, order)
        self.robot_spec = complete_robot_spec(robot_spec)
        if self.robot_spec["model"] not in OD_GN_MODELS:
            raise NotImplementedError(f"this controller serves {OD_GN_MODELS}")
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = int(horizon if horizon is not None else 10)           # optimal_decay_mpc_cbf.py:24: fixed at 10
        self._mc = od_model_constants(self.robot_spec)
        self.Q, self.R = np.diag(self._mc["Q"]), np.array(self._mc["R"])
        self.cbf_param = cbf_param or dict(self._mc["cbf_param"])
        self.nx = self._mc["nx"]
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
        if X.shape != (B, self.nx) or u_prev.shape != (B, 2) or goal.shape != (B, 2) or obs.shape[-1] != 7 \
                or (not shared and obs.shape[0] != B):
            raise ValueError(f"expected X[B,{self.nx}], u_prev[B,2], goal[B,2], obs[B,K,7] or obs[K,7]")
        u = torch.empty((B, 2), dtype=dt_, device=X.device)
        rho = torch.empty((B, 2 * self.horizon), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        z = torch.empty((B, 2 * self.horizon), dtype=dt_, device=X.device) if want_z else None
        p = make_od_params(self.robot_spec, self._mc, self.cbf_param, self.horizon, self.dt, self.robot_spec["radius"], self.io_dtype,
                           obs_shared=shared, tol=self.tol, max_iter=self.max_iter)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        sl = self.slices_for(lambda: self._lib.sc_odmpcgn_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(), u.data_ptr(), rho.data_ptr(),
                status.data_ptr(), iters.data_ptr(), z.data_ptr() if z is not None else None, stream)
        rc = self._lib.sc_odmpcgn_solve_batch(C.byref(p), *args) if sl is None else self._lib.sc_odmpcgn_solve_batch_sliced(C.byref(p), C.byref(sl), *args)
        _lib.check(rc, "sc_odmpcgn_solve_batch")
        return (u, rho, status, iters, z) if want_z else (u, rho, status, iters)


class OptimalDecayGnMPCCBF:
    """Drop-in for position_control.optimal_decay_mpc_cbf.OptimalDecayMPCCBF with a KinematicBicycle2D or Quad2D robot (single
    agent per call; the one NLP goes through the batched entry point on device ``device``)."""

    def __init__(self, robot, robot_spec, num_obs=5, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.status = "optimal"                               # optimal_decay_mpc_cbf.py:21
        self.num_obs = int(num_obs)
        self.device = device
        self.horizon = 10                                     # :24
        self.dt = robot.dt
        self._mc = od_model_constants(self.robot_spec)
        self.Q, self.R = np.diag(self._mc["Q"]), np.array(self._mc["R"])
        self.n_states, self.n_controls = self._mc["nx"], 2
        self.cbf_param = dict(self._mc["cbf_param"])
        self.omega1 = None                                    # :92-93
        self.omega2 = None
        self.goal = np.array([0, 0])
        self.obs = None
        self.setup_control_problem()

    def setup_control_problem(self):
        self._ctl = BatchedOptimalDecayGnMPCCBF(self.robot_spec, dt=self.dt, io_dtype="f64", horizon=self.horizon, cbf_param=self.cbf_param)
        self.u_prev = np.zeros(2)
        self.z = np.zeros(2 * self.horizon)
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
        nx = self._mc["nx"]
        X = np.zeros(nx)
        xs = np.asarray(robot_state, dtype=np.float64).reshape(-1)[:nx]
        X[: xs.shape[0]] = xs
        dev = torch.device("cuda", int(self.device))
        t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), dtype=torch.float64, device=dev)
        self._ctl.cbf_param = self.cbf_param                  # (users mutate cbf_param in place: README "online adaptive CBF")
        u, rho, st, it, z = self._ctl.solve(t(X[None]), t(self.u_prev[None]), t(np.asarray(self.goal, dtype=np.float64).reshape(-1)[None, :2]),
                                            t(self.obs[None]), want_z=True)
        self.iterations = int(it[0].item())
        self.solver_status = _lib.STATUS_STRINGS[int(st[0].item())]
        self.rho, self.z = rho[0].cpu().numpy(), z[0].cpu().numpy()
        self.omega1, self.omega2 = float(self.rho[0]), float(self.rho[1])
        self.u_prev = u[0].cpu().numpy().copy()
        return self.u_prev.reshape(-1, 1).copy()
