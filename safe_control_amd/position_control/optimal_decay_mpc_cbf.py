"""Optimal-decay MPC-CBF position controllers backed by the gfx950 HIP kernel (csrc/mpc_cbf.hip, OD variant).

``OptimalDecayMPCCBF`` keeps the plugin surface of the reference class of the same name
(position_control/optimal_decay_mpc_cbf.py:15-330): ``__init__(robot, robot_spec)``,
``setup_control_problem()``, ``update_tvp(goal, obs)``, ``solve_control_problem(robot_state, control_ref,
nearest_obs)``, attributes ``status``, ``cbf_param`` (alpha1/alpha2, omega1/omega2, p_sb1/p_sb2), ``horizon``,
``Q``, ``R``, ``goal``, ``obs``, ``omega1``, ``omega2``.  The reference copy is stale (five
5-wide obstacle slots, selected by a string `tracking.py` no longer lists): here obstacles are the 7-wide rows
of MPCCBF and ``num_obs`` is a parameter (default 5 like the reference's five slots).  DynamicUnicycle2D here;
KinematicBicycle2D and Quad2D in optimal_decay_mpc_cbf_gn.py (same class name through ``__new__``).

``BatchedOptimalDecayMPCCBF`` solves B agents' NLPs in one launch.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec
from .mpc_cbf import default_mpc_weights, make_params, pad_obstacles


def default_od_mpc_param(model, extension=False):
    """optimal_decay_mpc_cbf.py:54-91.  ``extension``: BASELINE config 5 asks for optimal decay on models the reference
    class rejects (Unicycle2D, :19-20) or serves with the plain row (Quad3D, :284-287); the build-defined semantics are
    the rel-degree-1 row of the reference's optimal-decay CBF-QP, d_h + alpha omega1 h_k (optimal_decay_cbf_qp.py:96-101),
    with MPCCBF's gain for the model (mpc_cbf.py:52-53) -- see oracle/od_mpc_rd1.py."""
    if model == "DynamicUnicycle2D":
        return {"alpha1": 0.01, "alpha2": 0.01, "omega1": 1.0, "p_sb1": 10.0, "omega2": 1.0, "p_sb2": 10.0}
    if model == "Unicycle2D" and extension:
        return {"alpha": 0.05, "omega1": 1.0, "p_sb1": 10.0, "omega2": 1.0, "p_sb2": 10.0}
    raise NotImplementedError(f"optimal-decay MPC-CBF on this kernel supports DynamicUnicycle2D (and Unicycle2D with extension=True); "
                              f"KinematicBicycle2D / Quad2D: BatchedOptimalDecayGnMPCCBF; Quad3D: BatchedOptimalDecayLinearMPCCBF; not {model}")


def make_od_mpc_params(robot_spec, cbf_param, Q, R, horizon, dt, radius, io_dtype, obs_shared=False, tol=1e-6,
                       max_iter=_lib.IPOPT_MAX_ITER, slack_reset=0):
    p = _lib.OdMpcCbfParams()
    p.mpc = make_params(robot_spec, cbf_param, Q, R, horizon, dt, radius, io_dtype, obs_shared=obs_shared, tol=tol,
                        max_iter=max_iter, slack_reset=slack_reset)
    p.omega_ref[0] = float(cbf_param.get("omega1", 1.0))
    p.omega_ref[1] = float(cbf_param.get("omega2", 1.0))
    p.p_sb[0] = float(cbf_param.get("p_sb1", 10.0))
    p.p_sb[1] = float(cbf_param.get("p_sb2", 10.0))
    return p


class OptimalDecayMPCCBF:
    """Drop-in for position_control.optimal_decay_mpc_cbf.OptimalDecayMPCCBF (single agent per call)."""

    def __new__(cls, robot, robot_spec, *args, **kwargs):
        # the reference serves every model of its accept list from this one class (:19); KinematicBicycle2D and Quad2D run on the
        # step()-barrier kernel
        if cls is OptimalDecayMPCCBF and robot_spec.get("model") in ("KinematicBicycle2D", "Quad2D"):
            from .optimal_decay_mpc_cbf_gn import OptimalDecayGnMPCCBF
            return OptimalDecayGnMPCCBF(robot, robot_spec, *args, **kwargs)
        if cls is OptimalDecayMPCCBF and robot_spec.get("model") == "Quad3D":        # the plain row with R u^2 (:284-287)
            from .mpc_cbf_linear import OptimalDecayLinearMPCCBF
            return OptimalDecayLinearMPCCBF(robot, robot_spec, *args, **kwargs)
        if cls is OptimalDecayMPCCBF and robot_spec.get("model") == "VTOL2D":        # one NLP per wavefront, one stage per lane
            from .mpc_cbf_vtol import OptimalDecayVtolMPCCBF
            return OptimalDecayVtolMPCCBF(robot, robot_spec, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, robot, robot_spec, num_obs=5, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.status = "optimal"                             # optimal_decay_mpc_cbf.py:21
        self.num_obs = int(num_obs)
        self.device = device
        self.horizon = int(self.robot_spec.get("mpc_horizon", 10))       # :24
        self.dt = robot.dt
        model = self.robot_spec["model"]
        self.Q, self.R = default_mpc_weights(model)          # :31-33, same as MPCCBF
        self.n_controls, self.n_states = 2, 4
        self.cbf_param = default_od_mpc_param(model)
        self.omega1 = None                                   # :92-93
        self.omega2 = None
        self.goal = np.array([0, 0])
        self.obs = None
        self.setup_control_problem()

    def setup_control_problem(self):
        if not 1 <= self.horizon <= _lib.MPCCBF_MAX_HORIZON:
            raise ValueError(f"mpc_horizon must be in [1, {_lib.MPCCBF_MAX_HORIZON}]")
        self._lib = _lib.load()
        self.u_prev = np.zeros(2, dtype=np.float64)
        self.z = np.zeros(2 * self.horizon, dtype=np.float64)
        self.rho = np.ones(2 * self.horizon, dtype=np.float64)
        self.iterations = 0
        self.solver_status = "optimal"

    def update_tvp(self, goal, obs):
        self.goal = np.array(goal)
        self.obs = pad_obstacles(obs, self.num_obs)

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        self.update_tvp(control_ref["goal"], nearest_obs)
        if control_ref["state_machine"] != "track":
            return control_ref["u_ref"]
        X = np.ascontiguousarray(np.asarray(robot_state, dtype=np.float64).reshape(-1)[:4])
        g = np.ascontiguousarray(np.asarray(self.goal, dtype=np.float64).reshape(-1)[:2])
        obs = np.ascontiguousarray(self.obs, dtype=np.float64)
        p = make_od_mpc_params(self.robot_spec, self.cbf_param, self.Q, self.R, self.horizon, self.dt,
                               self.robot.robot_radius, _lib.DTYPE_F64)
        u = np.zeros(2); st = np.zeros(1, dtype=np.int32); it = np.zeros(1, dtype=np.int32)
        rc = self._lib.sc_odmpccbf_solve_batch_host(
            C.byref(p), 1, self.num_obs, X.ctypes.data, self.u_prev.ctypes.data, g.ctypes.data, obs.ctypes.data,
            u.ctypes.data, self.rho.ctypes.data, st.ctypes.data, it.ctypes.data, self.z.ctypes.data, int(self.device))
        _lib.check(rc, "sc_odmpccbf_solve_batch_host")
        self.iterations = int(it[0])
        self.solver_status = _lib.STATUS_STRINGS[int(st[0])]
        self.omega1, self.omega2 = float(self.rho[0]), float(self.rho[1])
        self.u_prev = u.copy()
        return u.reshape(-1, 1).copy()


class BatchedOptimalDecayMPCCBF(_lib.SlicedSolver):
    """Optimal-decay MPC-CBF for B agents per launch on device tensors.

    ``solve(X[B,4], u_prev[B,2], goal[B,2], obs[B,K,7] | obs[K,7])`` -> ``u[B,2]``, ``rho[B,2N]`` (omega1_k, omega2_k
    per stage), ``status[B] int32``, ``iters[B] int32`` (and ``z[B,2N]`` if asked).  ``extension=True`` also serves
    Unicycle2D (states padded to 4 columns like everywhere in the batched engine; omega2_k is inert and stays at its
    reference) -- BASELINE config 5, no reference counterpart.
    """

    def __init__(self, robot_spec, dt=0.05, io_dtype="f32", horizon=None, cbf_param=None, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                 extension=False, iter_slices=None, classify_first=True, order=True):
        self.init_slices(iter_slices, classify_first, order)     # continuation launches (round 5): first cap 100, the rest of the budget behind it
        self.robot_spec = complete_robot_spec(robot_spec)
        model = self.robot_spec["model"]
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = int(horizon if horizon is not None else self.robot_spec.get("mpc_horizon", 10))
        self.Q, self.R = default_mpc_weights(model)
        self.extension = bool(extension)
        self.cbf_param = cbf_param or default_od_mpc_param(model, extension=self.extension)
        self.nx = 3 if model == "Unicycle2D" else 4
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
        if X.shape != (B, 4) or u_prev.shape != (B, 2) or goal.shape != (B, 2) or obs.shape[-1] != 7 \
                or (not shared and obs.shape[0] != B):
            raise ValueError("expected X[B,4], u_prev[B,2], goal[B,2], obs[B,K,7] or obs[K,7]")
        u = torch.empty((B, 2), dtype=dt_, device=X.device)
        rho = torch.empty((B, 2 * self.horizon), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        z = torch.empty((B, 2 * self.horizon), dtype=dt_, device=X.device) if want_z else None
        p = make_od_mpc_params(self.robot_spec, self.cbf_param, self.Q, self.R, self.horizon, self.dt,
                               self.robot_spec["radius"], self.io_dtype, obs_shared=shared, tol=self.tol,
                               max_iter=self.max_iter,
                               slack_reset=2 if (self.extension and self.robot_spec["model"] == "Unicycle2D") else 0)   # oracle/od_mpc_rd1.py
        stream = torch.cuda.current_stream(X.device).cuda_stream
        sl = self.slices_for(lambda: self._lib.sc_odmpccbf_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(), u.data_ptr(),
                rho.data_ptr(), status.data_ptr(), iters.data_ptr(), z.data_ptr() if z is not None else None, stream)
        rc = self._lib.sc_odmpccbf_solve_batch(C.byref(p), *args) if sl is None else self._lib.sc_odmpccbf_solve_batch_sliced(C.byref(p), C.byref(sl), *args)
        _lib.check(rc, "sc_odmpccbf_solve_batch")
        return (u, rho, status, iters, z) if want_z else (u, rho, status, iters)
