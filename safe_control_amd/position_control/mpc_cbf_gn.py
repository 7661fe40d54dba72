"""MPC-CBF for DoubleIntegrator2D, Quad2D and the KinematicBicycle2D family (HOCBF, C3BF, DPCBF) on the gfx950 kernel csrc/mpc_gn.hip.

``safe_control_amd.MPCCBF(robot, robot_spec, ...)`` returns a ``GnMPCCBF`` for these models (the reference serves every
model from the one MPCCBF class, position_control/mpc_cbf.py:7-100); ``BatchedGnMPCCBF`` solves B agents per launch on
device tensors.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec
from .mpc_cbf import apply_mpc_overrides, pad_obstacles

GN_MODELS = ("DoubleIntegrator2D", "Quad2D", "KinematicBicycle2D", "KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF")


def model_constants(robot_spec):
    """Weights, gains and bounds of MPCCBF.__init__ / create_mpc for the model (mpc_cbf.py:28-36, :60-76, :193-216) and the
    barrier inflation of its agent_barrier_dt (double_integrator2D.py:222, quad2D.py:179: 1.01)."""
    m = robot_spec["model"]
    if m == "DoubleIntegrator2D":
        return dict(nx=4, Q=[50.0, 50.0, 20.0, 20.0], R=[0.5, 0.5], cbf_param={"alpha1": 0.2, "alpha2": 0.2}, beta=1.01,
                    u_lo=[-robot_spec["ax_max"], -robot_spec["ay_max"]], u_hi=[robot_spec["ax_max"], robot_spec["ay_max"]],
                    circles_only=False)
    if m == "Quad2D":
        return dict(nx=6, Q=[25.0, 25.0, 50.0, 10.0, 10.0, 50.0], R=[0.5, 0.5], cbf_param={"alpha1": 0.15, "alpha2": 0.15},
                    beta=1.01, u_lo=[robot_spec["f_min"]] * 2, u_hi=[robot_spec["f_max"]] * 2, circles_only=True)
    if m == "KinematicBicycle2D":               # mpc_cbf.py:31-33,64-67,205-211; barrier inflation kinematic_bicycle2D.py:175 (1.1)
        return dict(nx=4, Q=[50.0, 50.0, 1.0, 1.0], R=[0.5, 5000.0], cbf_param={"alpha1": 0.1, "alpha2": 0.1}, beta=1.1,
                    u_lo=[-robot_spec["a_max"], -robot_spec["beta_max"]], u_hi=[robot_spec["a_max"], robot_spec["beta_max"]],
                    circles_only=True, slack_reset=2)
    if m in ("KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"):
        # mpc_cbf.py:31-33,68-73,205-211: one gain, row d_h + alpha h_k (:312-315); the barrier's own inflation (1.01 / 1.05) is fixed in
        # kinematic_bicycle2D_c3bf.py:77 / _dpcbf.py:86 and in the kernel
        return dict(nx=4, Q=[50.0, 50.0, 1.0, 1.0], R=[0.5, 5000.0], cbf_param={"alpha": 0.15}, beta=1.01 if m.endswith("C3BF") else 1.05,
                    u_lo=[-robot_spec["a_max"], -robot_spec["beta_max"]], u_hi=[robot_spec["a_max"], robot_spec["beta_max"]],
                    circles_only=True, slack_reset=2)
    raise NotImplementedError(m)


def make_params(robot_spec, mc, cbf_param, horizon, dt, radius, io_dtype, obs_shared=False, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                mu_init=0.1, mu_min=1e-9, acceptable_tol=1e-5, resto=None):
    p = _lib.MpcGnParams()
    p.model_id = _lib.MODEL_IDS[robot_spec["model"]]
    p.io_dtype = io_dtype
    p.horizon = int(horizon)
    p.max_iter = int(max_iter)
    p.obs_shared = 1 if obs_shared else 0
    p.circles_only = 1 if mc["circles_only"] else 0
    p.slack_reset = int(mc.get("slack_reset", 0))         # the bicycles: 2 (oracle/mpc_gn.py: kb_model)
    p.dt = float(dt)
    for i, v in enumerate(mc["Q"]):
        p.Q[i] = float(v)
    for i in range(2):
        p.R[i], p.u_lo[i], p.u_hi[i] = float(mc["R"][i]), float(mc["u_lo"][i]), float(mc["u_hi"][i])
    if "alpha" in cbf_param:                              # rel-degree-1 barriers: the one gain travels in alpha1
        p.alpha1, p.alpha2 = float(cbf_param["alpha"]), 0.0
    else:
        p.alpha1, p.alpha2 = float(cbf_param["alpha1"]), float(cbf_param["alpha2"])
    p.v_min = float(robot_spec.get("v_min", 0.0))
    p.v_max = float(robot_spec.get("v_max", 0.0))
    p.rear_ax_dist = float(robot_spec.get("rear_ax_dist", 0.0))
    p.mass = float(robot_spec.get("mass", 0.0))
    p.inertia = float(robot_spec.get("inertia", 0.0))
    p.robot_radius = float(radius)
    p.beta = float(mc["beta"])
    p.tol, p.acceptable_tol, p.mu_init, p.mu_min = float(tol), float(acceptable_tol), float(mu_init), float(mu_min)
    p.resto = resto if resto is not None else _lib.default_resto()     # feasibility restoration (sc_resto_params)
    return p


class GnMPCCBF:
    """Drop-in for position_control.mpc_cbf.MPCCBF with a DoubleIntegrator2D, Quad2D or KinematicBicycle2D robot."""

    def __init__(self, robot, robot_spec, show_mpc_traj=False, num_obs=5, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.status = "optimal"                               # mpc_cbf.py:10
        self.show_mpc_traj = show_mpc_traj
        self.num_obs = int(num_obs)
        self.device = device
        self.horizon = int(self.robot_spec.get("mpc_horizon", 10))
        self.dt = robot.dt
        self._mc = model_constants(self.robot_spec)
        self.Q, self.R = np.diag(self._mc["Q"]), np.array(self._mc["R"])
        self.n_states, self.n_controls = self._mc["nx"], 2
        self.goal = np.array([0, 0])
        self.cbf_param = apply_mpc_overrides(dict(self._mc["cbf_param"]), self.robot_spec)
        self.obs = None
        self.setup_control_problem()

    def setup_control_problem(self):
        if not 1 <= self.horizon <= 32:
            raise ValueError("mpc_horizon must be in [1, 32]")
        self._lib = _lib.load()
        self.u_prev = np.zeros(2)
        self.z = np.zeros(2 * self.horizon)
        self.iterations = 0
        self.solver_status = "optimal"
        # DoubleIntegrator2D: the NLP as do-mpc poses it (multiple shooting under IPOPT's algorithm, csrc/mpc_du_ms.hip, kernel 13) unless
        # robot_spec['mpc_formulation'] = 'condensed'; superellipsoid rows run on the condensed kernel.  KinematicBicycle2D: the same kernel ON
        # REQUEST (robot_spec['mpc_formulation'] = 'multiple_shooting'): where a plan slows down to v_min the kink of robot.step's speed clip sits
        # on the solution and the Newton iteration cycles to the iteration limit (3 % of the bench draws; DESIGN.md kernel 13), which the
        # condensed solve's l1 merit function does not
        self._ms = None
        want = self.robot_spec.get("mpc_formulation", "multiple_shooting" if self.robot_spec["model"] == "DoubleIntegrator2D" else "condensed")
        if self.robot_spec["model"] in ("DoubleIntegrator2D", "KinematicBicycle2D") and want == "multiple_shooting" and self.num_obs <= 16:
            from .mpc_cbf_ms import BatchedMSMPCCBF
            self._ms = BatchedMSMPCCBF(self.robot_spec, dt=self.dt, io_dtype="f64", horizon=self.horizon, cbf_param=self.cbf_param, check_circles=False)

    def update_tvp(self, goal, obs):
        self.goal = np.array(goal)
        self.obs = pad_obstacles(obs, self.num_obs)

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        self.update_tvp(control_ref["goal"], nearest_obs)
        if control_ref["state_machine"] != "track":           # mpc_cbf.py:379-381
            return control_ref["u_ref"]
        nx = self._mc["nx"]
        X = np.zeros(nx)
        xs = np.asarray(robot_state, dtype=np.float64).reshape(-1)[:nx]
        X[: xs.shape[0]] = xs
        g = np.ascontiguousarray(np.asarray(self.goal, dtype=np.float64).reshape(-1)[:2])
        obs = np.ascontiguousarray(self.obs, dtype=np.float64)
        se = bool((obs[:, 6] >= 0.5).any())
        if self._ms is not None and (not se or self.robot_spec["model"] == "DoubleIntegrator2D"):      # (superellipsoid rows: csrc/mpc_du_ms_se.hip)
            import torch
            self._ms.superellipsoids = se
            dev = torch.device("cuda", int(self.device))
            t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), dtype=torch.float64, device=dev)     # noqa: E731
            self._ms.cbf_param = self.cbf_param
            self._ms.robot_spec["radius"] = self.robot.robot_radius
            u, st, it, plan = self._ms.solve(t(X[None]), t(self.u_prev[None]), t(g[None]), t(obs[None]), want_plan=True)
            self.iterations = int(it[0].item())
            self.solver_status = _lib.STATUS_STRINGS[int(st[0].item())]
            self.z = plan[0, (self.horizon + 1) * 4:].cpu().numpy().copy()
            self.u_prev = u[0].cpu().numpy().copy()
            return self.u_prev.reshape(-1, 1).copy()
        p = make_params(self.robot_spec, self._mc, self.cbf_param, self.horizon, self.dt, self.robot.robot_radius, _lib.DTYPE_F64)
        u = np.zeros(2); st = np.zeros(1, dtype=np.int32); it = np.zeros(1, dtype=np.int32)
        rc = self._lib.sc_mpcgn_solve_batch_host(
            C.byref(p), 1, self.num_obs, X.ctypes.data, self.u_prev.ctypes.data, g.ctypes.data, obs.ctypes.data,
            u.ctypes.data, st.ctypes.data, it.ctypes.data, self.z.ctypes.data, int(self.device))
        _lib.check(rc, "sc_mpcgn_solve_batch_host")
        self.iterations = int(it[0])
        self.solver_status = _lib.STATUS_STRINGS[int(st[0])]
        self.u_prev = u.copy()
        return u.reshape(-1, 1).copy()


class BatchedGnMPCCBF(_lib.SlicedSolver):
    """``solve(X[B,nx], u_prev[B,2], goal[B,2], obs[B,K,7] | obs[K,7])`` -> ``u[B,2]``, ``status[B]``, ``iters[B]`` (and
    ``z[B,2N]`` if asked); nx = 4 (DoubleIntegrator2D, KinematicBicycle2D) or 6 (Quad2D)."""

    def __init__(self, robot_spec, dt=0.05, io_dtype="f64", horizon=None, cbf_param=None, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                 iter_slices=None, classify_first=True, order=True):
        self.init_slices(iter_slices, classify_first, order)      # continuation launches (include/safe_control_amd.h: sc_mpc_slices)
        self.robot_spec = complete_robot_spec(robot_spec)
        if self.robot_spec["model"] not in GN_MODELS:
            raise NotImplementedError(f"this controller serves {GN_MODELS}")
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = int(horizon if horizon is not None else self.robot_spec.get("mpc_horizon", 10))
        self._mc = model_constants(self.robot_spec)
        self.Q, self.R = np.diag(self._mc["Q"]), np.array(self._mc["R"])
        self.cbf_param = cbf_param or apply_mpc_overrides(dict(self._mc["cbf_param"]), self.robot_spec)
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
        nx = self._mc["nx"]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if X.shape != (B, nx) or u_prev.shape != (B, 2) or goal.shape != (B, 2) or obs.shape[-1] != 7 \
                or (not shared and obs.shape[0] != B):
            raise ValueError(f"expected X[B,{nx}], u_prev[B,2], goal[B,2], obs[B,K,7] or obs[K,7]")
        u = torch.empty((B, 2), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        z = torch.empty((B, 2 * self.horizon), dtype=dt_, device=X.device) if want_z else None
        p = make_params(self.robot_spec, self._mc, self.cbf_param, self.horizon, self.dt, self.robot_spec["radius"],
                        self.io_dtype, obs_shared=shared, tol=self.tol, max_iter=self.max_iter, resto=getattr(self, "resto", None))
        stream = torch.cuda.current_stream(X.device).cuda_stream
        sl = self.slices_for(lambda: self._lib.sc_mpcgn_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(), u.data_ptr(),
                status.data_ptr(), iters.data_ptr(), z.data_ptr() if z is not None else None, stream)
        if sl is None:
            rc = self._lib.sc_mpcgn_solve_batch(C.byref(p), *args)
        else:
            rc = self._lib.sc_mpcgn_solve_batch_sliced(C.byref(p), C.byref(sl), *args)
        _lib.check(rc, "sc_mpcgn_solve_batch")
        return (u, status, iters, z) if want_z else (u, status, iters)
