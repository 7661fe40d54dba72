"""MPC-CBF for the reference's linear robot models (SingleIntegrator2D, Quad3D) on the gfx950 kernel csrc/mpc_lin.hip.

``safe_control_amd.MPCCBF(robot, robot_spec, ...)`` returns a ``LinearMPCCBF`` for these models (the reference serves
every model from the one MPCCBF class, position_control/mpc_cbf.py:7-100); ``BatchedLinearMPCCBF`` solves B agents per
launch on device tensors.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.linear_models import LINEAR_MODELS, linear_model
from ..robots.spec import complete_robot_spec
from .mpc_cbf import apply_mpc_overrides, pad_obstacles


def make_params(mdl, cbf_param, horizon, radius, io_dtype, obs_shared=False, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER, mu_init=0.1,
                mu_min=1e-9, acceptable_tol=1e-5, resto=None, input_rterm="du", slack_reset=2):
    """``input_rterm``: "du" = MPCCBF's do-mpc delta-u penalty (mpc_cbf.py:180); "u2" = the R u^2 expression of the reference's
    OptimalDecayMPCCBF (optimal_decay_mpc_cbf.py:173-179) with the plain row that class gives Quad3D (:284-287): sc_mpclin_params.optimal_decay = 2."""
    p = _lib.MpcLinParams()
    p.optimal_decay = {"du": 0, "u2": 2}[input_rterm]
    p.slack_reset = int(slack_reset)                          # line search of the regular phase (oracle/mpc_lin.py: params)
    p.io_dtype = io_dtype
    p.nx, p.nu, p.ng = mdl["nx"], mdl["nu"], mdl["ng"]
    p.horizon = int(horizon)
    p.max_iter = int(max_iter)
    p.obs_shared = 1 if obs_shared else 0
    p.circles_only = 1 if mdl["circles_only"] else 0
    p.alpha = float(cbf_param["alpha"])
    p.robot_radius = float(radius)
    p.beta = 1.01                                             # agent_barrier_dt default (single_integrator2D.py:148, quad3D.py:275)
    p.tol, p.acceptable_tol, p.mu_init, p.mu_min = float(tol), float(acceptable_tol), float(mu_init), float(mu_min)
    qd = np.diag(mdl["Q"]) if np.ndim(mdl["Q"]) == 2 else np.asarray(mdl["Q"])
    for i in range(mdl["nx"]):
        p.Q[i] = float(qd[i])
    for i in range(mdl["nu"]):
        p.R[i], p.u_lo[i], p.u_hi[i] = float(mdl["R"][i]), float(mdl["u_lo"][i]), float(mdl["u_hi"][i])
    p.resto = resto if resto is not None else _lib.default_resto()     # feasibility restoration (sc_resto_params)
    return p


def build_model_blob(lib, p, mdl):
    """Constant matrices of the condensed problem (host, float64): sc_mpclin_build_model."""
    nd = lib.sc_mpclin_model_doubles(p.nx, p.nu, p.horizon)
    if nd == 0:
        raise ValueError("unsupported dimensions: need nx <= 12, nu <= 4, nu * horizon <= 128")
    blob = np.zeros(nd, dtype=np.float64)
    mats = [np.ascontiguousarray(mdl[k], dtype=np.float64) for k in ("Ae", "Be", "As", "Bs")]
    rc = lib.sc_mpclin_build_model(C.byref(p), *[m.ctypes.data for m in mats], blob.ctypes.data)
    _lib.check(rc, "sc_mpclin_build_model")
    return blob


class LinearMPCCBF:
    """Drop-in for position_control.mpc_cbf.MPCCBF with a SingleIntegrator2D or Quad3D robot (single agent per call)."""
    input_rterm = "du"

    def __init__(self, robot, robot_spec, show_mpc_traj=False, num_obs=5, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.status = "optimal"                               # mpc_cbf.py:10
        self.show_mpc_traj = show_mpc_traj
        self.num_obs = int(num_obs)
        self.device = device
        self.horizon = int(self.robot_spec.get("mpc_horizon", 10))
        self.dt = robot.dt
        self._mdl = linear_model(self.robot_spec, self.dt)
        self.Q, self.R = self._mdl["Q"], self._mdl["R"]
        self.n_states, self.n_controls = self._mdl["nx"], self._mdl["nu"]
        self.goal = np.zeros(self._mdl["ng"])                 # mpc_cbf.py:45, :81
        self.cbf_param = apply_mpc_overrides(dict(self._mdl["cbf_param"]), self.robot_spec)
        self.obs = None
        self.setup_control_problem()

    def setup_control_problem(self):
        self._lib = _lib.load()
        p = make_params(self._mdl, self.cbf_param, self.horizon, self.robot.robot_radius, _lib.DTYPE_F64, input_rterm=self.input_rterm)
        self._blob = build_model_blob(self._lib, p, self._mdl)
        self.u_prev = np.zeros(self.n_controls)
        self.z = np.zeros(self.n_controls * self.horizon)
        self.iterations = 0
        self.solver_status = "optimal"
        # SingleIntegrator2D under MPCCBF: the NLP as do-mpc poses it (multiple shooting under IPOPT's algorithm, csrc/mpc_du_ms.hip, kernel 13) on
        # request -- robot_spec['mpc_formulation'] = 'multiple_shooting'; circles only (a scene with superellipsoid rows runs on this class's kernel)
        self._ms = None
        if self._mdl["nx"] == 2 and self.input_rterm == "du" and self.robot_spec.get("mpc_formulation", "condensed") == "multiple_shooting" \
                and self.horizon <= 62 and self.num_obs <= 16:
            from .mpc_cbf_ms import BatchedMSMPCCBF
            self._ms = BatchedMSMPCCBF(self.robot_spec, dt=self.dt, io_dtype="f64", horizon=self.horizon, cbf_param=self.cbf_param, check_circles=False)

    def update_tvp(self, goal, obs):
        self.goal = np.array(goal)
        self.obs = pad_obstacles(obs, self.num_obs)

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        self.update_tvp(control_ref["goal"], nearest_obs)
        if control_ref["state_machine"] != "track":           # mpc_cbf.py:379-381
            return control_ref["u_ref"]
        nx, nu, ng = self._mdl["nx"], self._mdl["nu"], self._mdl["ng"]
        X = np.zeros(nx)
        xs = np.asarray(robot_state, dtype=np.float64).reshape(-1)[:nx]
        X[: xs.shape[0]] = xs
        g = np.zeros(ng)
        gs = np.asarray(self.goal, dtype=np.float64).reshape(-1)[:ng]
        g[: gs.shape[0]] = gs
        obs = np.ascontiguousarray(self.obs, dtype=np.float64)
        if self._ms is not None and not (obs[:, 6] >= 0.5).any():
            import torch
            dev = torch.device("cuda", int(self.device))
            t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), dtype=torch.float64, device=dev)     # noqa: E731
            self._ms.cbf_param = self.cbf_param
            self._ms.robot_spec["radius"] = self.robot.robot_radius
            um, sm, im, plan = self._ms.solve(t(X[None]), t(self.u_prev[None]), t(g[None]), t(obs[None]), want_plan=True)
            self.iterations = int(im[0].item())
            self.solver_status = _lib.STATUS_STRINGS[int(sm[0].item())]
            self.z = plan[0, (self.horizon + 1) * 4:].cpu().numpy().copy()
            self.u_prev = um[0].cpu().numpy().copy()
            return self.u_prev.reshape(-1, 1).copy()
        p = make_params(self._mdl, self.cbf_param, self.horizon, self.robot.robot_radius, _lib.DTYPE_F64, input_rterm=self.input_rterm)
        u = np.zeros(nu); st = np.zeros(1, dtype=np.int32); it = np.zeros(1, dtype=np.int32)
        rc = self._lib.sc_mpclin_solve_batch_host(
            C.byref(p), self._blob.ctypes.data, 1, self.num_obs, X.ctypes.data, self.u_prev.ctypes.data, g.ctypes.data,
            obs.ctypes.data, u.ctypes.data, st.ctypes.data, it.ctypes.data, self.z.ctypes.data, int(self.device))
        _lib.check(rc, "sc_mpclin_solve_batch_host")
        self.iterations = int(it[0])
        self.solver_status = _lib.STATUS_STRINGS[int(st[0])]
        self.u_prev = u.copy()
        return u.reshape(-1, 1).copy()


class OptimalDecayLinearMPCCBF(LinearMPCCBF):
    """``OptimalDecayMPCCBF(robot, {'model': 'Quad3D'})`` of the reference (optimal_decay_mpc_cbf.py:19 accepts the model): its
    compute_cbf_constraint gives Quad3D the PLAIN row d_h + alpha h_k (:284-287) -- the decay inputs omega1, omega2 are part of the
    model (:123-124) but touch no row, and their penalty p_sb (omega - 1)^2 keeps them at 1 -- with that class's weights (:41-43: the
    same Q, R as MPCCBF), gain 0.15 (:78-82), bounds u_min .. u_max (:211-215) and its input term R u^2 (:173-179).  Extension label as
    for every optimal-decay class: the reference copy is stale (five 5-wide obstacle slots), parity is against oracle/mpc_lin.py
    (rterm = "u2").  ``omega1`` / ``omega2`` report the inert decay inputs."""
    input_rterm = "u2"

    def __init__(self, robot, robot_spec, num_obs=5, device=0):
        super().__init__(robot, robot_spec, show_mpc_traj=False, num_obs=num_obs, device=device)
        self.cbf_param.update({"omega1": 1.0, "p_sb1": 10.0, "omega2": 1.0, "p_sb2": 10.0})     # :88-91
        self.omega1 = None
        self.omega2 = None

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        u = super().solve_control_problem(robot_state, control_ref, nearest_obs)
        if control_ref["state_machine"] == "track":
            self.omega1, self.omega2 = float(self.cbf_param["omega1"]), float(self.cbf_param["omega2"])
        return u


class BatchedLinearMPCCBF(_lib.SlicedSolver):
    """``solve(X[B,nx], u_prev[B,nu], goal[B,ng], obs[B,K,7] | obs[K,7])`` -> ``u[B,nu]``, ``status[B]``, ``iters[B]``
    (and ``z[B, nu*N]`` if asked) for SingleIntegrator2D (nx 2, nu 2, ng 2) or Quad3D (nx 12, nu 4, ng 3).
    ``iter_slices`` / ``classify_first`` / ``order``: continuation launches (include/safe_control_amd.h: sc_mpc_slices)."""

    def __init__(self, robot_spec, dt=0.05, io_dtype="f64", horizon=None, cbf_param=None, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                 iter_slices=None, classify_first=True, order=True, input_rterm="du"):
        self.init_slices(iter_slices, classify_first, order)
        self.input_rterm = input_rterm                        # "u2": OptimalDecayMPCCBF's semantics for Quad3D (make_params)
        self.robot_spec = complete_robot_spec(robot_spec)
        if self.robot_spec["model"] not in LINEAR_MODELS:
            raise NotImplementedError(f"linear-model MPC-CBF supports {LINEAR_MODELS}")
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = int(horizon if horizon is not None else self.robot_spec.get("mpc_horizon", 10))
        self._mdl = linear_model(self.robot_spec, self.dt)
        self.Q, self.R = self._mdl["Q"], self._mdl["R"]
        self.cbf_param = cbf_param or apply_mpc_overrides(dict(self._mdl["cbf_param"]), self.robot_spec)
        self.tol, self.max_iter = tol, max_iter
        self._lib = _lib.load()
        p = make_params(self._mdl, self.cbf_param, self.horizon, self.robot_spec["radius"], self.io_dtype, input_rterm=self.input_rterm)
        self._blob_host = build_model_blob(self._lib, p, self._mdl)
        self._blob_dev = {}

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def _blob(self, device):
        import torch
        key = str(device)
        if key not in self._blob_dev:
            self._blob_dev[key] = torch.tensor(self._blob_host, dtype=torch.float64, device=device)
        return self._blob_dev[key]

    def solve(self, X, u_prev, goal, obs, want_z=False):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_prev", u_prev), ("goal", goal), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        nx, nu, ng = self._mdl["nx"], self._mdl["nu"], self._mdl["ng"]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if X.shape != (B, nx) or u_prev.shape != (B, nu) or goal.shape != (B, ng) or obs.shape[-1] != 7 \
                or (not shared and obs.shape[0] != B):
            raise ValueError(f"expected X[B,{nx}], u_prev[B,{nu}], goal[B,{ng}], obs[B,K,7] or obs[K,7]")
        u = torch.empty((B, nu), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        z = torch.empty((B, nu * self.horizon), dtype=dt_, device=X.device) if want_z else None
        p = make_params(self._mdl, self.cbf_param, self.horizon, self.robot_spec["radius"], self.io_dtype,
                        obs_shared=shared, tol=self.tol, max_iter=self.max_iter, resto=getattr(self, "resto", None),
                        input_rterm=self.input_rterm)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        sl = self.slices_for(lambda: self._lib.sc_mpclin_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (self._blob(X.device).data_ptr(), B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(),
                obs.data_ptr(), u.data_ptr(), status.data_ptr(), iters.data_ptr(), z.data_ptr() if z is not None else None, stream)
        if sl is None:
            rc = self._lib.sc_mpclin_solve_batch(C.byref(p), *args)
        else:
            rc = self._lib.sc_mpclin_solve_batch_sliced(C.byref(p), C.byref(sl), *args)
        _lib.check(rc, "sc_mpclin_solve_batch")
        return (u, status, iters, z) if want_z else (u, status, iters)


class BatchedOptimalDecayLinearMPCCBF(BatchedLinearMPCCBF):
    """EXTENSION (BASELINE config 5; no reference counterpart): optimal-decay MPC-CBF for Quad3D with circles AND
    superellipsoid obstacles -- rows ``h(step(x_k,u_k)) - (1 - alpha rho_k) h(x_k) >= 0`` with one decay variable per stage
    (the rel-degree-1 form of the reference's optimal-decay CBF-QP, optimal_decay_cbf_qp.py:96-101,113-125), cost
    ``+ p_sb1 (rho_k - omega1)^2`` and the input term ``R u^2`` (optimal_decay_mpc_cbf.py:178-184); gains, weights and bounds are
    MPCCBF's for the model.  oracle/od_mpc_rd1.py states the problem and the method.
    ``solve(...)`` -> ``u[B,4]``, ``rho[B,N]``, ``status[B]``, ``iters[B]`` (and ``z[B,4N]`` if asked)."""

    def __init__(self, robot_spec, dt=0.05, io_dtype="f64", horizon=None, cbf_param=None, tol=1e-6, max_iter=_lib.IPOPT_MAX_ITER,
                 superellipsoids=True, iter_slices=None, classify_first=True, order=True):
        spec = complete_robot_spec(robot_spec)
        if spec["model"] != "Quad3D":
            raise NotImplementedError("the optimal-decay extension of the linear-model kernel serves Quad3D")
        self.od = {"omega1": 1.0, "p_sb1": 10.0}              # optimal_decay_mpc_cbf.py:88-89
        if cbf_param:
            self.od.update({k: cbf_param[k] for k in ("omega1", "p_sb1") if k in cbf_param})
        self._superellipsoids = bool(superellipsoids)
        super().__init__(robot_spec, dt=dt, io_dtype=io_dtype, horizon=horizon, cbf_param=cbf_param, tol=tol, max_iter=max_iter,
                         iter_slices=iter_slices, classify_first=classify_first, order=order)

    def _params(self, **kw):
        p = make_params(self._mdl, self.cbf_param, self.horizon, self.robot_spec["radius"], self.io_dtype, **kw)
        p.optimal_decay = 1
        p.od_omega_ref, p.od_p_sb = float(self.od["omega1"]), float(self.od["p_sb1"])
        if self._superellipsoids:
            p.circles_only = 0                               # the 7-wide obstacle rows of the other models
        return p

    def solve(self, X, u_prev, goal, obs, want_z=False):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_prev", u_prev), ("goal", goal), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        nx, nu, ng = self._mdl["nx"], self._mdl["nu"], self._mdl["ng"]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if X.shape != (B, nx) or u_prev.shape != (B, nu) or goal.shape != (B, ng) or obs.shape[-1] != 7 \
                or (not shared and obs.shape[0] != B):
            raise ValueError(f"expected X[B,{nx}], u_prev[B,{nu}], goal[B,{ng}], obs[B,K,7] or obs[K,7]")
        if "od" not in self._blob_dev:                       # the cost Hessian of the blob depends on the r-term: rebuild once
            p0 = self._params()
            self._blob_host = build_model_blob(self._lib, p0, self._mdl)
            self._blob_dev = {"od": True}
        u = torch.empty((B, nu), dtype=dt_, device=X.device)
        rho = torch.empty((B, self.horizon), dtype=dt_, device=X.device)
        status = torch.empty((B,), dtype=torch.int32, device=X.device)
        iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        z = torch.empty((B, nu * self.horizon), dtype=dt_, device=X.device) if want_z else None
        p = self._params(obs_shared=shared, tol=self.tol, max_iter=self.max_iter)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        sl = self.slices_for(lambda: self._lib.sc_mpclin_slices_workspace_bytes(C.byref(p), B, K), X.device)
        args = (self._blob(X.device).data_ptr(), B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(),
                obs.data_ptr(), u.data_ptr(), rho.data_ptr(), status.data_ptr(), iters.data_ptr(),
                z.data_ptr() if z is not None else None, stream)
        rc = self._lib.sc_odmpclin_solve_batch(C.byref(p), *args) if sl is None else self._lib.sc_odmpclin_solve_batch_sliced(C.byref(p), C.byref(sl), *args)
        _lib.check(rc, "sc_odmpclin_solve_batch")
        return (u, rho, status, iters, z) if want_z else (u, rho, status, iters)
