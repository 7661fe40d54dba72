"""MPC-CBF for DynamicUnicycle2D and DoubleIntegrator2D as do-mpc poses it -- multiple shooting, IPOPT's filter interior point with its
restoration phase -- on csrc/mpc_du_ms.hip (DESIGN.md kernel 13): BASELINE configs[2] in the reference's own formulation.

position_control/mpc_cbf.py:162-174 / :366-369: the states and inputs of every stage are variables, the dynamics are equality rows, every
stage starts at x0 and every input at the input applied last, IPOPT runs with its defaults and whatever it holds at the end is applied
(:384; ``status`` is hard-wired 'optimal', :10).  ``BatchedMSMPCCBF.solve`` launches that solve for B agents, one NLP per wavefront.  The
condensed kernel (``BatchedMPCCBF``, csrc/mpc_cbf.hip) solves the single-shooting form of the same NLP: the same optimum where the NLP has
one, another last iterate where it has no feasible point (tools/exp_ms_vs_condensed.py) -- it stays available as
``robot_spec['mpc_formulation'] = 'condensed'``.  Robots: DynamicUnicycle2D, Unicycle2D, SingleIntegrator2D, DoubleIntegrator2D, KinematicBicycle2D
(the kernel is templated on the model: csrc/mpc_du_ms_solver.hpp); superellipsoid obstacle rows for the first and the fourth (csrc/mpc_du_ms_se.hip,
picked from the rows' flags).  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec
from .mpc_cbf import apply_mpc_overrides, default_mpc_cbf_param, default_mpc_weights, make_params


class BatchedMSMPCCBF:
    """``solve(X[B,4], u_prev[B,2], goal[B,2], obs[B,K,7] | obs[K,7])`` -> ``u[B,2]``, ``status[B] int32``, ``iters[B] int32``
    [, ``plan[B, (N+1)*4 + N*2]``] [, ``trace[B, max_iter+1, 8]``].  ``ipopt``: overrides of IPOPT's option defaults
    (``_lib.IPOPT_DEFAULTS``).  ``superellipsoids``: None = look at the rows' flags (column 6 >= 0.5; one reduction and a host read per call, skipped
    with ``check_circles=False``: rows are then taken as circles) and launch the instantiation that evaluates superellipsoids when there are any;
    True / False = say so.  ``order``: launches of more than 1024 problems start the NLPs whose start point violates a
    CBF row first (a pre-pass kernel and a small workspace; results do not depend on it, the launch ends ~15 % sooner)."""

    def __init__(self, robot_spec=None, dt=0.05, io_dtype="f32", horizon=None, cbf_param=None, ipopt=None, max_iter=None, check_circles=True, order=True,
                 superellipsoids=None):
        self.robot_spec = complete_robot_spec(dict(robot_spec or {"model": "DynamicUnicycle2D"}))
        self.model = self.robot_spec["model"]
        if self.model not in ("DynamicUnicycle2D", "Unicycle2D", "SingleIntegrator2D", "DoubleIntegrator2D", "KinematicBicycle2D"):
            raise NotImplementedError("the multiple-shooting MPC-CBF kernel serves DynamicUnicycle2D, Unicycle2D, SingleIntegrator2D, DoubleIntegrator2D and "
                                      "KinematicBicycle2D (VTOL2D: BatchedVtolMSMPCCBF)")
        self.dt = float(dt)
        self.io_name = io_dtype
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = int(horizon if horizon is not None else self.robot_spec.get("mpc_horizon", 10))
        if not 1 <= self.horizon <= 62:
            raise ValueError("mpc_horizon must be in [1, 62]")
        self.Q, self.R = default_mpc_weights(self.model)
        self.cbf_param = cbf_param or apply_mpc_overrides(default_mpc_cbf_param(self.model), self.robot_spec)
        self.ipopt = dict(ipopt or {})
        if max_iter is not None:
            self.ipopt["max_iter"] = int(max_iter)
        self.max_iter = int(self.ipopt.get("max_iter", _lib.IPOPT_DEFAULTS["max_iter"]))
        self.check_circles = bool(check_circles)
        self.superellipsoids = superellipsoids                # None: look at the rows' flags (if check_circles) and pick the instantiation; True / False: say so
        self.order = bool(order)                              # launches of more than 1024 problems: problems whose start violates a CBF row go first
        self._order_ws = None
        self.iter_slices = ()
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    @property
    def plan_width(self):
        return (self.horizon + 1) * 4 + self.horizon * 2

    def solve(self, X, u_prev, goal, obs, want_plan=False, want_trace=False, out=None):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_prev", u_prev), ("goal", goal), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        if self.model == "SingleIntegrator2D" and X.shape == (B, 2):      # the reference's two-state rows: the kernel reads four columns, the last two unused
            X = torch.cat([X, torch.zeros_like(X)], dim=1)
        if X.shape != (B, 4) or u_prev.shape != (B, 2) or goal.shape != (B, 2) or obs.shape[-1] != 7 or (not shared and obs.shape[0] != B):
            raise ValueError("expected X[B,4], u_prev[B,2], goal[B,2], obs[B,K,7] or obs[K,7]")
        se = 1 if self.superellipsoids is True else 0
        if self.superellipsoids is None and self.check_circles and B > 0 and bool((obs[..., 6] >= 0.5).any().item()):
            se = 1                                            # rows with the superellipsoid flag: the slower instantiation that evaluates them
        if se and self.model not in ("DynamicUnicycle2D", "DoubleIntegrator2D"):
            raise NotImplementedError(f"superellipsoid obstacle rows: kernel 13 serves them for DynamicUnicycle2D and DoubleIntegrator2D, not for {self.model}")
        if out is None:
            u = torch.empty((B, 2), dtype=dt_, device=X.device)
            status = torch.empty((B,), dtype=torch.int32, device=X.device)
            iters = torch.empty((B,), dtype=torch.int32, device=X.device)
        else:
            u, status, iters = out[:3]
        plan = torch.empty((B, self.plan_width), dtype=dt_, device=X.device) if want_plan else None
        ip = _lib.default_ipopt(**self.ipopt)
        if self.order and B > 1024:                               # launch-order workspace (kept between calls): the long solves start first
            need = int(self._lib.sc_mpccbf_ms_workspace_bytes(B))
            if self._order_ws is None or self._order_ws.numel() < need or self._order_ws.device != X.device:
                self._order_ws = torch.empty((need,), dtype=torch.uint8, device=X.device)
            ip.resto_workspace, ip.resto_workspace_bytes = self._order_ws.data_ptr(), self._order_ws.numel()
        trace = torch.zeros((B, ip.max_iter + 1, 8), dtype=torch.float64, device=X.device) if want_trace else None
        p = make_params(self.robot_spec, self.cbf_param, self.Q, self.R, self.horizon, self.dt, self.robot_spec["radius"], self.io_dtype,
                        obs_shared=shared)
        p.superellipsoid_rows = se
        stream = torch.cuda.current_stream(X.device).cuda_stream
        rc = self._lib.sc_mpccbf_ms_solve_batch(C.byref(p), C.byref(ip), B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(),
                                                u.data_ptr(), status.data_ptr(), iters.data_ptr(), plan.data_ptr() if plan is not None else None,
                                                trace.data_ptr() if trace is not None else None, stream)
        _lib.check(rc, "sc_mpccbf_ms_solve_batch")
        res = [u, status, iters]
        if want_plan:
            res.append(plan)
        if want_trace:
            res.append(trace)
        return tuple(res)
