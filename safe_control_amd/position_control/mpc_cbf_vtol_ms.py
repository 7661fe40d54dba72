"""MPC-CBF for VTOL2D as do-mpc poses it -- multiple shooting, IPOPT's filter interior point -- on csrc/mpc_vtol_ms.hip (DESIGN.md
kernel 12), with the condensed kernel (csrc/mpc_vtol_wave.hip) behind it for the problems on which IPOPT would enter its
restoration phase.

position_control/mpc_cbf.py:162-174 / :366-369: states and inputs of every stage are variables, the dynamics are equality rows,
every stage starts at x0 and every input at the input applied last, IPOPT runs with its defaults.  ``BatchedVtolMSMPCCBF.solve``
launches that solve for B aircraft; problems that come back SC_STATUS_NEEDS_RESTO (line search below alpha_min: infeasible or
nearly so) are gathered and solved by ``BatchedVtolMPCCBF`` -- the condensed interior point and ITS restoration phase, whose
statuses (infeasible / optimal_inaccurate / optimal) they then carry.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec
from .mpc_cbf import apply_mpc_overrides
from .mpc_cbf_vtol import (CBF_VTOL, HORIZON_VTOL, OD_CBF_VTOL, Q_VTOL, R_VTOL, BatchedOptimalDecayVtolMPCCBF, BatchedVtolMPCCBF, make_od_params,
                           make_params)


class BatchedVtolMSMPCCBF:
    """``solve(X[B,6], u_prev[B,4], goal[B,2], obs[B,K,7] | obs[K,7])`` -> ``u[B,4]``, ``status[B]``, ``iters[B]`` [, ``plan[B, 31*6 + 30*4]``].
    ``ipopt``: overrides of IPOPT's option defaults (``_lib.IPOPT_DEFAULTS``).  ``restoration``: run IPOPT's restoration phase inside the
    kernel (elastic variables on the CBF rows; a workspace of ``sc_mpcvtol_ms_workspace_bytes`` is kept between calls) -- a stationary
    point of the violation comes back as SC_STATUS_INFEASIBLE.  ``fallback``: re-solve what still comes back SC_STATUS_NEEDS_RESTO (only
    with ``restoration=False``) with the condensed kernel, or hand the status to the caller."""

    def __init__(self, robot_spec=None, dt=0.05, io_dtype="f64", cbf_param=None, ipopt=None, fallback=True, max_iter=None, restoration=True):
        self.robot_spec = complete_robot_spec(dict(robot_spec or {"model": "VTOL2D"}))
        if self.robot_spec["model"] != "VTOL2D":
            raise NotImplementedError("this controller serves VTOL2D")
        self.dt = float(dt)
        self.io_name = io_dtype
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.horizon = HORIZON_VTOL
        self.Q, self.R = np.diag(Q_VTOL), np.array(R_VTOL)
        self.cbf_param = cbf_param or apply_mpc_overrides(dict(CBF_VTOL), self.robot_spec)
        self.ipopt = dict(ipopt or {})
        if max_iter is not None:
            self.ipopt["max_iter"] = int(max_iter)
        self.max_iter = int(self.ipopt.get("max_iter", _lib.IPOPT_DEFAULTS["max_iter"]))
        self.fallback = bool(fallback)
        self.restoration = bool(restoration)
        self._resto_ws = None
        self.condensed = BatchedVtolMPCCBF(dict(self.robot_spec), dt=dt, io_dtype=io_dtype, cbf_param=dict(self.cbf_param)) if fallback else None
        self.n_fallback = 0                                  # problems of the last call that went to the condensed kernel
        self.iter_slices = ()
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    @property
    def plan_width(self):
        return (self.horizon + 1) * 6 + self.horizon * 4

    def _ipopt_params(self, B, K, device):
        """sc_ipopt_params of this call; with ``restoration`` the workspace of the in-kernel restoration phase (grown on demand, kept)."""
        import torch
        ip = _lib.default_ipopt(**self.ipopt)
        if self.restoration:
            need = int(self._lib.sc_mpcvtol_ms_workspace_bytes(B, K))
            if self._resto_ws is None or self._resto_ws.numel() < need or self._resto_ws.device != device:
                self._resto_ws = torch.empty((max(need, 8),), dtype=torch.uint8, device=device)
            ip.resto_workspace, ip.resto_workspace_bytes = self._resto_ws.data_ptr(), self._resto_ws.numel()
        return ip

    def solve(self, X, u_prev, goal, obs, want_plan=False, want_trace=False, want_z=False):
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
        want_plan = want_plan or want_z
        plan = torch.empty((B, self.plan_width), dtype=dt_, device=X.device) if want_plan else None
        ip = self._ipopt_params(B, K, X.device)
        trace = torch.zeros((B, ip.max_iter + 1, 8), dtype=torch.float64, device=X.device) if want_trace else None
        p = make_params(self.robot_spec, self.cbf_param, self.horizon, self.dt, self.robot_spec["radius"], self.io_dtype, obs_shared=shared)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        rc = self._lib.sc_mpcvtol_ms_solve_batch(C.byref(p), C.byref(ip), B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(),
                                                 u.data_ptr(), status.data_ptr(), iters.data_ptr(), plan.data_ptr() if plan is not None else None,
                                                 trace.data_ptr() if trace is not None else None, stream)
        _lib.check(rc, "sc_mpcvtol_ms_solve_batch")
        self.n_fallback = 0
        if self.fallback:
            idx = torch.nonzero(status == _lib.STATUS_NEEDS_RESTO).flatten()
            self.n_fallback = int(idx.numel())
            if self.n_fallback:
                ob = obs if shared else obs[idx].contiguous()
                r = self.condensed.solve(X[idx].contiguous(), u_prev[idx].contiguous(), goal[idx].contiguous(), ob, want_z=plan is not None)
                u[idx], status[idx] = r[0], r[1]
                iters[idx] = iters[idx] + r[2]
                if plan is not None:                         # the condensed kernel returns inputs only: states of these plans are not filled in
                    plan[idx] = float("nan")
                    plan[idx, (self.horizon + 1) * 6:] = r[3]
        out = [u, status, iters]
        if want_plan:
            out.append(plan)
        if want_trace:
            out.append(trace)
        return tuple(out)


class BatchedOptimalDecayVtolMSMPCCBF(BatchedVtolMSMPCCBF):
    """Optimal-decay MPC-CBF for VTOL2D (optimal_decay_mpc_cbf.py with a VTOL2D robot) in the multiple-shooting form: the decay rates are two
    more inputs of a stage.  ``solve(...)`` -> ``u[B,4]``, ``rho[B,2N]``, ``status[B]``, ``iters[B]`` [, ``plan``].  Problems that come back
    SC_STATUS_NEEDS_RESTO go to the condensed optimal-decay kernel (``BatchedOptimalDecayVtolMPCCBF``)."""

    def __init__(self, robot_spec=None, dt=0.05, io_dtype="f64", cbf_param=None, ipopt=None, fallback=True, max_iter=None, restoration=True):
        super().__init__(robot_spec, dt=dt, io_dtype=io_dtype, cbf_param=cbf_param or dict(OD_CBF_VTOL), ipopt=ipopt, fallback=False, max_iter=max_iter,
                         restoration=restoration)
        self.fallback = bool(fallback)
        self.condensed = BatchedOptimalDecayVtolMPCCBF(dict(self.robot_spec), dt=dt, io_dtype=io_dtype, cbf_param=dict(self.cbf_param)) if fallback else None

    def solve(self, X, u_prev, goal, obs, want_plan=False, want_trace=False, want_z=False):
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
        want_plan = want_plan or want_z
        plan = torch.empty((B, self.plan_width), dtype=dt_, device=X.device) if want_plan else None
        ip = self._ipopt_params(B, K, X.device)
        trace = torch.zeros((B, ip.max_iter + 1, 8), dtype=torch.float64, device=X.device) if want_trace else None
        p = make_od_params(self.robot_spec, self.cbf_param, self.horizon, self.dt, self.robot_spec["radius"], self.io_dtype, obs_shared=shared)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        rc = self._lib.sc_odmpcvtol_ms_solve_batch(C.byref(p), C.byref(ip), B, K, X.data_ptr(), u_prev.data_ptr(), goal.data_ptr(), obs.data_ptr(),
                                                   u.data_ptr(), rho.data_ptr(), status.data_ptr(), iters.data_ptr(),
                                                   plan.data_ptr() if plan is not None else None, trace.data_ptr() if trace is not None else None, stream)
        _lib.check(rc, "sc_odmpcvtol_ms_solve_batch")
        self.n_fallback = 0
        if self.fallback:
            idx = torch.nonzero(status == _lib.STATUS_NEEDS_RESTO).flatten()
            self.n_fallback = int(idx.numel())
            if self.n_fallback:
                ob = obs if shared else obs[idx].contiguous()
                r = self.condensed.solve(X[idx].contiguous(), u_prev[idx].contiguous(), goal[idx].contiguous(), ob, want_z=plan is not None)
                u[idx], rho[idx], status[idx] = r[0], r[1], r[2]
                iters[idx] = iters[idx] + r[3]
                if plan is not None:
                    plan[idx] = float("nan")
                    plan[idx, (self.horizon + 1) * 6:] = r[4]
        out = [u, rho, status, iters]
        if want_plan:
            out.append(plan)
        if want_trace:
            out.append(trace)
        return tuple(out)
