"""Optimal-decay CBF-QP backed by the gfx950 HIP kernel (csrc/od_cbf_qp.hip).

``OptimalDecayCBFQP`` keeps the surface of the reference class
(position_control/optimal_decay_cbf_qp.py:13-158: ``__init__(robot, robot_spec)``,
``setup_control_problem()``, ``solve_control_problem(robot_state, control_ref, nearest_obs)``,
``.status``, ``.cbf_param`` with alpha/omega/p_sb keys).  The reference copy is stale -- control_step
hands it the (k,7) array of nearest obstacles while it expects one obstacle;
here a (k,7) array means "use the nearest = first row", a (7,) / (7,1) array is taken as is.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec
from .cbf_qp import REL_DEG2_MODELS, _pad_obstacle, make_params


class NotCompatibleError(Exception):
    """optimal_decay_cbf_qp.py:4-11."""

    def __init__(self, message="Currently not compatible with the robot model."):
        self.message = message
        super().__init__(self.message)


def default_od_param(model):
    """optimal_decay_cbf_qp.py:17-50."""
    if model in ("DynamicUnicycle2D", "KinematicBicycle2D", "Quad2D"):        # :17-32, :38-45
        return dict(alpha1=0.5, alpha2=0.5, omega1=1.0, p_sb1=10 ** 4, omega2=1.0, p_sb2=10 ** 4)
    if model in ("KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"):
        return dict(alpha=0.5, omega1=1.0, p_sb1=10 ** 4)
    raise NotCompatibleError("Infeasible or Collision")


def make_od_params(robot_spec, cbf_param, dt, radius, io_dtype, compute_dtype):
    p = _lib.OdCbfQpParams()
    p.qp = make_params(robot_spec, cbf_param, dt, radius, io_dtype, compute_dtype)
    p.omega_ref[0] = float(cbf_param.get("omega1", 1.0))
    p.omega_ref[1] = float(cbf_param.get("omega2", 1.0))
    p.p_sb[0] = float(cbf_param.get("p_sb1", 1e4))
    p.p_sb[1] = float(cbf_param.get("p_sb2", 1e4))
    return p


class OptimalDecayCBFQP:
    def __init__(self, robot, robot_spec, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        self.device = device
        self.cbf_param = default_od_param(self.robot_spec["model"])
        self.status = "optimal"
        self.omega = np.array([1.0, 1.0])
        self.setup_control_problem()

    def setup_control_problem(self):
        self._lib = _lib.load()

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        u_ref = np.ascontiguousarray(np.asarray(control_ref["u_ref"], dtype=np.float64).reshape(-1))
        nx = 6 if self.robot_spec["model"] == "Quad2D" else 4
        X = np.ascontiguousarray(np.asarray(robot_state, dtype=np.float64).reshape(-1)[:nx])
        has = np.array([0 if nearest_obs is None else 1], dtype=np.int32)
        obs = np.zeros(7)
        if nearest_obs is not None:
            ob = np.asarray(nearest_obs, dtype=np.float64)
            ob = ob[0] if (ob.ndim == 2 and ob.shape[1] >= 3 and ob.shape[0] != 7) or (ob.ndim == 2 and ob.shape == (7, 7)) else ob.reshape(-1)
            obs = _pad_obstacle(ob)
        obs = np.ascontiguousarray(obs)
        p = make_od_params(self.robot_spec, self.cbf_param, self.robot.dt, self.robot.robot_radius,
                           _lib.DTYPE_F64, _lib.DTYPE_F64)
        u = np.zeros(2); w = np.zeros(2); st = np.zeros(1, dtype=np.int32); h = np.zeros(1)
        rc = self._lib.sc_odcbfqp_solve_batch_host(C.byref(p), 1, X.ctypes.data, u_ref.ctypes.data, obs.ctypes.data,
                                                   has.ctypes.data, u.ctypes.data, w.ctypes.data, st.ctypes.data,
                                                   h.ctypes.data, int(self.device))
        _lib.check(rc, "sc_odcbfqp_solve_batch_host")
        if int(st[0]) == _lib.STATUS_BAD_OBSTACLE:
            raise ValueError("obstacle flag (last column) must be 0 (circle) or 1 (superellipsoid)")
        self.status = _lib.STATUS_STRINGS[int(st[0])]
        self.omega = w.copy()
        self.h = float(h[0])
        if int(st[0]) != _lib.STATUS_OPTIMAL:
            return None
        return u.reshape(-1, 1).copy()


class BatchedOptimalDecayCBFQP:
    """``solve(X[B,4], u_ref[B,2], obs[B,7], has_obs[B]|None)`` -> ``u[B,2], omega[B,2], status[B], h[B]``."""

    def __init__(self, robot_spec, dt=0.05, io_dtype="f32", compute_dtype="f64", cbf_param=None):
        self.robot_spec = complete_robot_spec(robot_spec)
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.compute_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[compute_dtype]
        self.cbf_param = cbf_param or default_od_param(self.robot_spec["model"])
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def solve(self, X, u_ref, obs, has_obs=None):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_ref", u_ref), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        nx = 6 if self.robot_spec["model"] == "Quad2D" else 4
        if X.shape != (B, nx) or u_ref.shape != (B, 2) or obs.shape != (B, 7):
            raise ValueError(f"expected X[B,{nx}], u_ref[B,2], obs[B,7]")
        u = torch.empty((B, 2), dtype=dt_, device=X.device)
        w = torch.empty((B, 2), dtype=dt_, device=X.device)
        st = torch.empty((B,), dtype=torch.int32, device=X.device)
        h = torch.empty((B,), dtype=dt_, device=X.device)
        p = make_od_params(self.robot_spec, self.cbf_param, self.dt, self.robot_spec["radius"], self.io_dtype,
                           self.compute_dtype)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        rc = self._lib.sc_odcbfqp_solve_batch(C.byref(p), B, X.data_ptr(), u_ref.data_ptr(), obs.data_ptr(),
                                              has_obs.data_ptr() if has_obs is not None else None,
                                              u.data_ptr(), w.data_ptr(), st.data_ptr(), h.data_ptr(), stream)
        _lib.check(rc, "sc_odcbfqp_solve_batch")
        return u, w, st, h
