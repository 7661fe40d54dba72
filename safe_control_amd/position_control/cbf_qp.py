"""CBF-QP position controllers backed by the gfx950 HIP kernels.

``CBFQP`` keeps the plugin surface of the reference class of the same name
(position_control/cbf_qp.py:4-199: ``__init__(robot, robot_spec, num_obs)``,
``setup_control_problem()``, ``solve_control_problem(robot_state, control_ref,
obs_list)``, ``.status``, ``.cbf_param``) so ``LocalTrackingController``
(tracking.py:140-142, 611-616, 627-634) can use it unchanged.

``BatchedCBFQP`` is the same controller for B agents at once on HBM-resident
tensors: one kernel launch assembles every agent's CBF rows and solves every
QP (csrc/cbf_qp_kernel.hpp).

No CPU fallback: the HIP library must be present (``_lib.load`` raises).
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..robots.spec import complete_robot_spec

REL_DEG2_MODELS = ("DynamicUnicycle2D", "KinematicBicycle2D", "DoubleIntegrator2D", "Quad2D")


def default_cbf_param(model):
    """position_control/cbf_qp.py:12-35 (models the batched engine supports)."""
    if model in REL_DEG2_MODELS:
        return {"alpha1": 1.5, "alpha2": 1.5}
    if model in ("SingleIntegrator2D", "Unicycle2D"):
        return {"alpha": 1.0}
    return {"alpha": 1.5}


def apply_cbf_overrides(cbf_param, robot_spec):
    """position_control/cbf_qp.py:37-43."""
    for key, src in (("alpha", "cbf_alpha"), ("alpha1", "cbf_alpha1"), ("alpha2", "cbf_alpha2")):
        if src in robot_spec:
            cbf_param[key] = float(robot_spec[src])
    return cbf_param


def input_bounds(robot_spec):
    """Input box of setup_control_problem: DU cbf_qp.py:62-65, KB family :70-73."""
    if robot_spec["model"] == "DynamicUnicycle2D":
        hi = (float(robot_spec["a_max"]), float(robot_spec["w_max"]))
    elif robot_spec["model"] == "SingleIntegrator2D":          # cbf_qp.py:54-57
        hi = (float(robot_spec["v_max"]), float(robot_spec["v_max"]))
    elif robot_spec["model"] == "Unicycle2D":                  # cbf_qp.py:58-61
        hi = (float(robot_spec["v_max"]), float(robot_spec["w_max"]))
    elif robot_spec["model"] == "DoubleIntegrator2D":          # cbf_qp.py:66-69
        hi = (float(robot_spec["a_max"]), float(robot_spec["a_max"]))
    elif robot_spec["model"] == "Quad2D":                      # cbf_qp.py:74-79: f_min <= u <= f_max
        return ((float(robot_spec["f_min"]),) * 2, (float(robot_spec["f_max"]),) * 2)
    else:
        hi = (float(robot_spec["a_max"]), float(robot_spec["beta_max"]))
    return (-hi[0], -hi[1]), hi


def _check_cbfqp_model(robot_spec):
    """Quad3D / VTOL2D have no CBF-QP in the reference either (quad3D.py:269-273 raises); Manipulator2D has its own class."""
    if robot_spec["model"] not in _lib.MODEL_IDS:
        raise ValueError(f"CBF-QP on the batched engine does not support model {robot_spec['model']!r} "
                         f"(supported: {sorted(_lib.MODEL_IDS)} and Manipulator2D through ManipulatorCBFQP)")


def make_params(robot_spec, cbf_param, dt, radius, io_dtype, compute_dtype, obs_shared=False):
    model = robot_spec["model"]
    p = _lib.CbfQpParams()
    p.model_id = _lib.MODEL_IDS[model]
    p.io_dtype = io_dtype
    p.compute_dtype = compute_dtype
    mode = robot_spec.get("cbf_mode", "cbf")            # cbf_qp.py:120
    if mode not in _lib.CBF_MODE:
        raise ValueError(f"cbf_mode must be 'cbf' or 'hard', got {mode!r}")
    p.cbf_mode = _lib.CBF_MODE[mode]
    p.obs_shared = 1 if obs_shared else 0
    p.robot_radius = float(radius)
    p.dt = float(dt)
    if model in REL_DEG2_MODELS:
        p.alpha1, p.alpha2 = float(cbf_param["alpha1"]), float(cbf_param["alpha2"])
    else:
        p.alpha1, p.alpha2 = float(cbf_param["alpha"]), 0.0
    lo, hi = input_bounds(robot_spec)
    p.u_min[0], p.u_min[1] = lo
    p.u_max[0], p.u_max[1] = hi
    p.rear_ax_dist = float(robot_spec.get("rear_ax_dist", 0.0))
    p.state_dim = _lib.STATE_DIM.get(model, 4)
    p.mass = float(robot_spec.get("mass", 1.0))
    return p


def _pad_obstacle(ob):
    """Obstacle row -> 7 values.  Shorter rows get zero velocity / flag like the
    reference's callers do (tracking.py:282-290, examples/test_tracking.py:147-148)."""
    ob = np.asarray(ob, dtype=np.float64).reshape(-1)
    if ob.shape[0] == 7:
        return ob
    if ob.shape[0] > 7:
        return ob[:7]
    if ob.shape[0] < 3:
        raise ValueError(f"Invalid obstacle format: {ob}")
    return np.concatenate([ob, np.zeros(7 - ob.shape[0])])


class CBFQP:
    """Drop-in for position_control.cbf_qp.CBFQP (single agent per call)."""

    def __new__(cls, robot, robot_spec, *args, **kwargs):
        # the reference's CBFQP serves the 3-joint arm from the same class (cbf_qp.py:94-104, :130-151); here it is its
        # own kernel and host class
        if cls is CBFQP and robot_spec.get("model") == "Manipulator2D":
            from .manipulator_cbf_qp import ManipulatorCBFQP
            return ManipulatorCBFQP(robot, robot_spec, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, robot, robot_spec, num_obs=1, device=0):
        self.robot = robot
        self.robot_spec = complete_robot_spec(robot_spec)
        _check_cbfqp_model(self.robot_spec)
        self.num_obs = int(num_obs)
        self.device = device
        self.cbf_param = apply_cbf_overrides(default_cbf_param(self.robot_spec["model"]), self.robot_spec)
        self.status = "optimal"
        self.setup_control_problem()

    def setup_control_problem(self):
        """The reference builds a cvxpy Problem here (cbf_qp.py:47-106); the HIP
        kernel needs no graph, only the library handle and buffers."""
        if not 1 <= self.num_obs <= _lib.CBFQP_MAX_OBS:
            raise ValueError(f"num_obs must be in [1, {_lib.CBFQP_MAX_OBS}]")
        self._lib = _lib.load()
        self._u = np.zeros(2, dtype=np.float64)
        self._status = np.zeros(1, dtype=np.int32)
        self._h = np.zeros(self.num_obs, dtype=np.float64)
        self.h = None

    def solve_control_problem(self, robot_state, control_ref, obs_list):
        u_ref = np.asarray(control_ref["u_ref"], dtype=np.float64).reshape(-1)
        if obs_list is None:                            # cbf_qp.py:113-118: u_ref, unclipped
            self.status = "optimal"
            return u_ref.reshape(-1, 1).copy()
        rows = [_pad_obstacle(o) for o in obs_list if o is not None][: self.num_obs]   # cbf_qp.py:122-128
        k = len(rows)
        K = max(k, 1)
        obs = np.zeros((K, 7), dtype=np.float64)
        if k:
            obs[:k] = np.asarray(rows)
        n_obs = np.array([k], dtype=np.int32)
        nx = _lib.STATE_DIM.get(self.robot_spec["model"], 4)
        xs = np.asarray(robot_state, dtype=np.float64).reshape(-1)[:nx]
        X = np.zeros(nx, dtype=np.float64)                # SingleIntegrator2D has 2 states: padded to the [B,4] layout
        X[: xs.shape[0]] = xs
        p = make_params(self.robot_spec, self.cbf_param, self.robot.dt, self.robot.robot_radius,
                        _lib.DTYPE_F64, _lib.DTYPE_F64)
        h = np.zeros(K, dtype=np.float64)
        rc = self._lib.sc_cbfqp_solve_batch_host(
            C.byref(p), 1, K, X.ctypes.data, u_ref.ctypes.data, obs.ctypes.data, n_obs.ctypes.data,
            self._u.ctypes.data, self._status.ctypes.data, h.ctypes.data, int(self.device))
        _lib.check(rc, "sc_cbfqp_solve_batch_host")
        st = int(self._status[0])
        if st == _lib.STATUS_BAD_OBSTACLE:
            raise ValueError("obstacle flag (last column) must be 0 (circle) or 1 (superellipsoid) "
                             f"for {self.robot_spec['model']}")
        self.status = _lib.STATUS_STRINGS[st]
        self.h = h[:k]
        if st != _lib.STATUS_OPTIMAL:
            return None                                  # cvxpy leaves u.value None when infeasible
        return self._u.reshape(-1, 1).copy()


class BatchedCBFQP:
    """CBF-QP for B agents per launch on device tensors.

    ``solve(X[B,4], u_ref[B,2], obs[B,K,7] | obs[K,7], n_obs[B]|None)`` ->
    ``u[B,2]`` (NaN where not optimal), ``status[B] int32``, ``h[B,K]``.
    Tensors must be contiguous CUDA tensors of ``io_dtype``; the launch goes on
    the current torch stream and does not synchronise.
    """

    def __init__(self, robot_spec, dt=0.05, io_dtype="f32", compute_dtype="f64", cbf_param=None):
        self.robot_spec = complete_robot_spec(robot_spec)
        _check_cbfqp_model(self.robot_spec)
        self.dt = float(dt)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.compute_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[compute_dtype]
        self.cbf_param = cbf_param or apply_cbf_overrides(default_cbf_param(self.robot_spec["model"]), self.robot_spec)
        self._lib = _lib.load()

    @property
    def torch_dtype(self):
        import torch
        return torch.float32 if self.io_dtype == _lib.DTYPE_F32 else torch.float64

    def solve(self, X, u_ref, obs, n_obs=None, want_h=True, out=None):
        import torch
        dt_ = self.torch_dtype
        for name, t in (("X", X), ("u_ref", u_ref), ("obs", obs)):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt_):
                raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dt_}")
        B = X.shape[0]
        shared = obs.dim() == 2
        K = obs.shape[-2]
        nx = _lib.STATE_DIM.get(self.robot_spec["model"], 4)
        if X.shape != (B, nx) or u_ref.shape != (B, 2) or obs.shape[-1] != 7 or (not shared and obs.shape[0] != B):
            raise ValueError(f"expected X[B,{nx}], u_ref[B,2], obs[B,K,7] or obs[K,7]")
        if n_obs is not None and not (n_obs.is_cuda and n_obs.dtype == torch.int32 and n_obs.shape == (B,)
                                      and n_obs.is_contiguous()):
            raise ValueError("n_obs must be a contiguous CUDA int32 tensor of shape [B]")
        if out is None:
            u = torch.empty((B, 2), dtype=dt_, device=X.device)
            status = torch.empty((B,), dtype=torch.int32, device=X.device)
            h = torch.empty((B, K), dtype=dt_, device=X.device) if want_h else None
        else:
            u, status, h = out
        p = make_params(self.robot_spec, self.cbf_param, self.dt, self.robot_spec["radius"],
                        self.io_dtype, self.compute_dtype, obs_shared=shared)
        stream = torch.cuda.current_stream(X.device).cuda_stream
        rc = self._lib.sc_cbfqp_solve_batch(
            C.byref(p), B, K, X.data_ptr(), u_ref.data_ptr(), obs.data_ptr(),
            n_obs.data_ptr() if n_obs is not None else None,
            u.data_ptr(), status.data_ptr(), h.data_ptr() if h is not None else None, stream)
        _lib.check(rc, "sc_cbfqp_solve_batch")
        return u, status, h
