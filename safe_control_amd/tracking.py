"""Batched closed-loop tracking controller backed by the fused rollout kernel (csrc/tracking.hip).

``BatchedTrackingController`` is the many-agent counterpart of the reference's
``LocalTrackingController`` (tracking.py:36-756) for the part of ``control_step`` that surrounds the
solve: goal / state machine, nearest-unpassed obstacle selection, nominal input, CBF-QP, collision
checks, robot step and return code all run on the GPU, ``n`` control steps per launch, with every
agent's state resident in registers between steps.  Rendering, sensing footprints, unknown-obstacle
detection and attitude controllers are out of scope.

Host side (this file) only prepares waypoints the way ``set_waypoints`` / ``filter_waypoints`` do
(tracking.py:197-249) and owns the device tensors.  No CPU fallback.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .position_control.cbf_qp import apply_cbf_overrides, default_cbf_param, make_params
from .robots.spec import complete_robot_spec


def _wrap(a):
    return ((a + math.pi) % (2.0 * math.pi)) - math.pi


class BatchedTrackingController:
    """B agents, one shared obstacle table ``obs [M,7]`` (known obstacles, tracking.py:113), one
    waypoint list per agent (or one shared list).

    ``control_step(n=1)`` advances every running agent ``n`` steps and returns the per-agent return
    codes (0 running, -1 all waypoints reached, -2 infeasible QP or collision; sticky), like
    ``LocalTrackingController.control_step`` does for one robot.
    """

    def __new__(cls, X0, robot_spec, *args, **kwargs):
        if cls is BatchedTrackingController and robot_spec.get("model") in ("Quad2D", "Quad3D", "VTOL2D"):
            return super().__new__(BatchedQuadTrackingController)      # 6 / 12 states, 2 / 4 inputs: its own select / apply kernels
        return super().__new__(cls)

    def __init__(self, X0, robot_spec, controller_type=None, dt=0.05, enable_rotation=True, obs=None,
                 dyn_obs=False, io_dtype="f64", device="cuda:0"):
        import torch
        self.torch = torch
        controller_type = controller_type or {"pos": "cbf_qp"}
        self.pos_controller_type = controller_type.get("pos", "cbf_qp")           # tracking.py:140-154
        if self.pos_controller_type not in ("cbf_qp", "mpc_cbf", "optimal_decay_mpc_cbf"):
            raise ValueError("position controllers of the batched loop: 'cbf_qp' (fused rollout), 'mpc_cbf', "
                             "'optimal_decay_mpc_cbf' (select / solve / apply per step)")
        self.robot_spec = complete_robot_spec(robot_spec)
        self.robot_spec.setdefault("exploration", False)
        self.model = self.robot_spec["model"]
        if self.model not in _lib.MODEL_IDS or self.model == "Quad2D":
            raise ValueError(f"the batched closed loop does not support model {self.model!r}")
        # the integrators keep their heading outside the state (robots/robot.py:66-72); their rotate state runs the
        # attitude controllers, which are out of scope: enable_rotation must be off and the heading then never changes
        self.integrator = self.model in ("SingleIntegrator2D", "DoubleIntegrator2D")
        if self.integrator and enable_rotation:
            raise ValueError("SingleIntegrator2D / DoubleIntegrator2D run with enable_rotation=False")
        if self.integrator and self.pos_controller_type not in ("cbf_qp", "mpc_cbf"):
            raise ValueError("integrators: 'cbf_qp' or 'mpc_cbf' (SingleIntegrator2D: csrc/mpc_lin.hip, DoubleIntegrator2D: csrc/mpc_gn.hip)")
        self.dt = float(dt)
        self.enable_rotation = bool(enable_rotation)
        self.dyn_obs = bool(dyn_obs)
        self.device = torch.device(device)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.tdtype = torch.float32 if io_dtype == "f32" else torch.float64
        self.num_constraints = int(self.robot_spec.get("num_constraints", 10))          # tracking.py:134-138
        self.reached_threshold = float(self.robot_spec.get("reached_threshold", 0.3))    # tracking.py:49-54
        self.rotation_threshold = 0.1                                                    # tracking.py:46
        self.fov_angle = math.radians(float(self.robot_spec.get("fov_angle", 70.0)))      # robots/robot.py:53-54
        self.cbf_param = apply_cbf_overrides(default_cbf_param(self.model), self.robot_spec)
        self._lib = _lib.load()

        X0 = np.asarray(X0, dtype=np.float64)
        if X0.ndim == 1:
            X0 = X0[None, :]
        self.yaw = None
        if self.integrator:                                    # robots/robot.py:66-72: X0 = [x, y, (vx, vy,) yaw]
            nst = 2 if self.model == "SingleIntegrator2D" else 4
            self.yaw = X0[:, nst].copy() if X0.shape[1] > nst else np.zeros(X0.shape[0])
            X0 = np.hstack([X0[:, :nst], np.zeros((X0.shape[0], 4 - nst))])
        elif X0.shape[1] == 3:                                 # tracking.py:66-68: initial speed 0
            X0 = np.hstack([X0, np.zeros((X0.shape[0], 1))])
        self.B = X0.shape[0]
        t = lambda a, dt_=None: torch.tensor(a, dtype=dt_ or self.tdtype, device=self.device)
        self.X = t(X0)
        self.state_machine = torch.zeros(self.B, dtype=torch.int32, device=self.device)      # 'idle'
        self.current_goal_index = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.goal = torch.zeros((self.B, 3), dtype=self.tdtype, device=self.device)          # gx, gy, valid
        self.ret = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.ret_step = torch.full((self.B,), -1, dtype=torch.int32, device=self.device)
        self.u_pos = torch.zeros((self.B, 2), dtype=self.tdtype, device=self.device)
        self.set_obstacles(obs)
        self.waypoints = None
        self.n_wp = None
        self.steps_done = 0
        self.mpc = None
        if self.pos_controller_type != "cbf_qp":
            if self.dyn_obs:
                raise ValueError("moving obstacle tables are stepped by the fused 'cbf_qp' rollout only")
            from .position_control.mpc_cbf import BatchedMPCCBF
            from .position_control.optimal_decay_mpc_cbf import BatchedOptimalDecayMPCCBF
            if self.model == "SingleIntegrator2D":             # linear model: csrc/mpc_lin.hip
                from .position_control.mpc_cbf_linear import BatchedLinearMPCCBF
                cls = BatchedLinearMPCCBF
            elif self.model in ("DoubleIntegrator2D", "KinematicBicycle2D", "KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"):
                # barrier through the robot's own step(): csrc/mpc_gn.hip
                if self.pos_controller_type != "mpc_cbf":
                    raise ValueError(f"{self.model}: 'cbf_qp' or 'mpc_cbf'")
                from .position_control.mpc_cbf_gn import BatchedGnMPCCBF
                cls = BatchedGnMPCCBF
            else:
                cls = BatchedMPCCBF if self.pos_controller_type == "mpc_cbf" else BatchedOptimalDecayMPCCBF
            self.mpc = cls(self.robot_spec, dt=self.dt, io_dtype=io_dtype)
            # DynamicUnicycle2D, Unicycle2D and DoubleIntegrator2D under 'mpc_cbf': the NLP as do-mpc poses it (multiple shooting, IPOPT's algorithm: csrc/mpc_du_ms.hip, kernel 13)
            # unless robot_spec['mpc_formulation'] = 'condensed'; a scene with superellipsoid rows runs on the condensed kernel
            self.mpc_ms = None
            ms_model = (cls is BatchedMPCCBF and self.model in ("DynamicUnicycle2D", "Unicycle2D")) or (self.model == "DoubleIntegrator2D" and self.pos_controller_type == "mpc_cbf")
            want = self.robot_spec.get("mpc_formulation", "multiple_shooting" if ms_model else "condensed")
            if self.model == "KinematicBicycle2D" and self.pos_controller_type == "mpc_cbf":      # on request only (position_control/mpc_cbf_gn.py: GnMPCCBF)
                ms_model = True
            if ms_model and want == "multiple_shooting" and self.num_constraints <= 16:
                from .position_control.mpc_cbf_ms import BatchedMSMPCCBF
                self.mpc_ms = BatchedMSMPCCBF(self.robot_spec, dt=self.dt, io_dtype=io_dtype, check_circles=False)
            self.u_prev = torch.zeros((self.B, 2), dtype=self.tdtype, device=self.device)   # do-mpc's u0 per agent
            self.mpc_status = torch.zeros(self.B, dtype=torch.int32, device=self.device)

    # -- obstacles -----------------------------------------------------------------------------
    def set_obstacles(self, obs):
        torch = self.torch
        if obs is None or len(obs) == 0:
            self.obs = torch.zeros((0, 7), dtype=self.tdtype, device=self.device)
            self.obs_has_superellipsoid = False
            return
        obs = np.asarray(obs, dtype=np.float64)
        if obs.shape[1] < 7:                                   # examples/test_tracking.py:147-148
            obs = np.hstack([obs, np.zeros((obs.shape[0], 7 - obs.shape[1]))])
        self.obs = torch.tensor(obs[:, :7], dtype=self.tdtype, device=self.device).contiguous()
        self.obs_has_superellipsoid = bool((obs[:, 6] >= 0.5).any())

    # -- waypoints: set_waypoints / filter_waypoints / first update_goal (tracking.py:197-249, 497-535) --
    def set_waypoints(self, waypoints):
        torch = self.torch
        X = self.X.double().cpu().numpy()
        # one list of [x, y(, theta)] rows shared by every agent, or one such list per agent
        shared = (isinstance(waypoints, np.ndarray) and waypoints.ndim == 2) or \
            (isinstance(waypoints, (list, tuple)) and len(waypoints) > 0 and np.ndim(waypoints[0]) == 1)
        lists = [np.asarray(waypoints, dtype=np.float64)] * self.B if shared else \
            [np.asarray(w, dtype=np.float64) for w in waypoints]
        filt = []
        for i in range(self.B):
            wp = lists[i]
            if len(wp) >= 2:                                   # filter_waypoints
                aug = np.vstack((X[i, :2], wp[:, :2]))
                dist = np.linalg.norm(np.diff(aug, axis=0), axis=1)
                mask = np.concatenate(([False], dist >= self.reached_threshold))
                wp = aug[mask]
            filt.append(np.asarray(wp, dtype=np.float64)[:, :2].reshape(-1, 2))
        W = max(1, max(len(w) for w in filt))
        wps = np.zeros((self.B, W, 2))
        n_wp = np.zeros(self.B, dtype=np.int32)
        idx = np.zeros(self.B, dtype=np.int32)
        sm = np.zeros(self.B, dtype=np.int32)
        goal = np.zeros((self.B, 3))
        for i in range(self.B):
            w = filt[i]
            n_wp[i] = len(w)
            wps[i, : len(w)] = w
            # update_goal with state machine 'idle' (tracking.py:517-535)
            g = None
            if len(w) > 0:
                if np.linalg.norm(X[i, :2] - w[0]) < self.reached_threshold:
                    idx[i] = 1
                if idx[i] < len(w):
                    g = w[idx[i]]
            if g is not None:                                   # tracking.py:214-226
                ang = math.atan2(g[1] - X[i, 1], g[0] - X[i, 0])
                yaw = self.yaw[i] if self.integrator else X[i, 2]
                in_fov = abs(_wrap(ang - yaw)) <= self.fov_angle / 2
                if not in_fov:
                    if self.robot_spec["exploration"]:
                        sm[i] = _lib.SM_ROTATE
                        goal[i] = [g[0], g[1], 1.0]
                    else:
                        sm[i] = _lib.SM_STOP
                else:
                    sm[i] = _lib.SM_TRACK
                    goal[i] = [g[0], g[1], 1.0]
        self.waypoints = torch.tensor(wps, dtype=self.tdtype, device=self.device).contiguous()
        self.n_wp = torch.tensor(n_wp, dtype=torch.int32, device=self.device)
        self.current_goal_index = torch.tensor(idx, dtype=torch.int32, device=self.device)
        self.state_machine = torch.tensor(sm, dtype=torch.int32, device=self.device)
        self.goal = torch.tensor(goal, dtype=self.tdtype, device=self.device).contiguous()
        self.ret.zero_()
        self.ret_step.fill_(-1)

    # -- params --------------------------------------------------------------------------------
    def _params(self, n_steps):
        rs = self.robot_spec
        p = _lib.TrackingParams()
        p.qp = make_params(rs, self.cbf_param, self.dt, rs["radius"], self.io_dtype, _lib.DTYPE_F64)
        p.n_steps = int(n_steps)
        p.step_offset = int(self.steps_done)                     # ret_step is an absolute control-step index
        p.max_waypoints = int(self.waypoints.shape[1])
        p.waypoints_shared = 0
        p.enable_rotation = 1 if self.enable_rotation else 0
        p.dyn_obs = 1 if self.dyn_obs else 0
        p.num_constraints = self.num_constraints
        p.reached_threshold = self.reached_threshold
        p.rotation_threshold = self.rotation_threshold
        p.v_max = float(rs["v_max"])
        p.v_min = float(rs.get("v_min", 0.0))
        if self.model == "DoubleIntegrator2D":                  # double_integrator2D.py:117-118, :151
            p.k_omega = 2.0
            p.k_a = float(rs.get("nominal_k_a", 1.0))
            p.k_v = float(rs.get("nominal_k_v", 1.0))
        elif self.model == "DynamicUnicycle2D":                 # robots/dynamic_unicycle2D.py:84-86
            p.k_omega = float(rs.get("nominal_k_omega", 2.0))
            p.k_a = float(rs.get("nominal_k_a", 1.0))
            p.k_v = float(rs.get("nominal_k_v", 1.0))
        else:                                                   # forwarded positionally by BaseRobot (robot.py:401-408)
            p.k_omega, p.k_a, p.k_v = 2.0, 1.0, 1.0
        p.delta_max = float(rs.get("delta_max", 0.0))
        p.wheel_base = float(rs.get("wheel_base", 0.0))
        return p

    # -- stepping -------------------------------------------------------------------------------
    def control_step(self, n=1, record=False):
        """Advance ``n`` control steps in one launch.  Returns ``ret`` [B] (and ``(traj_X [n,B,4],
        traj_U [n,B,2])`` when ``record``)."""
        torch = self.torch
        if self.waypoints is None:
            raise RuntimeError("call set_waypoints first")
        if self.mpc is not None:
            return self._control_step_split(n, record)
        p = self._params(n)
        tX = torch.empty((n, self.B, 4), dtype=self.tdtype, device=self.device) if record else None
        tU = torch.empty((n, self.B, 2), dtype=self.tdtype, device=self.device) if record else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.sc_tracking_rollout_batch(
            C.byref(p), self.B, int(self.obs.shape[0]), self.X.data_ptr(), self.waypoints.data_ptr(),
            self.n_wp.data_ptr(), self.current_goal_index.data_ptr(), self.state_machine.data_ptr(),
            self.goal.data_ptr(), self.obs.data_ptr() if self.obs.shape[0] else None, self.u_pos.data_ptr(),
            self.ret.data_ptr(), self.ret_step.data_ptr(),
            tX.data_ptr() if record else None, tU.data_ptr() if record else None, stream)
        _lib.check(rc, "sc_tracking_rollout_batch")
        self.steps_done += n
        return (self.ret, tX, tU) if record else self.ret

    def _dummy_rows(self, like):
        """[1000, 1000, 0, 0, 0, 0, 0] rows (update_tvp's padding, mpc_cbf.py:360) in the shape of one obstacle selection."""
        d = getattr(self, "_dummy_obs", None)
        if d is None or d.shape != like.shape[1:] or d.dtype != like.dtype:
            d = self.torch.zeros(like.shape[1:], dtype=like.dtype, device=like.device)
            d[:, 0] = 1000.0; d[:, 1] = 1000.0
            self._dummy_obs = d
        return d.unsqueeze(0)

    def _easy_problem(self, X, goal):
        """State and goal rows of a problem every MPC kernel solves in a few iterations (the slots of agents outside 'track')."""
        e = getattr(self, "_easy", None)
        if e is None or e[0].shape[1] != X.shape[1] or e[0].dtype != X.dtype:
            xe = self.torch.zeros((1, X.shape[1]), dtype=X.dtype, device=X.device)
            ge = self.torch.zeros((1, goal.shape[1]), dtype=goal.dtype, device=goal.device)
            ge[0, 0] = 1.0
            if X.shape[1] == 4:
                xe[0, 3] = 0.5                                     # unicycles / bicycle: 0.5 m/s along x; DoubleIntegrator2D: 0.5 m/s along y
            self._easy = e = (xe, ge)
        return e

    def _control_step_split(self, n, record):
        """control_step with an MPC position controller: per step  select (tracking.py:569-609) -> one MPC launch for
        the whole batch -> apply (tracking.py:627-668).  Agents whose state machine is not 'track' get u_ref
        (mpc_cbf.py:379-381) and keep their u_prev; MPC failures are not reported to the loop (mpc_cbf.py:10)."""
        torch = self.torch
        p = self._params(1)
        M, K, B = int(self.obs.shape[0]), self.num_constraints, self.B
        obs_sel = torch.empty((B, K, 7), dtype=self.tdtype, device=self.device)
        goal2 = torch.empty((B, 2), dtype=self.tdtype, device=self.device)
        u_ref = torch.empty((B, 2), dtype=self.tdtype, device=self.device)
        track = torch.empty(B, dtype=torch.int32, device=self.device)
        tX = torch.empty((n, B, 4), dtype=self.tdtype, device=self.device) if record else None
        tU = torch.empty((n, B, 2), dtype=self.tdtype, device=self.device) if record else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        obs_ptr = self.obs.data_ptr() if M else None
        for k in range(n):
            rc = self._lib.sc_tracking_select_batch(
                C.byref(p), B, M, self.X.data_ptr(), self.waypoints.data_ptr(), self.n_wp.data_ptr(),
                self.current_goal_index.data_ptr(), self.state_machine.data_ptr(), self.goal.data_ptr(), obs_ptr,
                self.ret.data_ptr(), obs_sel.data_ptr(), goal2.data_ptr(), u_ref.data_ptr(), track.data_ptr(), stream)
            _lib.check(rc, "sc_tracking_select_batch")
            Xm = self.X[:, :2].contiguous() if self.model == "SingleIntegrator2D" else self.X
            # OptimalDecayMPCCBF has five fixed obstacle slots (optimal_decay_mpc_cbf.py:249-252,333-339: padded_obs[:5]):
            # it sees the five nearest of the selected rows, MPCCBF all num_constraints of them
            obs_in = obs_sel[:, :5].contiguous() if (self.pos_controller_type == "optimal_decay_mpc_cbf" and K > 5) else obs_sel
            # Agents outside 'track' get u_ref and the reference does not solve for them (mpc_cbf.py:379-381).  Their slots of the batched
            # launch are handed ONE easy problem (_easy_problem: a short run to a goal 1 m ahead, dummy obstacle rows) instead of a solve
            # whose result is discarded and which may run the whole iteration budget: an agent turning on the spot (v = 0) makes a
            # degenerate NLP that crawled for up to 1700 iterations.  No host round trip: four selects on the device.
            tr = (track != 0).unsqueeze(1)
            xe, ge = self._easy_problem(Xm, goal2)
            X_in = torch.where(tr, Xm, xe).contiguous()
            goal_in = torch.where(tr, goal2, ge).contiguous()
            up_in = torch.where(tr, self.u_prev, torch.zeros_like(self.u_prev)).contiguous()
            obs_in = torch.where(tr.unsqueeze(2), obs_in, self._dummy_rows(obs_in)).contiguous()
            # kernel 13 serves superellipsoid rows for the two robots whose DT barrier has that branch (csrc/mpc_du_ms_se.hip); other robots' scenes
            # with such rows run on the condensed kernels as before
            ms = getattr(self, "mpc_ms", None)
            se_ok = self.model in ("DynamicUnicycle2D", "DoubleIntegrator2D")
            if ms is not None:
                ms.superellipsoids = bool(self.obs_has_superellipsoid and se_ok)
            mpc = ms if (ms is not None and (not self.obs_has_superellipsoid or se_ok)) else self.mpc
            out = mpc.solve(X_in, up_in, goal_in, obs_in)
            u_mpc, st = (out[0], out[2]) if self.pos_controller_type == "optimal_decay_mpc_cbf" else (out[0], out[1])
            its = out[3] if self.pos_controller_type == "optimal_decay_mpc_cbf" else out[2]
            self._raw_iters, self._raw_track = its, track                      # (what the launch ran, slots outside 'track' included)
            u = torch.where(tr, u_mpc, u_ref).contiguous()
            self.u_prev = torch.where(tr, u_mpc, self.u_prev).contiguous()
            self.mpc_status = torch.where(track != 0, st, self.mpc_status)      # (see BatchedQuadTrackingController: status / iterations of the last solve)
            self.mpc_iters = torch.where(track != 0, its, getattr(self, "mpc_iters", torch.zeros_like(its)))
            rc = self._lib.sc_tracking_apply_batch(
                C.byref(p), B, M, self.steps_done + k, self.X.data_ptr(), self.state_machine.data_ptr(), self.goal.data_ptr(), obs_ptr,
                u.data_ptr(), None, self.u_pos.data_ptr(), self.ret.data_ptr(), self.ret_step.data_ptr(), stream)
            _lib.check(rc, "sc_tracking_apply_batch")
            if record:
                tX[k] = self.X
                tU[k] = self.u_pos
        self.steps_done += n
        return (self.ret, tX, tU) if record else self.ret

    def run_all_steps(self, tf=30, chunk=200):
        """tracking.py:711-752 for every agent: run int(tf/dt) steps (agents stop at their first -1 / -2)."""
        total = int(tf / self.dt)
        done = 0
        while done < total:
            n = min(chunk, total - done)
            self.control_step(n)
            done += n
            if bool((self.ret != 0).all().item()):
                break
        return self.ret


class BatchedQuadTrackingController(BatchedTrackingController):
    """The batched closed loop for Quad2D (examples/test_tracking.py --model quad), Quad3D (--model quad3d) and VTOL2D
    (examples/test_vtol.py) with the reference's default position controller ``mpc_cbf``: per step  sc_quadtrack_select_batch -> one
    MPC-CBF launch for the batch (csrc/mpc_gn.hip / csrc/mpc_lin.hip / csrc/mpc_vtol_wave.hip) -> sc_quadtrack_apply_batch.
    VTOL2D: ``X0`` rows [x, z(, .)] (cruise at 5 m/s, tracking.py:94-99) or six states.  ``X0`` rows follow tracking.py:80-93 (Quad2D:
    [x, z(, .)] or six states; Quad3D: [x, y], [x, y, yaw], [x, y, z, yaw] or twelve states); waypoints are [x, y, z] rows for
    Quad3D (the example's third column -- a heading for the planar models -- is the altitude goal there, tracking.py:501-502)."""

    def __init__(self, X0, robot_spec, controller_type=None, dt=0.05, enable_rotation=True, obs=None, dyn_obs=False,
                 io_dtype="f64", device="cuda:0"):
        import torch
        self.torch = torch
        controller_type = controller_type or {"pos": "mpc_cbf"}
        self.pos_controller_type = controller_type.get("pos", "mpc_cbf")
        if self.pos_controller_type != "mpc_cbf":
            raise ValueError("Quad2D / Quad3D / VTOL2D closed loop: position controller 'mpc_cbf' (the reference's default)")
        if dyn_obs:
            raise ValueError("moving obstacle tables are stepped by the fused 'cbf_qp' rollout only")
        self.model = robot_spec["model"]
        self.q3 = self.model == "Quad3D"
        self.vt = self.model == "VTOL2D"
        self.robot_spec = complete_robot_spec(robot_spec)
        self.robot_spec.setdefault("exploration", False)
        self.integrator = False
        self.dt, self.enable_rotation, self.dyn_obs = float(dt), bool(enable_rotation), False
        self.device = torch.device(device)
        self.io_dtype = {"f32": _lib.DTYPE_F32, "f64": _lib.DTYPE_F64}[io_dtype]
        self.tdtype = torch.float32 if io_dtype == "f32" else torch.float64
        self.num_constraints = int(self.robot_spec.get("num_constraints", 10))
        self.reached_threshold = float(self.robot_spec.get("reached_threshold", 0.3))
        self.rotation_threshold = 0.1
        self.fov_angle = math.radians(float(self.robot_spec.get("fov_angle", 70.0)))
        self._lib = _lib.load()
        self.nx, self.nu, self.ng = (12, 4, 3) if self.q3 else ((6, 4, 2) if self.vt else (6, 2, 2))
        self._pos_cols = [0, 1, 2] if self.q3 else [0, 1]                  # position entries of a state row (Quad3D: x, y, z; planar: x, z)
        X0 = np.asarray(X0, dtype=np.float64)
        if X0.ndim == 1:
            X0 = X0[None, :]
        B = X0.shape[0]
        X = np.zeros((B, self.nx))
        if self.vt:                                            # tracking.py:94-99
            if X0.shape[1] in (2, 3):
                X[:, :2] = X0[:, :2]
                X[:, 3] = 5.0
            elif X0.shape[1] == 6:
                X = X0.copy()
            else:
                raise ValueError("Invalid initial state dimension for VTOL2D")
        elif not self.q3:                                      # tracking.py:80-84
            if X0.shape[1] in (2, 3):
                X[:, :2] = X0[:, :2]
            elif X0.shape[1] == 6:
                X = X0.copy()
            else:
                raise ValueError("Invalid initial state dimension for Quad2D")
        else:                                                  # tracking.py:85-93
            if X0.shape[1] == 2:
                X[:, :2] = X0
            elif X0.shape[1] == 3:
                X[:, 0], X[:, 1], X[:, 5] = X0[:, 0], X0[:, 1], X0[:, 2]
            elif X0.shape[1] == 4:
                X[:, 0], X[:, 1], X[:, 2], X[:, 5] = X0[:, 0], X0[:, 1], X0[:, 2], X0[:, 3]
            elif X0.shape[1] == 12:
                X = X0.copy()
            else:
                raise ValueError("Invalid initial state dimension for Quad3D")
        self.B = B
        t = lambda a, dt_=None: torch.tensor(a, dtype=dt_ or self.tdtype, device=self.device)
        self.X = t(X)
        self.state_machine = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.current_goal_index = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.goal = torch.zeros((B, 4), dtype=self.tdtype, device=self.device)            # gx, gy, gz, valid
        self.ret = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.ret_step = torch.full((B,), -1, dtype=torch.int32, device=self.device)
        self.u_pos = torch.zeros((B, self.nu), dtype=self.tdtype, device=self.device)
        self.set_obstacles(obs)
        self.waypoints = self.n_wp = None
        self.steps_done = 0
        if self.q3:
            from .position_control.mpc_cbf_linear import BatchedLinearMPCCBF as cls
        elif self.vt:
            # robot_spec['mpc_formulation']: 'multiple_shooting' (default: the NLP as do-mpc poses it, IPOPT's algorithm with its restoration phase,
            # csrc/mpc_vtol_ms.hip) or 'condensed' (single shooting, csrc/mpc_vtol_wave.hip)
            if self.robot_spec.get("mpc_formulation", "multiple_shooting") == "condensed":
                from .position_control.mpc_cbf_vtol import BatchedVtolMPCCBF as cls
            else:
                from .position_control.mpc_cbf_vtol_ms import BatchedVtolMSMPCCBF as cls
        else:
            from .position_control.mpc_cbf_gn import BatchedGnMPCCBF as cls
        # robot_spec['mpc_max_iter'] (not a reference key): iteration budget of a solve inside the loop.  The default is the reference solver's (IPOPT:
        # 3000), and ONE hard NLP of the batch then holds a control step for as long as its solve takes (measured on 256 VTOL2D aircraft
        # flying the example scene at once: 0.3 s per step while a few of them face an infeasible approach, 7 ms afterwards)
        kw = {"max_iter": int(self.robot_spec["mpc_max_iter"])} if "mpc_max_iter" in self.robot_spec else {}
        self.mpc = cls(self.robot_spec, dt=self.dt, io_dtype=io_dtype, **kw)
        self.u_prev = torch.zeros((B, self.nu), dtype=self.tdtype, device=self.device)
        self.mpc_status = torch.zeros(B, dtype=torch.int32, device=self.device)

    def set_waypoints(self, waypoints):
        """set_waypoints / filter_waypoints / first update_goal (tracking.py:197-262, 497-535) on the host."""
        torch = self.torch
        X = self.X.double().cpu().numpy()
        npos = 3 if self.q3 else 2
        shared = (isinstance(waypoints, np.ndarray) and waypoints.ndim == 2) or \
            (isinstance(waypoints, (list, tuple)) and len(waypoints) > 0 and np.ndim(waypoints[0]) == 1)
        lists = [np.asarray(waypoints, dtype=np.float64)] * self.B if shared else [np.asarray(w, dtype=np.float64) for w in waypoints]
        filt = []
        for i in range(self.B):
            wp = lists[i]
            if wp.shape[1] < npos:
                raise ValueError(f"{self.model} waypoints need {npos} columns")
            if len(wp) >= 2:
                aug = np.vstack((X[i, :npos], wp[:, :npos]))
                dist = np.linalg.norm(np.diff(aug, axis=0), axis=1)
                wp = aug[np.concatenate(([False], dist >= self.reached_threshold))]
            w3 = np.zeros((len(wp), 3))
            w3[:, :npos] = np.asarray(wp, dtype=np.float64)[:, :npos]
            filt.append(w3)
        W = max(1, max(len(w) for w in filt))
        wps = np.zeros((self.B, W, 3)); n_wp = np.zeros(self.B, dtype=np.int32); idx = np.zeros(self.B, dtype=np.int32)
        sm = np.zeros(self.B, dtype=np.int32); goal = np.zeros((self.B, 4))
        for i in range(self.B):
            w = filt[i]
            n_wp[i] = len(w)
            wps[i, : len(w)] = w
            g = None
            if len(w) > 0:
                if np.linalg.norm(X[i, :2] - w[0, :2]) < self.reached_threshold:
                    idx[i] = 1
                if idx[i] < len(w):
                    g = w[idx[i]]
            if g is not None:
                ang = math.atan2(g[1] - X[i, 1], g[0] - X[i, 0])
                # is_in_fov (robots/robot.py:854-872): always True for Quad2D; yaw = X[5] for Quad3D (robot.py:451-452)
                in_fov = (not self.q3) or abs(_wrap(ang - X[i, 5])) <= self.fov_angle / 2
                if not in_fov:
                    if self.robot_spec["exploration"]:
                        sm[i] = _lib.SM_ROTATE
                        goal[i] = [g[0], g[1], g[2], 1.0]
                    else:
                        sm[i] = _lib.SM_STOP
                else:
                    sm[i] = _lib.SM_TRACK
                    goal[i] = [g[0], g[1], g[2], 1.0]
        self.waypoints = torch.tensor(wps, dtype=self.tdtype, device=self.device).contiguous()
        self.n_wp = torch.tensor(n_wp, dtype=torch.int32, device=self.device)
        self.current_goal_index = torch.tensor(idx, dtype=torch.int32, device=self.device)
        self.state_machine = torch.tensor(sm, dtype=torch.int32, device=self.device)
        self.goal = torch.tensor(goal, dtype=self.tdtype, device=self.device).contiguous()
        self.ret.zero_()
        self.ret_step.fill_(-1)

    def _params(self, n_steps=1):
        rs = self.robot_spec
        p = _lib.QuadTrackParams()
        p.model = 1 if self.q3 else (2 if self.vt else 0)
        p.io_dtype = self.io_dtype
        p.max_waypoints = int(self.waypoints.shape[1])
        p.waypoints_shared = 0
        p.enable_rotation = 1 if self.enable_rotation else 0
        p.num_constraints = self.num_constraints
        p.dt, p.reached_threshold, p.rotation_threshold = self.dt, self.reached_threshold, self.rotation_threshold
        p.robot_radius = float(rs["radius"])
        p.mass = float(rs["mass"])
        if self.vt:
            for i, key in enumerate(_lib.VTOL_AIRFRAME_KEYS):
                p.airframe[i] = float(rs[key])
            p.pitch_limit = float(rs["pitch_max"])                 # tracking.py:493 compares |theta| [rad] with this number as given
        elif self.q3:
            p.Ix, p.Iy, p.Iz, p.L, p.nu = (float(rs[k]) for k in ("Ix", "Iy", "Iz", "L", "nu"))
            p.u_min, p.u_max = float(rs["u_min"]), float(rs["u_max"])
        else:
            p.inertia, p.f_min, p.f_max = float(rs["inertia"]), float(rs["f_min"]), float(rs["f_max"])
        return p

    def _easy_problem(self, X, goal):
        """The discarded slots' problem for the quadrotors (at rest, goal 1 m along x) and for VTOL2D (the cruise probe of
        tests/test_mpcvtol_gpu.py: 12 m/s at 10 m, goal 100 m ahead -- zero airspeed is a singular point of the aero model)."""
        e = getattr(self, "_easy", None)
        if e is None:
            xe = self.torch.zeros((1, self.nx), dtype=X.dtype, device=X.device)
            ge = self.torch.zeros((1, self.ng), dtype=goal.dtype, device=goal.device)
            ge[0, 0] = 1.0
            if self.vt:
                xe[0, 1], xe[0, 3] = 10.0, 12.0
                ge[0, 0], ge[0, 1] = 100.0, 10.0
            self._easy = e = (xe, ge)
        return e

    def control_step(self, n=1, record=False):
        torch = self.torch
        if self.waypoints is None:
            raise RuntimeError("call set_waypoints first")
        p = self._params()
        M, K, B = int(self.obs.shape[0]), self.num_constraints, self.B
        obs_sel = torch.empty((B, K, 7), dtype=self.tdtype, device=self.device)
        goal_c = torch.empty((B, self.ng), dtype=self.tdtype, device=self.device)
        u_ref = torch.empty((B, self.nu), dtype=self.tdtype, device=self.device)
        track = torch.empty(B, dtype=torch.int32, device=self.device)
        tX = torch.empty((n, B, self.nx), dtype=self.tdtype, device=self.device) if record else None
        tU = torch.empty((n, B, self.nu), dtype=self.tdtype, device=self.device) if record else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        obs_ptr = self.obs.data_ptr() if M else None
        for k in range(n):
            rc = self._lib.sc_quadtrack_select_batch(
                C.byref(p), B, M, self.X.data_ptr(), self.waypoints.data_ptr(), self.n_wp.data_ptr(), self.current_goal_index.data_ptr(),
                self.state_machine.data_ptr(), self.goal.data_ptr(), obs_ptr, self.ret.data_ptr(), obs_sel.data_ptr(), goal_c.data_ptr(),
                u_ref.data_ptr(), track.data_ptr(), stream)
            _lib.check(rc, "sc_quadtrack_select_batch")
            # (outside 'track' the slot gets a problem that is solved at once -- goal on the vehicle, dummy obstacle rows -- see
            # BatchedTrackingController._control_step_split)
            tr = (track != 0).unsqueeze(1)
            xe, ge = self._easy_problem(self.X, goal_c)
            X_in = torch.where(tr, self.X, xe).contiguous()
            goal_in = torch.where(tr, goal_c, ge).contiguous()
            up_in = torch.where(tr, self.u_prev, torch.zeros_like(self.u_prev)).contiguous()
            obs_in = torch.where(tr.unsqueeze(2), obs_sel, self._dummy_rows(obs_sel)).contiguous()
            out = self.mpc.solve(X_in, up_in, goal_in, obs_in)
            u_mpc, st = out[0], out[1]
            u = torch.where(tr, u_mpc, u_ref).contiguous()                       # mpc_cbf.py:379-381: u_ref passes through outside 'track'
            self.u_prev = torch.where(tr, u_mpc, self.u_prev).contiguous()
            # per-agent status / iteration count of the last MPC solve: the reference's `status` stays 'optimal' whatever IPOPT returned
            # (mpc_cbf.py:10); a caller that wants to know about unconverged steps reads these (SC_STATUS_*)
            self.mpc_status = torch.where(track != 0, st, self.mpc_status)
            self.mpc_iters = torch.where(track != 0, out[2], getattr(self, "mpc_iters", torch.zeros_like(out[2])))
            rc = self._lib.sc_quadtrack_apply_batch(
                C.byref(p), B, M, self.steps_done + k, self.X.data_ptr(), self.state_machine.data_ptr(), self.goal.data_ptr(), obs_ptr,
                u.data_ptr(), self.u_pos.data_ptr(), self.ret.data_ptr(), self.ret_step.data_ptr(), stream)
            _lib.check(rc, "sc_quadtrack_apply_batch")
            if record:
                tX[k] = self.X
                tU[k] = self.u_pos
        self.steps_done += n
        return (self.ret, tX, tU) if record else self.ret
