"""ctypes binding of the HIP library (include/safe_control_amd.h).

There is NO CPU fallback: if ``lib/libsafe_control_hip.so`` is missing or a
symbol is absent, importing a solver raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C safe_control_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAFE_CONTROL_AMD_LIB points the binding at another build of the same library (an integrator's own build; the allocator-guard build
# tests/test_codegen_guard_gpu.py compares with).  Still no fallback: the named file must exist and export every symbol.
LIB_PATH = os.environ.get("SAFE_CONTROL_AMD_LIB") or os.path.join(_HERE, "lib", "libsafe_control_hip.so")

SC_OK, SC_ERR_INVALID_ARGUMENT, SC_ERR_UNSUPPORTED, SC_ERR_HIP, SC_ERR_NO_DEVICE = 0, 1, 2, 3, 4            # sc_error (include/safe_control_amd.h)
STATUS_OPTIMAL, STATUS_INFEASIBLE, STATUS_INACCURATE, STATUS_BAD_OBSTACLE = 0, 1, 2, 3
STATUS_STRINGS = {0: "optimal", 1: "infeasible", 2: "optimal_inaccurate", 3: "bad_obstacle", 4: "needs_restoration"}
DTYPE_F32, DTYPE_F64 = 0, 1
CBF_MODE = {"cbf": 0, "hard": 1}
CBFQP_MAX_OBS = 32

MODEL_IDS = {
    "DynamicUnicycle2D": 0,
    "KinematicBicycle2D": 1,
    "KinematicBicycle2D_C3BF": 2,
    "KinematicBicycle2D_DPCBF": 3,
    "SingleIntegrator2D": 4,
    "DoubleIntegrator2D": 5,
    "Quad2D": 6,
    "Unicycle2D": 7,
}

STATE_DIM = {"Quad2D": 6}          # everything else: 4


class CbfQpParams(C.Structure):
    """Mirror of ``sc_cbfqp_params``."""
    _fields_ = [
        ("model_id", C.c_int32), ("io_dtype", C.c_int32), ("compute_dtype", C.c_int32),
        ("cbf_mode", C.c_int32), ("obs_shared", C.c_int32), ("state_dim", C.c_int32),
        ("robot_radius", C.c_double), ("dt", C.c_double),
        ("alpha1", C.c_double), ("alpha2", C.c_double),
        ("u_min", C.c_double * 2), ("u_max", C.c_double * 2),
        ("rear_ax_dist", C.c_double), ("mass", C.c_double),
    ]


class QuadTrackParams(C.Structure):
    """Mirror of ``sc_quadtrack_params``."""
    _fields_ = [
        ("model", C.c_int32), ("io_dtype", C.c_int32), ("max_waypoints", C.c_int32), ("waypoints_shared", C.c_int32),
        ("enable_rotation", C.c_int32), ("num_constraints", C.c_int32), ("reserved0", C.c_int32), ("reserved1", C.c_int32),
        ("dt", C.c_double), ("reached_threshold", C.c_double), ("rotation_threshold", C.c_double), ("robot_radius", C.c_double),
        ("mass", C.c_double), ("inertia", C.c_double), ("f_min", C.c_double), ("f_max", C.c_double),
        ("Ix", C.c_double), ("Iy", C.c_double), ("Iz", C.c_double), ("L", C.c_double), ("nu", C.c_double),
        ("u_min", C.c_double), ("u_max", C.c_double), ("airframe", C.c_double * 21), ("pitch_limit", C.c_double),
    ]


class BackupCbfParams(C.Structure):
    """Mirror of ``sc_backupcbf_params``."""
    _fields_ = [
        ("io_dtype", C.c_int32), ("n_steps", C.c_int32), ("bullet_shared", C.c_int32), ("reserved", C.c_int32),
        ("dt", C.c_double), ("backup_horizon", C.c_double), ("fd_eps", C.c_double),
        ("robot_radius", C.c_double), ("a_max", C.c_double), ("v_max", C.c_double), ("safety_margin", C.c_double),
        ("alpha", C.c_double), ("alpha_terminal", C.c_double), ("backup_kp", C.c_double), ("backup_kd", C.c_double),
        ("hallway_length", C.c_double), ("half_width", C.c_double),
        ("pocket_x_min", C.c_double), ("pocket_x_max", C.c_double), ("pocket_y_min", C.c_double), ("pocket_y_max", C.c_double),
        ("goal_x_min", C.c_double), ("goal_x_max", C.c_double),
        ("bullet_speed", C.c_double), ("bullet_length", C.c_double), ("bullet_width", C.c_double), ("bullet_start_x", C.c_double),
    ]


MPCCBF_MAX_HORIZON = 32


class RestoParams(C.Structure):
    """Mirror of ``sc_resto_params``: the feasibility-restoration phase of the MPC interior point."""
    _fields_ = [("rho", C.c_double), ("kappa", C.c_double), ("theta_tol", C.c_double), ("tol", C.c_double),
                ("small_alpha", C.c_double), ("small_iter", C.c_int32), ("max_entries", C.c_int32), ("slack_reset", C.c_int32),
                ("retry_max", C.c_int32), ("stall_theta", C.c_double), ("stall_iter", C.c_int32), ("gauss_newton", C.c_int32)]


def default_resto(**over):
    """IPOPT's penalty (1000), return at a tenth of the violation, certificate threshold 1e-6, restoration tolerance 1e-2 (in units of rho * violation),
    hand-over after 4 steps shorter than 0.02, at most two entries, slack reset in the restoration's line search, three damped retries of a
    failed restoration step, stall certificate after 40 iterations without 1 % less violation above 1e-3 (oracle/mpc_cbf.py: DEFAULTS)."""
    r = RestoParams(rho=1000.0, kappa=0.1, theta_tol=1e-6, tol=1e-2, small_alpha=0.02, small_iter=4, max_entries=2, slack_reset=1,
                    retry_max=3, stall_theta=1e-3, stall_iter=40)                # (round 4: damped retries, stall certificate)
    for k, v in over.items():
        setattr(r, k, v)
    return r


MPC_MAX_SLICES = 8
STATUS_PENDING = -1
# The reference hands its NLPs to IPOPT with default options (position_control/mpc_cbf.py:163-173): max_iter = 3000.  The batched
# classes serve that budget with continuation launches: the launch over the whole batch stops at FIRST_CAP iterations, the few
# unfinished solves continue in a second launch (sc_mpc_slices); a classify-only pre-pass starts the long solves first.
IPOPT_MAX_ITER = 3000
FIRST_CAP = 100


class MpcSlices(C.Structure):
    """Mirror of ``sc_mpc_slices``: the continuation launches of one sc_mpc*_solve_batch_sliced call."""
    _fields_ = [("n_caps", C.c_int32), ("it_stop", C.c_int32 * MPC_MAX_SLICES), ("order", C.c_int32), ("classify_first", C.c_int32),
                ("reserved", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


def make_slices(caps=(), order=True, classify_first=False):
    """An MpcSlices without its workspace (SlicedSolver attaches one)."""
    caps = [int(c) for c in caps]
    if len(caps) > MPC_MAX_SLICES:
        raise ValueError(f"at most {MPC_MAX_SLICES} iteration caps")
    sl = MpcSlices(n_caps=len(caps), order=1 if order else 0, classify_first=1 if classify_first else 0)
    for i, c in enumerate(caps):
        sl.it_stop[i] = c
    return sl


class SlicedSolver:
    """What the batched MPC classes share for their continuation launches: the schedule (``iter_slices``: iteration caps of the
    launches before the last one; ``classify_first``; ``order``) and a workspace tensor that is kept between calls and grows with
    the batch.  ``slices_for(B, need_bytes, device)`` returns the ctypes struct to pass, or None for a single launch."""

    def init_slices(self, iter_slices=None, classify_first=True, order=True):
        """iter_slices: None = the default schedule (FIRST_CAP), () = one launch"""
        self.iter_slices = (FIRST_CAP,) if iter_slices is None else tuple(int(c) for c in iter_slices)
        self.classify_first, self.order_slices = bool(classify_first), bool(order)
        self._slice_ws = None

    def slices_for(self, need_bytes_fn, device):
        import torch
        caps = [c for c in self.iter_slices if c < self.max_iter]
        if not caps and not self.classify_first:
            return None
        sl = make_slices(caps, self.order_slices, self.classify_first)
        need = int(need_bytes_fn())
        if self._slice_ws is None or self._slice_ws.numel() < need or self._slice_ws.device != device:
            self._slice_ws = torch.empty((need,), dtype=torch.uint8, device=device)
        sl.workspace, sl.workspace_bytes = self._slice_ws.data_ptr(), self._slice_ws.numel()
        return sl


class MpcCbfParams(C.Structure):
    """Mirror of ``sc_mpccbf_params``."""
    _fields_ = [
        ("model_id", C.c_int32), ("io_dtype", C.c_int32), ("horizon", C.c_int32), ("max_iter", C.c_int32),
        ("obs_shared", C.c_int32), ("acceptable_iter", C.c_int32), ("slack_reset", C.c_int32), ("superellipsoid_rows", C.c_int32),
        ("dt", C.c_double), ("Q", C.c_double * 4), ("R", C.c_double * 2),
        ("alpha1", C.c_double), ("alpha2", C.c_double), ("v_max", C.c_double), ("u_max", C.c_double * 2),
        ("robot_radius", C.c_double), ("beta", C.c_double), ("tol", C.c_double), ("acceptable_tol", C.c_double),
        ("mu_init", C.c_double), ("mu_min", C.c_double), ("resto", RestoParams),
        ("v_min", C.c_double), ("rear_ax_dist", C.c_double),
    ]


class OdCbfQpParams(C.Structure):
    """Mirror of ``sc_odcbfqp_params``."""
    _fields_ = [("qp", CbfQpParams), ("omega_ref", C.c_double * 2), ("p_sb", C.c_double * 2)]


class OdMpcCbfParams(C.Structure):
    """Mirror of ``sc_odmpccbf_params``."""
    _fields_ = [("mpc", MpcCbfParams), ("omega_ref", C.c_double * 2), ("p_sb", C.c_double * 2)]


class MpcLinParams(C.Structure):
    """Mirror of ``sc_mpclin_params``."""
    _fields_ = [
        ("io_dtype", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32), ("ng", C.c_int32), ("horizon", C.c_int32),
        ("max_iter", C.c_int32), ("obs_shared", C.c_int32), ("acceptable_iter", C.c_int32), ("circles_only", C.c_int32),
        ("optimal_decay", C.c_int32), ("slack_reset", C.c_int32), ("reserved", C.c_int32),
        ("alpha", C.c_double), ("robot_radius", C.c_double), ("beta", C.c_double), ("tol", C.c_double),
        ("acceptable_tol", C.c_double), ("mu_init", C.c_double), ("mu_min", C.c_double),
        ("Q", C.c_double * 12), ("R", C.c_double * 4), ("u_lo", C.c_double * 4), ("u_hi", C.c_double * 4),
        ("od_omega_ref", C.c_double), ("od_p_sb", C.c_double), ("resto", RestoParams),
    ]


class MpcGnParams(C.Structure):
    """Mirror of ``sc_mpcgn_params``."""
    _fields_ = [
        ("model_id", C.c_int32), ("io_dtype", C.c_int32), ("horizon", C.c_int32), ("max_iter", C.c_int32),
        ("obs_shared", C.c_int32), ("acceptable_iter", C.c_int32), ("circles_only", C.c_int32), ("slack_reset", C.c_int32),
        ("dt", C.c_double), ("Q", C.c_double * 6), ("R", C.c_double * 2), ("alpha1", C.c_double), ("alpha2", C.c_double),
        ("u_lo", C.c_double * 2), ("u_hi", C.c_double * 2), ("v_min", C.c_double), ("v_max", C.c_double),
        ("rear_ax_dist", C.c_double), ("mass", C.c_double), ("inertia", C.c_double), ("robot_radius", C.c_double),
        ("beta", C.c_double), ("tol", C.c_double), ("acceptable_tol", C.c_double), ("mu_init", C.c_double), ("mu_min", C.c_double),
        ("resto", RestoParams),
    ]


class MpcVtolParams(C.Structure):
    """Mirror of ``sc_mpcvtol_params``."""
    _fields_ = [
        ("io_dtype", C.c_int32), ("horizon", C.c_int32), ("max_iter", C.c_int32), ("obs_shared", C.c_int32),
        ("acceptable_iter", C.c_int32), ("slack_reset", C.c_int32), ("kernel", C.c_int32), ("reserved", C.c_int32),
        ("dt", C.c_double), ("Q", C.c_double * 6), ("R", C.c_double * 4), ("alpha1", C.c_double), ("alpha2", C.c_double),
        ("u_lo", C.c_double * 4), ("u_hi", C.c_double * 4), ("v_max", C.c_double), ("descent_speed_max", C.c_double),
        ("pitch_max", C.c_double), ("robot_radius", C.c_double), ("beta", C.c_double), ("tol", C.c_double),
        ("acceptable_tol", C.c_double), ("mu_init", C.c_double), ("mu_min", C.c_double), ("airframe", C.c_double * 21),
        ("resto", RestoParams),
    ]


VTOL_AIRFRAME_KEYS = ("mass", "inertia", "S_wing", "rho", "C_L0", "C_Lalpha", "M", "alpha_0", "C_Ldelta_e", "C_D0", "C_Dalpha",
                      "C_Ddelta_e", "C_m0", "C_malpha", "C_mdelta_e", "chord", "k_front", "k_rear", "k_pusher", "ell_f", "ell_r")


class OdMpcVtolParams(C.Structure):
    """Mirror of ``sc_odmpcvtol_params``."""
    _fields_ = [("mpc", MpcVtolParams), ("omega_ref", C.c_double * 2), ("p_sb", C.c_double * 2)]


class OdMpcGnParams(C.Structure):
    """Mirror of ``sc_odmpcgn_params``."""
    _fields_ = [("mpc", MpcGnParams), ("omega_ref", C.c_double * 2), ("p_sb", C.c_double * 2)]


MANIP_MAX_ROWS = 250


class ManipCbfQpParams(C.Structure):
    """Mirror of ``sc_manip_cbfqp_params``."""
    _fields_ = [
        ("io_dtype", C.c_int32), ("cbf_mode", C.c_int32), ("obs_shared", C.c_int32), ("num_rows", C.c_int32),
        ("link_steps", C.c_int32 * 3), ("reserved", C.c_int32),
        ("robot_radius", C.c_double), ("dt", C.c_double), ("alpha", C.c_double), ("w_max", C.c_double),
        ("beta", C.c_double), ("link_lengths", C.c_double * 3), ("base_pos", C.c_double * 2),
    ]


class ManipTrackingParams(C.Structure):
    """Mirror of ``sc_manip_tracking_params``."""
    _fields_ = [
        ("qp", ManipCbfQpParams),
        ("n_steps", C.c_int32), ("max_waypoints", C.c_int32), ("waypoints_shared", C.c_int32), ("enable_rotation", C.c_int32),
        ("step_offset", C.c_int32), ("reserved0", C.c_int32),
        ("Kp", C.c_double), ("reached_threshold", C.c_double), ("rotation_threshold", C.c_double),
    ]


STATUS_NEEDS_RESTO = 4          # sc_mpcvtol_ms_solve_batch: IPOPT would enter its restoration phase here (the host class re-solves with the condensed kernel)


class IpoptParams(C.Structure):
    """Mirror of ``sc_ipopt_params``: IPOPT's option names with IPOPT's documented defaults (oracle/ms_ipopt.py: OPTS)."""
    _fields_ = [("max_iter", C.c_int32), ("acceptable_iter", C.c_int32)] + [(k, C.c_double) for k in (
        "tol", "dual_inf_tol", "constr_viol_tol", "compl_inf_tol",
        "acceptable_tol", "acceptable_dual_inf_tol", "acceptable_constr_viol_tol", "acceptable_compl_inf_tol",
        "nlp_scaling_max_gradient", "nlp_scaling_min_value", "bound_relax_factor", "bound_push", "bound_frac", "constr_mult_init_max",
        "mu_init", "mu_linear_decrease_factor", "mu_superlinear_decrease_power", "barrier_tol_factor", "tau_min",
        "kappa_sigma", "kappa_d", "s_max",
        "theta_max_fact", "theta_min_fact", "eta_phi", "delta", "s_phi", "s_theta", "gamma_phi", "gamma_theta", "alpha_min_frac", "alpha_red_factor",
        "obj_max_inc",
        "first_hessian_perturbation", "min_hessian_perturbation", "max_hessian_perturbation", "perturb_inc_fact_first", "perturb_inc_fact",
        "perturb_dec_fact",
        "resto_penalty_parameter", "resto_proximity_weight", "required_infeasibility_reduction", "bound_mult_reset_threshold",
        "resto_failure_feasibility_threshold", "resto_theta_max_fact")] + [("resto_workspace", C.c_void_p), ("resto_workspace_bytes", C.c_int64),
                                                                      ("stall_alpha", C.c_double), ("stall_iter", C.c_int32), ("floor_iter", C.c_int32)]


IPOPT_DEFAULTS = dict(
    max_iter=3000, acceptable_iter=15, tol=1e-8, dual_inf_tol=1.0, constr_viol_tol=1e-4, compl_inf_tol=1e-4,
    acceptable_tol=1e-6, acceptable_dual_inf_tol=1e10, acceptable_constr_viol_tol=1e-2, acceptable_compl_inf_tol=1e-2,
    nlp_scaling_max_gradient=100.0, nlp_scaling_min_value=1e-8, bound_relax_factor=1e-8, bound_push=1e-2, bound_frac=1e-2,
    constr_mult_init_max=1e3, mu_init=0.1, mu_linear_decrease_factor=0.2, mu_superlinear_decrease_power=1.5, barrier_tol_factor=10.0,
    tau_min=0.99, kappa_sigma=1e10, kappa_d=1e-5, s_max=100.0, theta_max_fact=1e4, theta_min_fact=1e-4, eta_phi=1e-8, delta=1.0,
    s_phi=2.3, s_theta=1.1, gamma_phi=1e-8, gamma_theta=1e-5, alpha_min_frac=0.05, alpha_red_factor=0.5, obj_max_inc=5.0,
    first_hessian_perturbation=1e-4, min_hessian_perturbation=1e-20, max_hessian_perturbation=1e20, perturb_inc_fact_first=100.0,
    perturb_inc_fact=8.0, perturb_dec_fact=1.0 / 3.0,
    resto_penalty_parameter=1000.0, resto_proximity_weight=1.0, required_infeasibility_reduction=0.9, bound_mult_reset_threshold=1e3,
    resto_failure_feasibility_threshold=1e-6, resto_theta_max_fact=1e8, stall_alpha=1e-4, stall_iter=60,
    floor_iter=30)          # (stall_alpha / stall_iter / floor_iter: NOT IPOPT options, see sc_ipopt_params)


def default_ipopt(**over):
    p = IpoptParams()
    for k, v in dict(IPOPT_DEFAULTS, **over).items():
        setattr(p, k, v)
    return p


SM_IDLE, SM_TRACK, SM_STOP, SM_ROTATE = 0, 1, 2, 3
SM_NAMES = {0: "idle", 1: "track", 2: "stop", 3: "rotate"}
TRACKING_MAX_CONSTRAINTS = 16


class TrackingParams(C.Structure):
    """Mirror of ``sc_tracking_params``."""
    _fields_ = [
        ("qp", CbfQpParams),
        ("n_steps", C.c_int32), ("max_waypoints", C.c_int32), ("waypoints_shared", C.c_int32),
        ("enable_rotation", C.c_int32), ("dyn_obs", C.c_int32), ("num_constraints", C.c_int32),
        ("step_offset", C.c_int32), ("reserved0", C.c_int32),
        ("reached_threshold", C.c_double), ("rotation_threshold", C.c_double),
        ("v_max", C.c_double), ("v_min", C.c_double),
        ("k_omega", C.c_double), ("k_a", C.c_double), ("k_v", C.c_double),
        ("delta_max", C.c_double), ("wheel_base", C.c_double),
    ]


# every symbol include/safe_control_amd.h declares, with its ctypes signature
SYMBOLS = {
    "sc_version": (C.c_int, []),
    "sc_last_error": (C.c_char_p, []),
    "sc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sc_cbfqp_solve_batch": (C.c_int, [C.POINTER(CbfQpParams), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sc_cbfqp_solve_batch_host": (C.c_int, [C.POINTER(CbfQpParams), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "sc_mpcgn_solve_batch": (C.c_int, [C.POINTER(MpcGnParams), C.c_int64, C.c_int32] + [C.c_void_p] * 9),
    "sc_mpcgn_solve_batch_host": (C.c_int, [C.POINTER(MpcGnParams), C.c_int64, C.c_int32] + [C.c_void_p] * 8 + [C.c_int]),
    "sc_mpcvtol_workspace_bytes": (C.c_size_t, [C.POINTER(MpcVtolParams), C.c_int64, C.c_int32]),
    "sc_mpcvtol_solve_batch": (C.c_int, [C.POINTER(MpcVtolParams), C.c_int64, C.c_int32] + [C.c_void_p] * 9 + [C.c_size_t, C.c_void_p]),
    "sc_mpcvtol_solve_batch_host": (C.c_int, [C.POINTER(MpcVtolParams), C.c_int64, C.c_int32] + [C.c_void_p] * 8 + [C.c_int]),
    "sc_mpcvtol_ms_solve_batch": (C.c_int, [C.POINTER(MpcVtolParams), C.POINTER(IpoptParams), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_mpcvtol_ms_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "sc_mpccbf_ms_solve_batch": (C.c_int, [C.POINTER(MpcCbfParams), C.POINTER(IpoptParams), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_mpccbf_ms_lds_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sc_mpccbf_ms_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "sc_odmpcvtol_solve_batch": (C.c_int, [C.POINTER(OdMpcVtolParams), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_odmpcgn_solve_batch": (C.c_int, [C.POINTER(OdMpcGnParams), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_mpclin_model_doubles": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sc_mpclin_build_model": (C.c_int, [C.POINTER(MpcLinParams)] + [C.c_void_p] * 5),
    "sc_mpclin_solve_batch": (C.c_int, [C.POINTER(MpcLinParams), C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 9),
    "sc_odmpclin_solve_batch": (C.c_int, [C.POINTER(MpcLinParams), C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_mpclin_solve_batch_host": (C.c_int, [C.POINTER(MpcLinParams), C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 8 + [C.c_int]),
    "sc_manip_tracking_rollout_batch": (C.c_int, [C.POINTER(ManipTrackingParams), C.c_int64, C.c_int32] + [C.c_void_p] * 13),
    "sc_manip_cbfqp_solve_batch": (C.c_int, [C.POINTER(ManipCbfQpParams), C.c_int64, C.c_int32] + [C.c_void_p] * 8),
    "sc_manip_cbfqp_solve_batch_host": (C.c_int, [C.POINTER(ManipCbfQpParams), C.c_int64, C.c_int32] + [C.c_void_p] * 7 + [C.c_int]),
    "sc_mpccbf_solve_batch": (C.c_int, [C.POINTER(MpcCbfParams), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "sc_mpcgn_slices_workspace_bytes": (C.c_size_t, [C.POINTER(MpcGnParams), C.c_int64, C.c_int32]),
    "sc_mpcgn_solve_batch_sliced": (C.c_int, [C.POINTER(MpcGnParams), C.POINTER(MpcSlices), C.c_int64, C.c_int32] + [C.c_void_p] * 9),
    "sc_mpclin_slices_workspace_bytes": (C.c_size_t, [C.POINTER(MpcLinParams), C.c_int64, C.c_int32]),
    "sc_mpclin_solve_batch_sliced": (C.c_int, [C.POINTER(MpcLinParams), C.POINTER(MpcSlices), C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 9),
    "sc_mpcvtol_slices_workspace_bytes": (C.c_size_t, [C.POINTER(MpcVtolParams), C.c_int64, C.c_int32]),
    "sc_mpcvtol_solve_batch_sliced": (C.c_int, [C.POINTER(MpcVtolParams), C.POINTER(MpcSlices), C.c_int64, C.c_int32] + [C.c_void_p] * 9),
    "sc_mpccbf_slices_workspace_bytes": (C.c_size_t, [C.POINTER(MpcCbfParams), C.c_int64, C.c_int32]),
    "sc_mpccbf_solve_batch_sliced": (C.c_int, [C.POINTER(MpcCbfParams), C.POINTER(MpcSlices), C.c_int64, C.c_int32] + [C.c_void_p] * 9),
    "sc_odmpccbf_slices_workspace_bytes": (C.c_size_t, [C.POINTER(OdMpcCbfParams), C.c_int64, C.c_int32]),
    "sc_odmpccbf_solve_batch_sliced": (C.c_int, [C.POINTER(OdMpcCbfParams), C.POINTER(MpcSlices), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_odmpcgn_slices_workspace_bytes": (C.c_size_t, [C.POINTER(OdMpcGnParams), C.c_int64, C.c_int32]),
    "sc_odmpcgn_solve_batch_sliced": (C.c_int, [C.POINTER(OdMpcGnParams), C.POINTER(MpcSlices), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_odmpclin_solve_batch_sliced": (C.c_int, [C.POINTER(MpcLinParams), C.POINTER(MpcSlices), C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_odmpcvtol_ms_solve_batch": (C.c_int, [C.POINTER(OdMpcVtolParams), C.POINTER(IpoptParams), C.c_int64, C.c_int32] + [C.c_void_p] * 11),
    "sc_odmpcvtol_slices_workspace_bytes": (C.c_size_t, [C.POINTER(OdMpcVtolParams), C.c_int64, C.c_int32]),
    "sc_odmpcvtol_solve_batch_sliced": (C.c_int, [C.POINTER(OdMpcVtolParams), C.POINTER(MpcSlices), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_odmpccbf_solve_batch": (C.c_int, [C.POINTER(OdMpcCbfParams), C.c_int64, C.c_int32] + [C.c_void_p] * 10),
    "sc_odmpccbf_solve_batch_host": (C.c_int, [C.POINTER(OdMpcCbfParams), C.c_int64, C.c_int32] + [C.c_void_p] * 9 + [C.c_int]),
    "sc_odcbfqp_solve_batch": (C.c_int, [C.POINTER(OdCbfQpParams), C.c_int64] + [C.c_void_p] * 9),
    "sc_odcbfqp_solve_batch_host": (C.c_int, [C.POINTER(OdCbfQpParams), C.c_int64] + [C.c_void_p] * 8 + [C.c_int]),
    "sc_neighbor_obstacles_batch": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "sc_neighbor_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int64, C.c_int64, C.c_int32]),
    "sc_neighbor_obstacles_batch_ws": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sc_quadtrack_select_batch": (C.c_int, [C.POINTER(QuadTrackParams), C.c_int64, C.c_int32] + [C.c_void_p] * 13),
    "sc_quadtrack_apply_batch": (C.c_int, [C.POINTER(QuadTrackParams), C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 9),
    "sc_backupcbf_solve_batch": (C.c_int, [C.POINTER(BackupCbfParams), C.c_int64] + [C.c_void_p] * 10),
    "sc_backupcbf_rollout_batch": (C.c_int, [C.POINTER(BackupCbfParams), C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 9),
    "sc_tracking_rollout_batch": (C.c_int, [C.POINTER(TrackingParams), C.c_int64, C.c_int32] + [C.c_void_p] * 13),
    "sc_tracking_select_batch": (C.c_int, [C.POINTER(TrackingParams), C.c_int64, C.c_int32] + [C.c_void_p] * 13),
    "sc_tracking_apply_batch": (C.c_int, [C.POINTER(TrackingParams), C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 10),
    "sc_mpccbf_solve_batch_host": (C.c_int, [C.POINTER(MpcCbfParams), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int]),
}

_lib = None

# SC_VERSION_MAJOR * 1000 + SC_VERSION_MINOR of the header the ctypes mirrors above were written for
ABI_VERSION = 9


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library or raise HipLibraryError (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `make -C safe_control_amd/csrc` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    got = lib.sc_version()
    if got != ABI_VERSION:
        raise HipLibraryError(
            f"{LIB_PATH} reports ABI version {got}, this binding was written for {ABI_VERSION}: a stale library would misread the "
            "parameter structs -- rebuild it with `make -C safe_control_amd/csrc`")
    _lib = lib
    return lib


def check(rc, what):
    if rc != SC_OK:
        msg = load().sc_last_error()
        raise HipLibraryError(f"{what} failed (sc_error {rc}): {msg.decode() if msg else ''}")
