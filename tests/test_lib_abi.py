"""CPU tests: the C-ABI library loads and exports every symbol include/safe_control_amd.h declares.

No compute calls (there is no GPU here); argument validation happens before any HIP call,
so error paths can be exercised.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from safe_control_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "safe_control_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.sc_version() == _lib.ABI_VERSION == 9


def test_params_struct_layout_matches_header():
    # 6 x int32 + 10 x double, no padding surprises
    assert C.sizeof(_lib.CbfQpParams) == 6 * 4 + 10 * 8
    assert _lib.CbfQpParams.robot_radius.offset == 24
    assert _lib.CbfQpParams.rear_ax_dist.offset == 24 + 8 * 8
    assert _lib.CbfQpParams.mass.offset == 24 + 9 * 8


def test_argument_validation_without_gpu():
    lib = _lib.load()
    p = _lib.CbfQpParams()
    p.model_id, p.io_dtype, p.compute_dtype, p.dt = 0, 0, 1, 0.05
    one = np.zeros(64)
    ptr = one.ctypes.data
    assert lib.sc_cbfqp_solve_batch(None, 1, 8, ptr, ptr, ptr, None, ptr, ptr, None, None) == 1
    assert lib.sc_cbfqp_solve_batch(C.byref(p), 1, 0, ptr, ptr, ptr, None, ptr, ptr, None, None) == 1
    assert lib.sc_cbfqp_solve_batch(C.byref(p), 1, 33, ptr, ptr, ptr, None, ptr, ptr, None, None) == 2
    assert b"SC_CBFQP_MAX_OBS" in lib.sc_last_error()
    assert lib.sc_cbfqp_solve_batch(C.byref(p), 1, 8, None, ptr, ptr, None, ptr, ptr, None, None) == 1
    p.model_id = 9
    assert lib.sc_cbfqp_solve_batch(C.byref(p), 1, 8, ptr, ptr, ptr, None, ptr, ptr, None, None) == 1
    p.model_id, p.io_dtype, p.compute_dtype = 0, 1, 0
    assert lib.sc_cbfqp_solve_batch(C.byref(p), 1, 8, ptr, ptr, ptr, None, ptr, ptr, None, None) == 2
    p.model_id, p.io_dtype, p.compute_dtype = 2, 0, 1          # KB family needs rear_ax_dist
    assert lib.sc_cbfqp_solve_batch(C.byref(p), 1, 8, ptr, ptr, ptr, None, ptr, ptr, None, None) == 1
    # B == 0 is a no-op
    p.model_id = 0
    assert lib.sc_cbfqp_solve_batch(C.byref(p), 0, 8, None, None, None, None, None, None, None, None) == 0


def test_argument_validation_of_the_newer_entry_points_without_gpu():
    """Manipulator2D CBF-QP, linear-model MPC-CBF, step()-barrier MPC-CBF: everything is rejected before a launch."""
    import safe_control_amd as sca
    from safe_control_amd.position_control import manipulator_cbf_qp as MQ, mpc_cbf_gn as GN, mpc_cbf_linear as ML
    from safe_control_amd.robots.linear_models import linear_model
    lib = _lib.load()
    buf = np.zeros(4096)
    ptr = buf.ctypes.data
    # Manipulator2D
    spec = sca.complete_robot_spec({"model": "Manipulator2D"})
    p = MQ.make_params(spec, 1.0, 0.05, 0.25, _lib.DTYPE_F64, 150, (0.0, 0.0))
    assert lib.sc_manip_cbfqp_solve_batch(C.byref(p), 0, 3, None, None, None, None, None, None, None, None) == 0      # B == 0
    assert lib.sc_manip_cbfqp_solve_batch(C.byref(p), 1, 0, ptr, ptr, ptr, None, ptr, ptr, None, None) == 1
    assert lib.sc_manip_cbfqp_solve_batch(C.byref(p), 1, 3, None, ptr, ptr, None, ptr, ptr, None, None) == 1
    p.num_rows = 251
    assert lib.sc_manip_cbfqp_solve_batch(C.byref(p), 1, 3, ptr, ptr, ptr, None, ptr, ptr, None, None) == 2
    assert b"SC_MANIP_MAX_ROWS" in lib.sc_last_error()
    p.num_rows, p.link_steps[1] = 150, 0
    assert lib.sc_manip_cbfqp_solve_batch(C.byref(p), 1, 3, ptr, ptr, ptr, None, ptr, ptr, None, None) == 1
    # linear models
    mdl = linear_model(sca.complete_robot_spec({"model": "Quad3D"}), 0.05)
    q = ML.make_params(mdl, mdl["cbf_param"], 10, 0.25, _lib.DTYPE_F64)
    blob = ML.build_model_blob(lib, q, mdl)
    bp = blob.ctypes.data
    assert lib.sc_mpclin_solve_batch(C.byref(q), bp, 0, 8, None, None, None, None, None, None, None, None, None) == 0
    assert lib.sc_mpclin_solve_batch(C.byref(q), None, 1, 8, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 1     # no model blob
    assert lib.sc_mpclin_solve_batch(C.byref(q), bp, 1, 0, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 1
    assert lib.sc_mpclin_solve_batch(C.byref(q), bp, 1, 200, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 2     # LDS
    q.u_hi[2] = q.u_lo[2]
    assert lib.sc_mpclin_solve_batch(C.byref(q), bp, 1, 8, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 1
    q.u_hi[2], q.nu = 10.0, 5
    assert lib.sc_mpclin_build_model(C.byref(q), ptr, ptr, ptr, ptr, ptr) == 1
    # step()-barrier models
    gspec = sca.complete_robot_spec({"model": "Quad2D"})
    mc = GN.model_constants(gspec)
    g = GN.make_params(gspec, mc, mc["cbf_param"], 10, 0.05, 0.25, _lib.DTYPE_F64)
    assert lib.sc_mpcgn_solve_batch(C.byref(g), 0, 8, None, None, None, None, None, None, None, None, None) == 0
    assert lib.sc_mpcgn_solve_batch(C.byref(g), 1, 8, ptr, ptr, None, ptr, ptr, ptr, None, None, None) == 1
    g.model_id = _lib.MODEL_IDS["Unicycle2D"]
    assert lib.sc_mpcgn_solve_batch(C.byref(g), 1, 8, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 2           # sc_mpccbf serves it
    for name in ("KinematicBicycle2D", "KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"):                      # need their own spec
        g.model_id = _lib.MODEL_IDS[name]
        assert lib.sc_mpcgn_solve_batch(C.byref(g), 1, 8, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 1       # rear_ax_dist = 0
    g.model_id, g.horizon = _lib.MODEL_IDS["Quad2D"], 33
    assert lib.sc_mpcgn_solve_batch(C.byref(g), 1, 8, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 2
    g.horizon, g.mass = 10, 0.0
    assert lib.sc_mpcgn_solve_batch(C.byref(g), 1, 8, ptr, ptr, ptr, ptr, ptr, ptr, None, None, None) == 1


def test_argument_validation_of_the_round2_entry_points_without_gpu():
    """Backup-CBF QP, quadrotor select / apply, optimal-decay linear-model MPC: rejected before a launch; host tables equal the oracle's."""
    from oracle import backup_cbf as OB
    from safe_control_amd.position_control import backup_cbf_qp as BK
    lib = _lib.load()
    buf = np.zeros(4096)
    ptr = buf.ctypes.data
    env = BK.default_evade_env()
    oe = OB.default_env()
    assert all(env[k] == oe[k] for k in BK.ENV_KEYS)                       # the host mirror of EvadeEnv == the pinned oracle's
    p = BK.make_params(env, {"radius": 0.5, "a_max": 2.0, "v_max": 1.5, "safety_margin": 0.5}, 0.1, 12.0, _lib.DTYPE_F64)
    assert p.n_steps == 120 and p.fd_eps == 1e-5 and (p.alpha, p.alpha_terminal) == (1.0, 2.0)
    assert lib.sc_backupcbf_solve_batch(C.byref(p), 0, None, None, None, None, None, None, None, None, None, None) == 0     # B == 0
    assert lib.sc_backupcbf_solve_batch(None, 1, ptr, None, ptr, ptr, ptr, None, None, None, None, None) == 1
    assert lib.sc_backupcbf_solve_batch(C.byref(p), 1, None, None, ptr, ptr, ptr, None, None, None, None, None) == 1
    p.n_steps = 129
    assert lib.sc_backupcbf_solve_batch(C.byref(p), 1, ptr, None, ptr, ptr, ptr, None, None, None, None, None) == 2
    p.n_steps, p.pocket_x_max = 120, p.pocket_x_min
    assert lib.sc_backupcbf_solve_batch(C.byref(p), 1, ptr, None, ptr, ptr, ptr, None, None, None, None, None) == 1
    p = BK.make_params(env, {}, 0.1, 12.0, _lib.DTYPE_F64)
    assert lib.sc_backupcbf_rollout_batch(C.byref(p), 1, -1, 0, ptr, ptr, ptr, ptr, None, None, ptr, ptr, None) == 1
    assert lib.sc_backupcbf_rollout_batch(C.byref(p), 1, 1, 0, ptr, ptr, ptr, ptr, None, None, None, ptr, None) == 1       # ret is NULL
    # quadrotor select / apply
    q = _lib.QuadTrackParams()
    q.model, q.io_dtype, q.max_waypoints, q.num_constraints = 1, _lib.DTYPE_F64, 2, 10
    q.dt, q.mass, q.Ix, q.Iy, q.Iz, q.L, q.nu = 0.05, 3.0, 0.5, 0.5, 0.5, 0.3, 0.1
    iptr = np.zeros(64, dtype=np.int32).ctypes.data
    assert lib.sc_quadtrack_select_batch(C.byref(q), 0, 0, *([None] * 13)) == 0
    assert lib.sc_quadtrack_select_batch(C.byref(q), 1, 0, None, ptr, iptr, iptr, iptr, ptr, None, iptr, ptr, ptr, ptr, iptr, None) == 1
    q.num_constraints = 17
    assert lib.sc_quadtrack_select_batch(C.byref(q), 1, 0, ptr, ptr, iptr, iptr, iptr, ptr, None, iptr, ptr, ptr, ptr, iptr, None) == 2
    q.num_constraints, q.model = 10, 2
    assert lib.sc_quadtrack_apply_batch(C.byref(q), 1, 0, 0, ptr, iptr, ptr, None, ptr, ptr, iptr, iptr, None) == 1
    q.model, q.L = 1, 0.0
    assert lib.sc_quadtrack_apply_batch(C.byref(q), 1, 0, 0, ptr, iptr, ptr, None, ptr, ptr, iptr, iptr, None) == 1
    q.model, q.inertia, q.robot_radius = 0, 0.01, 0.25
    assert lib.sc_quadtrack_apply_batch(C.byref(q), 1, 1, 0, ptr, iptr, ptr, None, ptr, ptr, iptr, iptr, None) == 1       # M > 0 without a table


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()


def test_product_never_imports_oracle():
    """The shipped package must not reference the oracle (no CPU fallback)."""
    pkg = os.path.join(ROOT, "safe_control_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f


def test_header_is_plain_c(tmp_path):
    """include/safe_control_amd.h is the boundary a C / cgo / JNI caller binds: it must compile as C99 on its own and its
    structs must have the sizes the ctypes mirrors assume."""
    import subprocess
    import safe_control_amd  # noqa: F401
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "safe_control_amd.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sc_cbfqp_params), sizeof(sc_mpccbf_params), sizeof(sc_tracking_params),\n'
                   '         sizeof(sc_manip_cbfqp_params), sizeof(sc_mpclin_params), sizeof(sc_mpcgn_params), sizeof(sc_odmpccbf_params));\n'
                   '  printf("%zu %zu %zu\\n", sizeof(sc_manip_tracking_params), sizeof(sc_odcbfqp_params), sizeof(sc_backupcbf_params));\n'
                   '  printf("%zu\\n", sizeof(sc_quadtrack_params));\n'
                   '  printf("%zu %zu %zu %zu\\n", sizeof(sc_resto_params), sizeof(sc_mpcvtol_params), sizeof(sc_odmpcgn_params), sizeof(sc_mpc_slices));\n'
                   '  printf("%zu\\n", sizeof(sc_odmpcvtol_params));\n'
                   '  printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(sc_mpccbf_params, resto), offsetof(sc_mpclin_params, resto), offsetof(sc_mpcgn_params, resto),\n'
                   '         offsetof(sc_mpcvtol_params, resto), offsetof(sc_mpcvtol_params, airframe), offsetof(sc_quadtrack_params, airframe));\n'
                   '  printf("%zu %zu %zu\\n", offsetof(sc_mpc_slices, order), offsetof(sc_mpc_slices, workspace), offsetof(sc_odmpcgn_params, omega_ref));\n'
                   '  printf("%zu %zu\\n", offsetof(sc_odmpcvtol_params, omega_ref), offsetof(sc_odmpcvtol_params, p_sb));\n'
                   '  printf("%zu %zu %zu\\n", offsetof(sc_resto_params, retry_max), offsetof(sc_resto_params, stall_theta), offsetof(sc_resto_params, stall_iter));\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    mirrors = [_lib.CbfQpParams, _lib.MpcCbfParams, _lib.TrackingParams, _lib.ManipCbfQpParams, _lib.MpcLinParams, _lib.MpcGnParams,
               _lib.OdMpcCbfParams, _lib.ManipTrackingParams, _lib.OdCbfQpParams, _lib.BackupCbfParams, _lib.QuadTrackParams,
               _lib.RestoParams, _lib.MpcVtolParams, _lib.OdMpcGnParams, _lib.MpcSlices, _lib.OdMpcVtolParams]
    assert sizes[:len(mirrors)] == [C.sizeof(m) for m in mirrors]
    # field offsets of the blocks a padding or order mismatch would move
    offs = [_lib.MpcCbfParams.resto.offset, _lib.MpcLinParams.resto.offset, _lib.MpcGnParams.resto.offset, _lib.MpcVtolParams.resto.offset,
            _lib.MpcVtolParams.airframe.offset, _lib.QuadTrackParams.airframe.offset, _lib.MpcSlices.order.offset, _lib.MpcSlices.workspace.offset,
            _lib.OdMpcGnParams.omega_ref.offset, _lib.OdMpcVtolParams.omega_ref.offset, _lib.OdMpcVtolParams.p_sb.offset,
            _lib.RestoParams.retry_max.offset, _lib.RestoParams.stall_theta.offset, _lib.RestoParams.stall_iter.offset]
    assert sizes[len(mirrors):] == offs


def test_argument_validation_of_the_round4_entry_points_without_gpu():
    """Continuation launches (sc_mpc_slices), the restoration's retry / stall fields and optimal-decay MPC-CBF for VTOL2D: bad arguments are
    rejected before anything is launched."""
    from safe_control_amd.position_control import mpc_cbf as MC, mpc_cbf_vtol as MV
    from safe_control_amd.robots.spec import complete_robot_spec
    lib = _lib.load()
    buf = np.zeros(4096)
    ptr = buf.ctypes.data
    iptr = np.zeros(64, dtype=np.int32).ctypes.data
    spec = complete_robot_spec({"model": "DynamicUnicycle2D"})
    Q, R = MC.default_mpc_weights("DynamicUnicycle2D")
    p = MC.make_params(spec, {"alpha1": 0.15, "alpha2": 0.15}, Q, R, 10, 0.05, 0.25, _lib.DTYPE_F64)
    assert (p.resto.retry_max, p.resto.stall_iter, p.resto.stall_theta) == (3, 40, 1e-3)
    args = (ptr, ptr, ptr, ptr, ptr, iptr, iptr, None)
    sl = _lib.make_slices([100])
    assert lib.sc_mpccbf_solve_batch_sliced(C.byref(p), C.byref(_lib.make_slices([])), 0, 8, *args, None) == 0    # B == 0, one plain launch
    assert lib.sc_mpccbf_solve_batch_sliced(C.byref(p), C.byref(sl), 4, 8, *args, None) == 1                       # caps without a workspace
    bad = _lib.make_slices([20, 10])
    assert lib.sc_mpccbf_solve_batch_sliced(C.byref(p), C.byref(bad), 4, 8, *args, None) == 1                      # caps must increase
    need = lib.sc_mpccbf_slices_workspace_bytes(C.byref(p), 4, 8)
    assert need > 4 * 8 * (16 + 2 * 20)                                                                           # a state record per problem
    p.resto.retry_max = 9
    assert lib.sc_mpccbf_solve_batch(C.byref(p), 4, 8, *args, None) == 1
    p.resto.retry_max, p.resto.stall_iter, p.resto.stall_theta = 3, 10, 0.0
    assert lib.sc_mpccbf_solve_batch(C.byref(p), 4, 8, *args, None) == 1
    # optimal-decay MPC-CBF for VTOL2D
    vspec = complete_robot_spec({"model": "VTOL2D"})
    q = MV.make_od_params(vspec, dict(MV.OD_CBF_VTOL), 30, 0.05, vspec["radius"], _lib.DTYPE_F64)
    assert q.mpc.alpha1 == 0.35 and q.p_sb[0] == 10.0 and q.omega_ref[1] == 1.0 and q.mpc.resto.stall_iter == 0
    vargs = (ptr, ptr, ptr, ptr, ptr, ptr, iptr, iptr, None)
    assert lib.sc_odmpcvtol_solve_batch(C.byref(q), 0, 8, *vargs, None) == 0                                       # B == 0
    assert lib.sc_odmpcvtol_solve_batch(None, 1, 8, *vargs, None) == 1
    assert lib.sc_odmpcvtol_solve_batch(C.byref(q), 1, 17, *vargs, None) != 0                                      # K > 16
    q.p_sb[1] = 0.0
    assert lib.sc_odmpcvtol_solve_batch(C.byref(q), 1, 8, *vargs, None) == 1
    q.p_sb[1], q.mpc.kernel = 10.0, 1
    assert lib.sc_odmpcvtol_solve_batch(C.byref(q), 1, 8, *vargs, None) == 2                                       # the lane kernel is retired
    q.mpc.kernel, q.mpc.resto.stall_iter = 0, 40
    assert lib.sc_odmpcvtol_solve_batch(C.byref(q), 1, 8, *vargs, None) == 2                                       # the VTOL2D kernels keep round 3's restoration
