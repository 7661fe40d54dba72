#!/usr/bin/env python3
"""Golden vectors for the Backup-CBF QP from the reference's own code.

Run ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_backup.py

Executed verbatim from the reference (imported through tests/golden/_ref_import.py with the recording cvxpy stand-in
tests/golden/_cvx_record.py): ``BackupCBF.solve_control_problem`` (position_control/backup_cbf_qp.py:563-794) with its
rollout + finite-difference sensitivities (:236-320), ``_h_safety`` / ``_h_terminal`` and their finite-difference
gradients (:343-560), the row assembly (:620-676) and the scaled QP statement (:678-735), on the evade scenario of
examples/evade/test_evade.py --algo backupcbf: ``EvadeEnv`` (envs/evade_env.py), ``DoubleIntegrator2D``,
``EvadeBackupController`` (position_control/backup_controller.py:420-572) and the example's nominal controller.

NOT produced by the reference: the QP minimiser (OSQP is absent; the recorded problem is solved by the oracle's exact
active-set solver -- unique minimiser of a strictly convex QP).

Writes tests/golden/backup_cbf.npz:
  cases   single calls at drawn (state, bullet position): phi, S, the recorded QP rows (A x >= b in scaled variables,
          box rows last), u_ref, u_safe, status flags, h_min
  loop    the example's closed loop (state, control, bullet position, using_backup per step) until the goal."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402

_ref_import.install()
import _cvx_record  # noqa: E402

sys.modules["cvxpy"] = _cvx_record

from safe_control.envs.evade_env import EvadeEnv  # noqa: E402
from safe_control.robots.double_integrator2D import DoubleIntegrator2D  # noqa: E402
from safe_control.position_control.backup_controller import EvadeBackupController  # noqa: E402
from safe_control.position_control.backup_cbf_qp import BackupCBF  # noqa: E402

sys.path.insert(0, os.path.join(_ref_import.REFERENCE_ROOT, "examples", "evade"))
_shield = type(sys)("safe_control.shielding")               # the example imports the two shields it does not use here
sys.modules.setdefault("safe_control.shielding", _shield)
for _n in ("gatekeeper", "mps"):
    _m = type(sys)(f"safe_control.shielding.{_n}")
    _m.Gatekeeper = _m.MPS = object
    sys.modules.setdefault(f"safe_control.shielding.{_n}", _m)
import test_evade as EV  # noqa: E402  (examples/evade/test_evade.py: configuration classes + the nominal controller)

MAXROWS = 128


def build(dt=0.1, horizon=12.0):
    cfg = EV.TestConfig(algo_type="backupcbf")
    cfg.simulation.dt, cfg.simulation.backup_horizon_time = dt, horizon
    e = cfg.env
    env = EvadeEnv(hallway_length=e.hallway_length, hallway_width=e.hallway_width, pocket_x=e.pocket_x,
                   pocket_length=e.pocket_length, pocket_width=e.pocket_width, goal_length=e.goal_length,
                   bullet_speed=e.bullet_speed, bullet_length=e.bullet_length, bullet_start_x=e.bullet_start_x)
    env._draw_bullet_bill = lambda: None                      # no figure
    spec = cfg.robot.to_dict()
    spec["safety_margin"] = cfg.simulation.safety_margin
    goal_bounds = {"x_min": env.goal_x_min, "x_max": env.goal_x_max, "y_min": -env.half_width, "y_max": env.half_width}
    nominal = EV.EvadeNominalController(spec)
    backup = EvadeBackupController(spec, dt, env.get_pocket_center(), env.get_pocket_bounds(), goal_bounds)
    dyn = DoubleIntegrator2D(dt, spec)
    sh = BackupCBF(robot=dyn, robot_spec=spec, dt=dt, backup_horizon=horizon, ax=None)
    sh.set_backup_controller(backup)
    sh.set_environment(env)

    def get_obstacles(t=0.0):                                 # examples/evade/test_evade.py:373-384
        b = env.get_bullet_state()
        if not b["active"]:
            return None
        f = b.copy()
        f["x"] = b["x"] + b["vx"] * t
        return f

    sh.set_moving_obstacles(get_obstacles)
    return cfg, env, spec, nominal, backup, dyn, sh


def one_call(sh, nominal, state):
    """solve_control_problem at `state` with the example's one-step nominal reference; returns the recorded pieces."""
    _cvx_record.LAST.clear()
    u_nom = nominal.compute_control(state.reshape(-1, 1)).flatten()
    sh.set_nominal_trajectory(None, np.tile(u_nom, (3, 1)))        # [T, 2] like the example's rollout; only row 0 is read (:179-181)
    phi, S = sh._integrate_backup_trajectory(state.copy())
    u = sh.solve_control_problem(state.reshape(-1, 1)).flatten()
    L = _cvx_record.LAST
    rows = np.zeros((MAXROWS, 3))
    n_rows = 0
    qp_status = -1                                            # -1: no QP was stated (no rows survived the |lhs| filter)
    if "A" in L:
        n_rows = L["A"].shape[0] - 4                         # the four box rows come last
        rows[:n_rows, 0:2] = L["A"][:n_rows]
        rows[:n_rows, 2] = L["b"][:n_rows]
        qp_status = int(L["status"])
    return dict(phi=phi, S=S, rows=rows, n_rows=n_rows, qp_status=qp_status, u_nom=u_nom, u=u,
                using_backup=bool(sh._using_backup), h_min=float(sh._last_h_min))


def gen():
    out = {}
    # ---- single calls ------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(20260113)
    for tag, dt, hor, n_cases in (("a", 0.1, 12.0, 14), ("b", 0.1, 4.0, 10), ("c", 0.05, 2.0, 8)):
        cfg, env, spec, nominal, backup, dyn, sh = build(dt, hor)
        X, BX, keys = [], [], None
        rec = []
        for i in range(n_cases):
            kind = i % 5
            if kind == 0:      # hallway, bullet close behind
                x = np.array([rng.uniform(8, 50), rng.uniform(-1.2, 1.2), rng.uniform(0, 1.5), rng.uniform(-0.3, 0.3)])
                bx = x[0] - rng.uniform(4, 14)
            elif kind == 1:    # below the pocket
                x = np.array([rng.uniform(26.5, 33.5), rng.uniform(-1.0, 1.4), rng.uniform(-0.5, 1.0), rng.uniform(-0.3, 0.8)])
                bx = x[0] - rng.uniform(3, 20)
            elif kind == 2:    # inside the pocket
                x = np.array([rng.uniform(26.5, 33.5), rng.uniform(2.8, 5.2), rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4)])
                bx = rng.uniform(0, 60)
            elif kind == 3:    # bullet far away or ahead
                x = np.array([rng.uniform(5, 50), rng.uniform(-1.0, 1.0), rng.uniform(0.5, 1.5), rng.uniform(-0.2, 0.2)])
                bx = x[0] + rng.uniform(6, 20) if rng.uniform() < 0.5 else -10.0
            else:              # near the goal
                x = np.array([rng.uniform(50, 58.5), rng.uniform(-1.0, 1.0), rng.uniform(0.0, 1.5), rng.uniform(-0.2, 0.2)])
                bx = x[0] - rng.uniform(5, 30)
            env.bullet_x = float(bx)
            r = one_call(sh, nominal, x)
            rec.append(r); X.append(x); BX.append(bx)
        out[f"{tag}_dt"], out[f"{tag}_horizon"] = dt, hor
        out[f"{tag}_X"], out[f"{tag}_bullet_x"] = np.array(X), np.array(BX)
        for k in rec[0]:
            out[f"{tag}_{k}"] = np.array([r[k] for r in rec])
        print(tag, "cases", n_cases, "rows", out[f"{tag}_n_rows"], "qp", out[f"{tag}_qp_status"], "backup", out[f"{tag}_using_backup"].astype(int))
    # ---- the example's closed loop (examples/evade/test_evade.py:425-500) -------------------------------------------------
    cfg, env, spec, nominal, backup, dyn, sh = build()
    state = np.array([cfg.simulation.initial_x, 0.0, 0.0, 0.0]).reshape(-1, 1)
    T = int(cfg.simulation.tf / cfg.simulation.dt)
    Xs, Us, Bs, UB, HM = [], [], [], [], []
    outcome = 0
    for step in range(T):
        pos = state[:2, 0].copy()
        u_nom = nominal.compute_control(state).flatten()
        sh.set_nominal_trajectory(None, np.tile(u_nom, (3, 1)))          # only nominal_u_traj[0] is read (:179-181)
        Xs.append(state.flatten().copy()); Bs.append(env.bullet_x)
        control = sh.solve_control_problem(state)
        Us.append(control.flatten().copy()); UB.append(sh.is_using_backup()); HM.append(sh._last_h_min)
        state = dyn.step(state, control)
        vx, vy = state[2, 0], state[3, 0]
        vm = np.sqrt(vx ** 2 + vy ** 2)
        if vm > cfg.robot.v_max:
            state[2, 0] = vx * cfg.robot.v_max / vm
            state[3, 0] = vy * cfg.robot.v_max / vm
        env.step_bullet(cfg.simulation.dt)
        if env.check_obstacle_collision(pos, cfg.robot.radius)[0]:
            outcome = -2
            break
        if env.check_goal_reached(pos):
            outcome = 1
            break
    out["loop_X"], out["loop_U"], out["loop_bullet_x"] = np.array(Xs), np.array(Us), np.array(Bs)
    out["loop_using_backup"], out["loop_h_min"], out["loop_outcome"] = np.array(UB), np.array(HM), outcome
    out["loop_final_state"] = state.flatten()
    print("closed loop:", len(Xs), "steps, outcome", outcome, "backup steps", int(np.sum(UB)), "min h", float(np.min(HM)))
    # environment / robot constants the oracle restates (checked by the CPU tests)
    out["env"] = np.array([env.hallway_length, env.half_width, env.pocket_x_min, env.pocket_x_max, env.pocket_y_min, env.pocket_y_max,
                           env.goal_x_min, env.goal_x_max, env.bullet_speed, env.bullet_length, env.bullet_width, env.bullet_start_x])
    out["spec"] = np.array([spec["radius"], spec["a_max"], spec["v_max"], spec["safety_margin"], sh.alpha, sh.alpha_terminal])
    np.savez_compressed(os.path.join(HERE, "backup_cbf.npz"), **out)
    print("wrote backup_cbf.npz", os.path.getsize(os.path.join(HERE, "backup_cbf.npz")), "bytes")


if __name__ == "__main__":
    gen()
