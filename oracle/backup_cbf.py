"""Backup-CBF QP: float64 numpy restatement of the reference's shielding controller on its evade scenario.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned on tests/golden/backup_cbf.npz, which
tests/golden/make_golden_backup.py produced by running the reference's own code (rollout, sensitivities, rows, QP
statement) -- only the QP minimiser comes from this repo's exact solver (OSQP is not installable; unique minimiser).

Follows, function by function:
  BackupCBF.solve_control_problem        position_control/backup_cbf_qp.py:563-794
  BackupCBF._integrate_backup_trajectory :236-320   (robot.step + forward differences, eps = 1e-5)
  BackupCBF._h_safety / _grad_h_safety   :343-447 / :449-461   (evade branch :359-392, rectangular moving obstacle :419-430)
  BackupCBF._h_terminal / _grad          :463-547 / :549-561   (pocket bounds :481-494, speed :524-535, safety at T :537-541)
  EvadeBackupController.compute_control  position_control/backup_controller.py:456-571
  DoubleIntegrator2D.f / g / step        robots/double_integrator2D.py:46-107
  EvadeEnv geometry, get_bullet_state    envs/evade_env.py:30-83, 386-406
  EvadeNominalController, moving-obstacle prediction, closed loop   examples/evade/test_evade.py:128-166, 373-384, 425-470
"""
import numpy as np

from . import qp as oqp

FD_EPS = 1e-5                                               # backup_cbf_qp.py:282, :451, :551


def default_env(**over):
    """EvadeEnv(...) as examples/evade/test_evade.py:60-72,278-288 constructs it (envs/evade_env.py:51-83)."""
    e = dict(hallway_length=60.0, hallway_width=4.0, pocket_x=25.0, pocket_length=10.0, pocket_width=4.0, goal_length=5.0,
             bullet_speed=3.0, bullet_length=3.0, bullet_start_x=-10.0)
    e.update(over)
    e["bullet_width"] = e.get("bullet_width") or e["hallway_width"]
    e["half_width"] = e["hallway_width"] / 2
    e["pocket_x_min"], e["pocket_x_max"] = e["pocket_x"], e["pocket_x"] + e["pocket_length"]
    e["pocket_y_min"], e["pocket_y_max"] = e["half_width"], e["half_width"] + e["pocket_width"]
    e["pocket_cx"] = (e["pocket_x_min"] + e["pocket_x_max"]) / 2
    e["pocket_cy"] = (e["pocket_y_min"] + e["pocket_y_max"]) / 2
    e["goal_x_min"], e["goal_x_max"] = e["hallway_length"] - e["goal_length"], e["hallway_length"]
    return e


def default_spec(**over):
    """examples/evade/test_evade.py:75-88,298-299; gains backup_cbf_qp.py:93-94."""
    s = dict(radius=0.5, a_max=2.0, v_max=1.5, safety_margin=0.5, alpha=1.0, alpha_terminal=2.0)
    s.update(over)
    return s


def di_step(x, u, dt, v_max):
    """DoubleIntegrator2D.step (double_integrator2D.py:79-107): Euler, then the speed rescaled to v_max."""
    xn = np.array([x[0] + x[2] * dt, x[1] + x[3] * dt, x[2] + u[0] * dt, x[3] + u[1] * dt])
    vm = np.sqrt(xn[2] ** 2 + xn[3] ** 2)
    if vm > v_max:
        s = v_max / vm
        xn[2] *= s
        xn[3] *= s
    return xn


def _clamp(ax, ay, a_max):
    am = np.sqrt(ax ** 2 + ay ** 2)
    if am > a_max:
        ax, ay = ax * a_max / am, ay * a_max / am
    return np.array([ax, ay])


def nominal_control(x, spec):
    """EvadeNominalController.compute_control (examples/evade/test_evade.py:141-166)."""
    ax = 2.0 * (spec["v_max"] - x[2])
    ay = 2.0 * (0.0 - x[1]) + 2.0 * (0.0 - x[3])
    return _clamp(ax, ay, spec["a_max"])


def backup_control(x, env, spec):
    """EvadeBackupController.compute_control (backup_controller.py:456-571), Kp = Kd = 2."""
    Kp, Kd = 2.0, 2.0
    px, py, vx, vy = x
    a_max = spec["a_max"]
    if env["goal_x_min"] <= px <= env["goal_x_max"] and -env["half_width"] <= py <= env["half_width"]:
        return _clamp(-Kd * vx, -Kd * vy, a_max)
    x_min, x_max, y_min, y_max = env["pocket_x_min"], env["pocket_x_max"], env["pocket_y_min"], env["pocket_y_max"]
    cx, cy = env["pocket_cx"], env["pocket_cy"]
    margin = spec["radius"] + 0.1
    dist = np.sqrt((px - cx) ** 2 + (py - cy) ** 2)
    if x_min + margin <= px <= x_max - margin and y_min + margin <= py <= y_max - margin and dist < 1.0:
        return _clamp(-Kd * vx, -Kd * vy, a_max)
    if x_min - 2.0 <= px <= x_max + 2.0:
        if x_min + margin <= px <= x_max - margin:
            ax = Kp * (cx - px) - Kd * vx
            ay = Kp * (cy - py) - Kd * vy
        else:
            ty = max(py, 3.0) if py > y_min else 0.0
            ax = Kp * (cx - px) - Kd * vx
            ay = Kp * (ty - py) - Kd * vy
    else:
        ty = max(py, 3.0) if (py > y_min and px > x_max) else 0.0
        ex, ey = cx - px, ty - py
        ax = Kp * np.sign(ex) * min(abs(ex), 3.0) - Kd * vx
        ay = Kp * ey - Kd * vy
    return _clamp(ax, ay, a_max)


def h_safety(x, t, bullet_x, env, spec):
    """BackupCBF._h_safety on the evade environment with the bullet predicted at constant speed."""
    px, py = x[0], x[1]
    R = spec["radius"]
    h = py + env["half_width"] - R
    h = min(h, px - R)
    h = min(h, env["hallway_length"] - px - R)
    if env["pocket_x_min"] <= px <= env["pocket_x_max"]:
        h = min(h, env["pocket_y_max"] - py - R)
        if py > env["half_width"]:
            h = min(h, px - env["pocket_x_min"] - R, env["pocket_x_max"] - px - R)
    else:
        h = min(h, env["half_width"] - py - R)
    # get_bullet_state (evade_env.py:386-406) + test_evade.py:373-384: centre shifted by L/6, length 4L/3, moves with vx t
    ox = bullet_x + env["bullet_length"] / 6 + env["bullet_speed"] * t
    dx = max(abs(px - ox) - env["bullet_length"] * (1 + 1 / 3) / 2, 0)
    dy = max(abs(py - 0.0) - env["bullet_width"] / 2, 0)
    h = min(h, np.sqrt(dx ** 2 + dy ** 2) - R - spec["safety_margin"])
    return h


def h_terminal(x, bullet_x, env, spec, horizon):
    """BackupCBF._h_terminal: pocket box with margin R + 0.2, speed below v_max, and the safety value at t = horizon."""
    m = spec["radius"] + 0.2
    h = min(x[0] - env["pocket_x_min"] - m, env["pocket_x_max"] - x[0] - m, x[1] - env["pocket_y_min"] - m, env["pocket_y_max"] - x[1] - m)
    h = min(h, spec["v_max"] - np.sqrt(x[2] ** 2 + x[3] ** 2))
    return min(h, h_safety(x, horizon, bullet_x, env, spec))


def _fd_grad(fun, x):
    h0 = fun(x)
    g = np.zeros(4)
    for i in range(4):
        xp = x.copy()
        xp[i] += FD_EPS
        g[i] = (fun(xp) - h0) / FD_EPS
    return g


def rollout(x0, N, dt, env, spec):
    """_integrate_backup_trajectory: phi [N,4], S [N,4,4] (S_0 = I, S_{i} = A_{i-1} S_{i-1}, A by forward differences of
    step(x, backup(x)))."""
    phi = np.zeros((N, 4))
    S = np.zeros((N, 4, 4))
    x = np.asarray(x0, dtype=float).copy()
    Sc = np.eye(4)
    phi[0], S[0] = x, Sc
    for i in range(1, N):
        xn = di_step(x, backup_control(x, env, spec), dt, spec["v_max"])
        A = np.zeros((4, 4))
        for j in range(4):
            xp = x.copy()
            xp[j] += FD_EPS
            A[:, j] = (di_step(xp, backup_control(xp, env, spec), dt, spec["v_max"]) - xn) / FD_EPS
        Sc = A @ Sc
        x = xn
        phi[i], S[i] = x, Sc
    return phi, S


def assemble_rows(x0, phi, S, bullet_x, dt, horizon, env, spec):
    """Rows ``lhs . u >= rhs`` in PHYSICAL inputs (backup_cbf_qp.py:620-676) with the keep mask (|lhs| > 1e-6), plus
    min(h along the rollout, h_terminal) (:577-581).  Row N-1 is the terminal row."""
    N = phi.shape[0]
    f0 = np.array([x0[2], x0[3], 0.0, 0.0])
    g0 = np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    lhs = np.zeros((N, 2)); rhs = np.zeros(N); keep = np.zeros(N, dtype=bool)
    hv = [h_safety(phi[i], i * dt, bullet_x, env, spec) for i in range(N)]
    hT = h_terminal(phi[-1], bullet_x, env, spec, horizon)
    for i in range(1, N):
        t = i * dt
        h = h_safety(phi[i], t, bullet_x, env, spec)
        g = _fd_grad(lambda z: h_safety(z, t, bullet_x, env, spec), phi[i])
        dh_dt = (h_safety(phi[i], t + dt, bullet_x, env, spec) - h) / dt
        fpi = (phi[i + 1] - phi[i]) / dt if i < N - 1 else (phi[i] - phi[i - 1]) / dt
        gS = g @ S[i]
        lhs[i - 1] = gS @ g0
        rhs[i - 1] = -(gS @ f0) + (g @ fpi) - dh_dt - spec["alpha"] * h
        keep[i - 1] = np.linalg.norm(lhs[i - 1]) > 1e-6
    gT = _fd_grad(lambda z: h_terminal(z, bullet_x, env, spec, horizon), phi[-1])
    gS = gT @ S[-1]
    lhs[N - 1] = gS @ g0
    rhs[N - 1] = -(gS @ f0 + spec["alpha_terminal"] * hT)
    keep[N - 1] = np.linalg.norm(lhs[N - 1]) > 1e-6
    return lhs, rhs, keep, min(min(hv), hT)


def solve(x0, u_nom, bullet_x, dt=0.1, horizon=12.0, env=None, spec=None, return_info=False):
    """BackupCBF.solve_control_problem -> u_safe (2,), and with return_info the pieces the fixtures hold."""
    env = env or default_env()
    spec = spec or default_spec()
    x0 = np.asarray(x0, dtype=float).reshape(4)
    N = int(horizon / dt)
    phi, S = rollout(x0, N, dt, env, spec)
    lhs, rhs, keep, h_min = assemble_rows(x0, phi, S, bullet_x, dt, horizon, env, spec)
    u_ref = np.asarray(u_nom, dtype=float).reshape(2)
    scale = np.array([spec["a_max"], spec["a_max"]])
    G, h = lhs[keep], rhs[keep]
    info = dict(phi=phi, S=S, rows=np.column_stack([G * scale[None, :], h]), n_rows=int(keep.sum()), h_min=h_min, qp_status=-1,
                using_backup=False)
    if G.shape[0] == 0:
        u = u_ref                                           # :770-774: no rows -> the (unclipped) reference
    else:
        u_ref = np.clip(u_ref, -scale, scale)
        us_ref = u_ref / scale
        if np.any(np.isnan(G)) or np.any(np.isnan(h)):
            us, st = None, oqp.STATUS_INFEASIBLE
        else:
            A = np.vstack([G * scale[None, :], np.eye(2), -np.eye(2)])
            c = np.concatenate([-h, np.ones(2), np.ones(2)])
            us, st = oqp.solve_qpn(A, c, us_ref)
        info["qp_status"] = int(st)
        if st == oqp.STATUS_OPTIMAL:
            u = scale * us
            info["using_backup"] = bool(np.linalg.norm(us - us_ref) > 0.1)      # Q_u = [1, 1] (:104)
        elif h_min > 0.01:
            u = u_ref
        else:
            u = backup_control(x0, env, spec)
            info["using_backup"] = True
    return (u, info) if return_info else u


def closed_loop(x0, bullet_x0, steps, dt=0.1, horizon=12.0, env=None, spec=None):
    """The example's loop (test_evade.py:425-500 without the figure): X[T,4], U[T,2], bullet_x[T], using_backup[T],
    h_min[T], outcome (1 goal, -2 collision, 0 still running)."""
    env = env or default_env()
    spec = spec or default_spec()
    x = np.asarray(x0, dtype=float).copy()
    bx = float(bullet_x0)
    Xs, Us, Bs, UB, HM = [], [], [], [], []
    outcome = 0
    for _ in range(steps):
        pos = x[:2].copy()
        u, info = solve(x, nominal_control(x, spec), bx, dt, horizon, env, spec, return_info=True)
        Xs.append(x.copy()); Us.append(u.copy()); Bs.append(bx); UB.append(info["using_backup"]); HM.append(info["h_min"])
        x = di_step(x, u, dt, spec["v_max"])
        vm = np.sqrt(x[2] ** 2 + x[3] ** 2)
        if vm > spec["v_max"]:
            x[2:] *= spec["v_max"] / vm
        bx += env["bullet_speed"] * dt                      # EvadeEnv.step_bullet (evade_env.py:360-384): respawn past the hallway
        if bx > env["hallway_length"] + env["bullet_length"]:
            bx = env["bullet_start_x"]
        if bullet_hits(pos, bx, env, spec["radius"]):       # the example checks the pre-step position against the stepped bullet

            outcome = -2
            break
        if env["goal_x_min"] <= pos[0] <= env["goal_x_max"] and -env["half_width"] <= pos[1] <= env["half_width"]:
            outcome = 1
            break
    return np.array(Xs), np.array(Us), np.array(Bs), np.array(UB), np.array(HM), outcome, x


def bullet_hits(pos, bullet_x, env, radius):
    """EvadeEnv.check_obstacle_collision (evade_env.py:454-485): circle against the bullet's box, nose included."""
    L = env["bullet_length"]
    cx = min(max(pos[0], bullet_x - L / 2), bullet_x + L / 2 + L / 3)
    cy = min(max(pos[1], -env["bullet_width"] / 2), env["bullet_width"] / 2)
    return np.sqrt((pos[0] - cx) ** 2 + (pos[1] - cy) ** 2) < radius
