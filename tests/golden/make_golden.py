#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is produced by the *reference's own code* (imported through
tests/golden/_ref_import.py): dynamics f/g/step, nominal_input/stop/rotate_to,
agent_barrier outputs, the CBF rows A1/b1 written by
CBFQP.solve_control_problem (position_control/cbf_qp.py:108-187, executed
verbatim), get_nearest_unpassed_obs, and closed-loop control_step
trajectories of LocalTrackingController / LocalTrackingControllerDyn.

What is NOT produced by the reference: the QP minimiser u*.  cvxpy/GUROBI are
not installable here, so ``Problem.solve`` is replaced by the oracle's exact
active-set enumerator (oracle/qp.py).  u* is pinned by uniqueness of the
minimiser of a strictly convex QP, not by a reference solver run; the fixture
field names say so (``u_star_oracle``).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402

_ref_import.install()

from oracle import qp as oqp  # noqa: E402

from safe_control.utils.headless_plot import NullAxes  # noqa: E402
from safe_control.robots.robot import BaseRobot  # noqa: E402
from safe_control.position_control.cbf_qp import CBFQP  # noqa: E402
from safe_control.tracking import LocalTrackingController  # noqa: E402
from safe_control.utils import env as ref_env  # noqa: E402

DT = 0.05


class OracleProblem:
    """Stands where cvxpy.Problem would: solves with the oracle enumerator."""

    def __init__(self, ctrl, lo, hi):
        self.ctrl, self.lo, self.hi = ctrl, lo, hi
        self.status = "optimal"

    def solve(self, **_):
        c = self.ctrl
        Gb, cb = oqp.box_rows(self.lo, self.hi)
        u, st = oqp.solve_qp2(np.vstack([c.A1.value, Gb]),
                              np.concatenate([c.b1.value.reshape(-1), cb]),
                              np.asarray(c.u_ref.value, dtype=float).reshape(-1))
        self.status = "optimal" if st == 0 else "infeasible"
        c.u.value = None if u is None else u.reshape(2, 1)


def bounds_for(spec):
    if spec["model"] == "DynamicUnicycle2D":
        hi = np.array([spec["a_max"], spec["w_max"]])
    else:
        hi = np.array([spec["a_max"], spec["beta_max"]])
    return -hi, hi


def make_robot(model_name, extra=None):
    spec = {"model": model_name}
    if model_name == "DynamicUnicycle2D":
        spec.update(a_max=1.0, w_max=0.5, radius=0.25)
    else:
        spec.update(a_max=5.0, radius=0.3)
    if extra:
        spec.update(extra)
    robot = BaseRobot(np.zeros((4, 1)), spec, DT, NullAxes())
    return robot, spec


def draw_state(rng, model_name):
    x, y = rng.uniform(0, 14, 2)
    th = rng.uniform(-np.pi, np.pi)
    v = rng.uniform(0, 1) if model_name == "DynamicUnicycle2D" else rng.uniform(0.2, 3.5)
    return np.array([x, y, th, v])


def draw_circles(rng, X, K, R, moving=False, rho_max=4.0, overlap=False):
    obs = np.zeros((K, 7))
    for k in range(K):
        r = rng.uniform(0.2, 1.0)
        lo = 0.7 * (r + R) if overlap else r + R + 0.05
        rho = rng.uniform(lo, rho_max)
        phi = rng.uniform(-np.pi, np.pi)
        obs[k, 0:3] = [X[0] + rho * np.cos(phi), X[1] + rho * np.sin(phi), r]
        if moving:
            obs[k, 3:5] = rng.uniform(-0.5, 0.5, 2)
    return obs


def draw_superellipsoids(rng, X, K, R):
    obs = np.zeros((K, 7))
    for k in range(K):
        a, b = rng.uniform(0.3, 1.2, 2)
        e = float(rng.choice([4, 6, 10]))
        tho = rng.uniform(-np.pi, np.pi)
        rho = rng.uniform(max(a, b) + R + 0.3, 4.5)
        phi = rng.uniform(-np.pi, np.pi)
        obs[k] = [X[0] + rho * np.cos(phi), X[1] + rho * np.sin(phi), a, b, e, tho, 1.0]
    return obs


# ---------------------------------------------------------------------------
def gen_callbacks(rng):
    out = {}
    for name in ["DynamicUnicycle2D", "KinematicBicycle2D", "KinematicBicycle2D_C3BF",
                 "KinematicBicycle2D_DPCBF"]:
        robot, spec = make_robot(name)
        N = 96
        rec = {k: [] for k in ["X", "U", "goal", "theta_des", "obs", "f", "g", "step",
                               "nominal", "stop", "rotate", "h", "g1", "g2"]}
        for i in range(N):
            X = draw_state(rng, name)
            robot.X = X.reshape(-1, 1).copy()
            U = rng.uniform(-1, 1, 2) * np.array([spec["a_max"], 0.5])
            goal = rng.uniform(0, 14, 2)
            th_des = rng.uniform(-np.pi, np.pi)
            if name == "DynamicUnicycle2D" and i % 3 == 2:
                obs = draw_superellipsoids(rng, X, 1, spec["radius"])[0]
            else:
                obs = draw_circles(rng, X, 1, spec["radius"], moving=("_" in name))[0]
            rec["X"].append(X); rec["U"].append(U); rec["goal"].append(goal)
            rec["theta_des"].append(th_des); rec["obs"].append(obs)
            rec["f"].append(robot.f().reshape(-1))
            rec["g"].append(np.asarray(robot.g(), dtype=float))
            rec["nominal"].append(robot.nominal_input(goal).reshape(-1))
            rec["stop"].append(np.asarray(robot.stop(), dtype=float).reshape(-1))
            rec["rotate"].append(robot.rotate_to(th_des).reshape(-1))
            bar = robot.agent_barrier(obs)
            rec["h"].append(float(np.asarray(bar[0]).reshape(-1)[0]))
            if len(bar) == 3:
                rec["g1"].append(float(np.asarray(bar[1]).reshape(-1)[0]))
                rec["g2"].append(np.asarray(bar[2], dtype=float).reshape(-1))
            else:
                rec["g1"].append(np.nan)
                rec["g2"].append(np.asarray(bar[1], dtype=float).reshape(-1))
            Xn = robot.step(U.reshape(-1, 1)).reshape(-1).copy()
            rec["step"].append(Xn)
        for k, v in rec.items():
            out[f"{name}/{k}"] = np.array(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "callbacks.npz"), **out)
    print("callbacks.npz", len(out), "arrays")


# ---------------------------------------------------------------------------
def run_cbfqp_case(ctrl, robot, X, u_ref, obs_list):
    robot.X = X.reshape(-1, 1).copy()
    control_ref = {"state_machine": "track", "u_ref": u_ref.reshape(2, 1), "goal": None}
    u = ctrl.solve_control_problem(robot.X, control_ref, obs_list)
    return (ctrl.A1.value.copy(), ctrl.b1.value.reshape(-1).copy(),
            None if u is None else np.asarray(u, dtype=float).reshape(-1), ctrl.status)


def gen_cbfqp(rng):
    out = {}
    KMAX = 12
    groups = [
        # name, model, num_obs, n_cases, obstacle kind, spec extras
        ("du_circle", "DynamicUnicycle2D", 8, 160, "circle", None),
        ("du_circle_hard", "DynamicUnicycle2D", 8, 48, "circle", {"cbf_mode": "hard"}),
        ("du_superellipsoid", "DynamicUnicycle2D", 8, 96, "super", None),
        ("du_mixed_trunc", "DynamicUnicycle2D", 5, 64, "mixed", None),      # K > num_obs and K < num_obs
        ("du_overlap", "DynamicUnicycle2D", 8, 96, "overlap", None),        # many infeasible
        ("kb_circle", "KinematicBicycle2D", 8, 64, "circle", None),
        ("c3bf", "KinematicBicycle2D_C3BF", 10, 96, "moving", None),
        ("c3bf_k16", "KinematicBicycle2D_C3BF", 16, 48, "moving16", None),
        ("dpcbf", "KinematicBicycle2D_DPCBF", 10, 96, "moving", None),
    ]
    for gname, model, num_obs, n, kind, extra in groups:
        robot, spec = make_robot(model, extra)
        ctrl = CBFQP(robot, spec, num_obs=num_obs)
        lo, hi = bounds_for(spec)
        ctrl.cbf_controller = OracleProblem(ctrl, lo, hi)
        rec = {k: [] for k in ["X", "u_ref", "obs", "k", "A", "b", "u_star_oracle", "status_oracle"]}
        kmax = 16 if kind == "moving16" else KMAX
        for i in range(n):
            X = draw_state(rng, model)
            R = spec["radius"]
            if kind == "circle":
                K = 8; obs = draw_circles(rng, X, K, R)
            elif kind == "super":
                K = int(rng.integers(1, 9)); obs = draw_superellipsoids(rng, X, K, R)
            elif kind == "mixed":
                K = int(rng.integers(1, 10))
                obs = np.vstack([draw_circles(rng, X, K, R)[: (K + 1) // 2],
                                 draw_superellipsoids(rng, X, K, R)[: K // 2]]) if K > 1 else draw_circles(rng, X, 1, R)
                K = obs.shape[0]
            elif kind == "overlap":
                K = 8; obs = draw_circles(rng, X, K, R, rho_max=2.0, overlap=True)
            elif kind == "moving":
                K = int(rng.integers(1, 11)); obs = draw_circles(rng, X, K, R, moving=True, rho_max=6.0)
                obs[:, 5:7] = rng.uniform(-5, 5, (K, 2))      # y_min/y_max slots are ignored by C3BF/DPCBF
            elif kind == "moving16":
                K = 16; obs = draw_circles(rng, X, K, R, moving=True, rho_max=8.0)
            goal = rng.uniform(0, 14, 2)
            robot.X = X.reshape(-1, 1).copy()
            u_ref = robot.nominal_input(goal).reshape(-1)
            if i % 7 == 3:
                u_ref = u_ref * 4.0                               # push u_ref outside the box
            A, b, u, status = run_cbfqp_case(ctrl, robot, X, u_ref, list(obs))
            obs_p = np.full((kmax, 7), np.nan); obs_p[:K] = obs
            rec["X"].append(X); rec["u_ref"].append(u_ref); rec["obs"].append(obs_p); rec["k"].append(K)
            rec["A"].append(A); rec["b"].append(b)
            rec["u_star_oracle"].append(np.full(2, np.nan) if u is None else u)
            rec["status_oracle"].append(0 if status == "optimal" else 1)
        for k, v in rec.items():
            out[f"{gname}/{k}"] = np.array(v)
        out[f"{gname}/meta"] = np.array([num_obs, spec["radius"], lo[0], lo[1], hi[0], hi[1],
                                         spec.get("rear_ax_dist", 0.0)], dtype=np.float64)
        n_inf = int(np.sum(rec["status_oracle"]))
        print(f"{gname}: {n} cases, {n_inf} infeasible")
    # obs_list None: returns u_ref unclipped, status optimal (cbf_qp.py:113-118)
    robot, spec = make_robot("DynamicUnicycle2D")
    ctrl = CBFQP(robot, spec, num_obs=8)
    u_ref = np.array([3.0, -2.0])
    robot.X = np.array([1.0, 2.0, 0.3, 0.5]).reshape(-1, 1)
    u = ctrl.solve_control_problem(robot.X, {"u_ref": u_ref.reshape(2, 1)}, None)
    out["none/u_ref"] = u_ref
    out["none/u"] = np.asarray(u, dtype=float).reshape(-1)
    out["none/status_optimal"] = np.array([ctrl.status == "optimal"])
    np.savez_compressed(os.path.join(HERE, "cbfqp_cases.npz"), **out)


# ---------------------------------------------------------------------------
def gen_nearest(rng):
    out = {}
    for name in ["DynamicUnicycle2D", "KinematicBicycle2D_C3BF"]:
        spec = {"model": name, "radius": 0.25}
        if name == "DynamicUnicycle2D":
            spec.update(a_max=1.0, w_max=0.5)
        ctl = LocalTrackingController(np.array([1.0, 1.0, 0.0, 0.5]), spec,
                                      controller_type={"pos": "cbf_qp"}, dt=DT, env=ref_env.Env())
        Xs, tables, counts, sel, nsel = [], [], [], [], []
        for i in range(64):
            M = int(rng.integers(1, 25))
            table = np.zeros((24, 7)); table[:] = np.nan
            obs = np.zeros((M, 7))
            obs[:, 0:2] = rng.uniform(0, 14, (M, 2)); obs[:, 2] = rng.uniform(0.2, 1.0, M)
            table[:M] = obs
            X = draw_state(rng, name)
            ctl.robot.X = X.reshape(-1, 1).copy(); ctl.robot.yaw = X[2]
            ctl.obs = obs.copy()
            got = ctl.get_nearest_unpassed_obs([], obs_num=10)
            s = np.full((10, 7), np.nan); s[: len(got)] = got
            Xs.append(X); tables.append(table); counts.append(M); sel.append(s); nsel.append(len(got))
        out[f"{name}/X"] = np.array(Xs); out[f"{name}/table"] = np.array(tables)
        out[f"{name}/m"] = np.array(counts); out[f"{name}/sel"] = np.array(sel); out[f"{name}/nsel"] = np.array(nsel)
    np.savez_compressed(os.path.join(HERE, "nearest_obs.npz"), **out)
    print("nearest_obs.npz")


# ---------------------------------------------------------------------------
def gen_closed_loop():
    """BASELINE config 1: examples/test_tracking.py --model du --algo cbf_qp (no sensor, headless)."""
    out = {}
    known = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0],
                      [7.0, 7.0, 3.0], [4.0, 3.5, 1.5], [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6],
                      [11.0, 5.0, 0.8], [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
    known = np.hstack((known, np.zeros((known.shape[0], 4))))
    wps = np.array([[2, 2, np.pi / 2], [2, 12, 0], [12, 12, 0], [12, 2, 0]], dtype=np.float64)
    for tag, table in [("du14", known), ("du3", known[:3])]:
        spec = {"model": "DynamicUnicycle2D", "w_max": 0.5, "a_max": 1.0, "radius": 0.25}
        ctl = LocalTrackingController(np.append(wps[0], 1.0), spec, controller_type={"pos": "cbf_qp"},
                                      dt=DT, env=ref_env.Env())
        lo, hi = bounds_for(spec)
        ctl.pos_controller.cbf_controller = OracleProblem(ctl.pos_controller, lo, hi)
        ctl.obs = table.copy()
        ctl.set_waypoints(wps)
        Xs, Us, rets, sms = [ctl.robot.X.reshape(-1).copy()], [], [], []
        for _ in range(2000):
            ret = ctl.control_step()
            rets.append(ret); sms.append(["idle", "track", "stop", "rotate"].index(ctl.state_machine))
            if ret == -2:
                break
            Xs.append(ctl.robot.X.reshape(-1).copy()); Us.append(ctl.get_control_input().reshape(-1).copy())
            if ret == -1:
                break
        out[f"{tag}/obs"] = table; out[f"{tag}/waypoints"] = wps
        out[f"{tag}/X"] = np.array(Xs); out[f"{tag}/U"] = np.array(Us)
        out[f"{tag}/ret"] = np.array(rets); out[f"{tag}/sm"] = np.array(sms)
        print(tag, "steps", len(rets), "last ret", rets[-1])

    # moving obstacles: dynamic_env/main.py LocalTrackingControllerDyn with C3BF
    from safe_control.dynamic_env.main import LocalTrackingControllerDyn
    rng = np.random.default_rng(7)
    obs = np.zeros((8, 7))
    obs[:, 0] = rng.uniform(6, 20, 8); obs[:, 1] = rng.uniform(1, 9, 8); obs[:, 2] = 0.4
    obs[:, 3] = -0.5; obs[:, 4] = rng.choice([-0.5, 0.5], 8); obs[:, 5] = 0.0; obs[:, 6] = 10.0
    wps = np.array([[1, 5, 0], [22, 5, 0]], dtype=np.float64)
    for tag, model in [("c3bf_dyn", "KinematicBicycle2D_C3BF"), ("dpcbf_dyn", "KinematicBicycle2D_DPCBF")]:
        spec = {"model": model, "a_max": 5.0, "radius": 0.3}
        ctl = LocalTrackingControllerDyn(np.append(wps[0], 1.0), spec, controller_type={"pos": "cbf_qp"},
                                         dt=DT, env=ref_env.Env())
        lo, hi = bounds_for(spec)
        ctl.pos_controller.cbf_controller = OracleProblem(ctl.pos_controller, lo, hi)
        ctl.obs = obs.copy()
        # cone / parabola drawing (dynamic_env/robot.py) is rendering only: skip it
        ctl.robot.draw_collision_cone = lambda *a, **k: None
        ctl.robot.draw_collision_parabola = lambda *a, **k: None
        ctl.set_waypoints(wps)
        Xs, Us, rets = [ctl.robot.X.reshape(-1).copy()], [], []
        for _ in range(600):
            ret = ctl.control_step()
            rets.append(ret)
            if ret == -2:
                break
            Xs.append(ctl.robot.X.reshape(-1).copy()); Us.append(ctl.get_control_input().reshape(-1).copy())
            if ret == -1:
                break
        out[f"{tag}/obs0"] = obs; out[f"{tag}/waypoints"] = wps
        out[f"{tag}/X"] = np.array(Xs); out[f"{tag}/U"] = np.array(Us); out[f"{tag}/ret"] = np.array(rets)
        out[f"{tag}/obs_final"] = ctl.obs.copy()
        print(tag, "steps", len(rets), "last ret", rets[-1])
    np.savez_compressed(os.path.join(HERE, "closed_loop.npz"), **out)


# ---------------------------------------------------------------------------
def gen_integrators():
    """SingleIntegrator2D / DoubleIntegrator2D callbacks and CBF-QP rows, own seed and file so
    the earlier fixtures stay byte-identical."""
    rng = np.random.default_rng(77)
    out = {}
    for name, nx in [("SingleIntegrator2D", 2), ("DoubleIntegrator2D", 4)]:
        spec = {"model": name, "radius": 0.25}
        if name == "SingleIntegrator2D":
            spec.update(v_max=1.0)
            X0 = np.zeros((3, 1))
        else:
            spec.update(v_max=1.0, a_max=1.5)
            X0 = np.zeros((5, 1))
        robot = BaseRobot(X0, spec, DT, NullAxes())
        ctrl = CBFQP(robot, spec, num_obs=6)
        hi = np.array([spec["v_max"]] * 2) if name == "SingleIntegrator2D" else np.array([spec["a_max"]] * 2)
        ctrl.cbf_controller = OracleProblem(ctrl, -hi, hi)
        rec = {k: [] for k in ["X", "goal", "U", "u_ref", "obs", "k", "f", "g", "step", "nominal", "stop", "A", "b",
                               "u_star_oracle", "status_oracle"]}
        for i in range(120):
            x, y = rng.uniform(0, 14, 2)
            X = np.array([x, y]) if nx == 2 else np.array([x, y, *rng.uniform(-0.8, 0.8, 2)])
            goal = rng.uniform(0, 14, 2)
            U = rng.uniform(-1, 1, 2)
            K = int(rng.integers(1, 7))
            Xp = np.array([X[0], X[1], 0.0, 0.0])
            if i % 3 == 2:
                obs = draw_superellipsoids(rng, Xp, K, spec["radius"])
            else:
                obs = draw_circles(rng, Xp, K, spec["radius"], overlap=(i % 5 == 0), rho_max=3.0)
            robot.X = X.reshape(-1, 1).copy()
            rec["f"].append(np.asarray(robot.f(), dtype=float).reshape(-1))
            rec["g"].append(np.asarray(robot.g(), dtype=float))
            u_ref = robot.nominal_input(goal).reshape(-1)
            if i % 7 == 3:
                u_ref = u_ref * 4.0
            rec["nominal"].append(robot.nominal_input(goal).reshape(-1))
            rec["stop"].append(np.asarray(robot.stop(), dtype=float).reshape(-1))
            A, b, u, status = run_cbfqp_case(ctrl, robot, X, u_ref, list(obs))
            robot.X = X.reshape(-1, 1).copy()
            Xn = robot.robot.step(robot.X.copy(), U.reshape(-1, 1)).reshape(-1).copy()
            obs_p = np.full((6, 7), np.nan); obs_p[:K] = obs
            Xpad = np.zeros(4); Xpad[:nx] = X
            Xnpad = np.zeros(4); Xnpad[:nx] = Xn
            rec["X"].append(Xpad); rec["goal"].append(goal); rec["U"].append(U); rec["u_ref"].append(u_ref)
            rec["obs"].append(obs_p); rec["k"].append(K); rec["step"].append(Xnpad)
            rec["A"].append(A); rec["b"].append(b)
            rec["u_star_oracle"].append(np.full(2, np.nan) if u is None else u)
            rec["status_oracle"].append(0 if status == "optimal" else 1)
        for k, v in rec.items():
            out[f"{name}/{k}"] = np.array(v)
        print(name, "cases", len(rec["k"]), "infeasible", int(np.sum(rec["status_oracle"])))
    np.savez_compressed(os.path.join(HERE, "integrators.npz"), **out)


def gen_quad2d():
    """Quad2D (6 states, thrust inputs with an asymmetric box): f, g, step, agent_barrier, CBF rows."""
    rng = np.random.default_rng(99)
    out = {}
    spec = {"model": "Quad2D", "f_min": 3.0, "f_max": 10.0, "radius": 0.25}      # examples/test_tracking.py:112-118
    robot = BaseRobot(np.zeros((6, 1)), spec, DT, NullAxes())
    ctrl = CBFQP(robot, spec, num_obs=6)
    ctrl.cbf_controller = OracleProblem(ctrl, np.array([spec["f_min"]] * 2), np.array([spec["f_max"]] * 2))
    rec = {k: [] for k in ["X", "U", "u_ref", "obs", "k", "f", "g", "step", "h", "hdot", "dhd", "A", "b",
                           "u_star_oracle", "status_oracle"]}
    for i in range(140):
        X = np.array([*rng.uniform(0, 14, 2), rng.uniform(-0.6, 0.6), *rng.uniform(-1.5, 1.5, 2), rng.uniform(-1, 1)])
        U = rng.uniform(3.0, 10.0, 2)
        K = int(rng.integers(1, 7))
        obs = draw_circles(rng, np.array([X[0], X[1], 0, 0]), K, spec["radius"], overlap=(i % 6 == 0), rho_max=3.5)
        u_ref = rng.uniform(2.0, 11.0, 2)                                       # some outside [f_min, f_max]
        robot.X = X.reshape(-1, 1).copy()
        rec["f"].append(np.asarray(robot.f(), dtype=float).reshape(-1))
        rec["g"].append(np.asarray(robot.g(), dtype=float))
        bar = robot.agent_barrier(obs[0])
        rec["h"].append(float(np.asarray(bar[0]).reshape(-1)[0])); rec["hdot"].append(float(np.asarray(bar[1]).reshape(-1)[0]))
        rec["dhd"].append(np.asarray(bar[2], dtype=float).reshape(-1))
        A, b, u, status = run_cbfqp_case(ctrl, robot, X, u_ref, list(obs))
        robot.X = X.reshape(-1, 1).copy()
        Xn = robot.robot.step(robot.X.copy(), U.reshape(-1, 1)).reshape(-1).copy()
        obs_p = np.full((6, 7), np.nan); obs_p[:K] = obs
        rec["X"].append(X); rec["U"].append(U); rec["u_ref"].append(u_ref); rec["obs"].append(obs_p); rec["k"].append(K)
        rec["step"].append(Xn); rec["A"].append(A); rec["b"].append(b)
        rec["u_star_oracle"].append(np.full(2, np.nan) if u is None else u)
        rec["status_oracle"].append(0 if status == "optimal" else 1)
    for k, v in rec.items():
        out[f"Quad2D/{k}"] = np.array(v)
    out["Quad2D/meta"] = np.array([spec["f_min"], spec["f_max"], spec["radius"], robot.robot_spec["mass"], robot.robot_spec["inertia"]])
    print("Quad2D cases", len(rec["k"]), "infeasible", int(np.sum(rec["status_oracle"])))
    np.savez_compressed(os.path.join(HERE, "quad2d.npz"), **out)


def gen_unicycle2d():
    """Unicycle2D (3 states, inputs v, omega; rel-deg-1 barrier with the sigma(s) term): f, g, step, nominal_input,
    stop, rotate_to and agent_barrier from the reference.  CBFQP.solve_control_problem cannot run for this model as
    checked in (agent_barrier indexes the obstacle as a column, obs[2][0], cbf_qp.py hands over 1-D rows), so the
    barrier is called with a column and the rows are assembled HERE the way cbf_qp.py:155-165 does for rel-deg-1
    models (A = dh_dx g, b = dh_dx f + alpha h, alpha = 1.0 from cbf_qp.py:14-15)."""
    from oracle import qp as oqp
    rng = np.random.default_rng(4242)
    spec = {"model": "Unicycle2D", "v_max": 1.0, "w_max": 0.5, "radius": 0.25}
    robot = BaseRobot(np.zeros((3, 1)), spec, DT, NullAxes())
    rec = {k: [] for k in ["X", "goal", "U", "u_ref", "obs", "k", "f", "g", "step", "nominal", "stop", "rotate", "h", "dh",
                           "A", "b", "u_star_oracle", "status_oracle"]}
    alpha = 1.0
    hi = np.array([spec["v_max"], spec["w_max"]])
    for i in range(140):
        X = np.array([*rng.uniform(0, 14, 2), rng.uniform(-np.pi, np.pi)])
        goal = rng.uniform(0, 14, 2)
        U = np.array([rng.uniform(-1, 1), rng.uniform(-0.5, 0.5)])
        K = int(rng.integers(1, 7))
        obs = draw_circles(rng, np.array([X[0], X[1], X[2], 0.0]), K, spec["radius"], overlap=(i % 6 == 0), rho_max=3.0)
        robot.X = X.reshape(-1, 1).copy()
        fx = np.asarray(robot.f(), dtype=float).reshape(-1)
        gx = np.asarray(robot.g(), dtype=float)
        u_ref = robot.nominal_input(goal).reshape(-1)
        if i % 7 == 3:
            u_ref = u_ref * 3.0
        A = np.zeros((6, 2)); b = np.zeros(6); hs = np.full(6, np.nan); dhs = np.full((6, 3), np.nan)
        for r in range(K):
            h, dh = robot.robot.agent_barrier(robot.X, obs[r].reshape(-1, 1), spec["radius"])
            h = float(np.asarray(h).reshape(-1)[0]); dh = np.asarray(dh, dtype=float).reshape(-1)
            A[r] = dh @ gx
            b[r] = dh @ fx + alpha * h
            hs[r] = h; dhs[r] = dh
        Gb = np.vstack([A, np.eye(2), -np.eye(2)]); cb = np.concatenate([b, hi, hi])
        u, status = oqp.solve_qp2(Gb, cb, u_ref)
        Xn = robot.robot.step(robot.X.copy(), U.reshape(-1, 1)).reshape(-1).copy()
        robot.X = X.reshape(-1, 1).copy()
        obs_p = np.full((6, 7), np.nan); obs_p[:K] = obs
        pad = lambda v: np.concatenate([v, [0.0]])
        rec["X"].append(pad(X)); rec["goal"].append(goal); rec["U"].append(U); rec["u_ref"].append(u_ref)
        rec["obs"].append(obs_p); rec["k"].append(K); rec["f"].append(pad(fx)); rec["g"].append(np.vstack([gx, np.zeros((1, 2))]))
        rec["step"].append(pad(Xn)); rec["nominal"].append(robot.nominal_input(goal).reshape(-1))
        rec["stop"].append(np.asarray(robot.stop(), dtype=float).reshape(-1))
        rec["rotate"].append(np.asarray(robot.rotate_to(0.7), dtype=float).reshape(-1))
        rec["h"].append(hs); rec["dh"].append(dhs); rec["A"].append(A); rec["b"].append(b)
        rec["u_star_oracle"].append(np.full(2, np.nan) if u is None else u)
        rec["status_oracle"].append(int(status))
    out = {f"Unicycle2D/{k}": np.array(v) for k, v in rec.items()}
    print("Unicycle2D cases", len(rec["k"]), "infeasible", int(np.sum(rec["status_oracle"])))
    np.savez_compressed(os.path.join(HERE, "unicycle2d.npz"), **out)


def gen_manipulator():
    """Manipulator2D (3 joints, joint-velocity inputs, one CBF row per link circle per obstacle): f, g, step,
    get_end_effector, get_jacobian, nominal_input, get_link_circles, agent_barrier from the reference's robot class, and
    the rows A1/b1 written by CBFQP.solve_control_problem (cbf_qp.py:110-151, executed verbatim; 'cbf' and 'hard'
    modes).  u* comes from the oracle's n-variable exact solver standing in for cvxpy/GUROBI."""
    rng = np.random.default_rng(777)
    out = {}
    NR = 80                                               # num_obs of the controller = row cap (tracking.py:134-138 uses 150)
    for mode in ("cbf", "hard"):
        spec = {"model": "Manipulator2D", "w_max": 2.0, "Kp": 5.0, "radius": 0.25, "cbf_mode": mode}   # examples/test_tracking.py:124-131
        robot = BaseRobot(np.zeros((3, 1)), spec, DT, NullAxes())
        robot.robot.base_pos = np.array([5.0, 3.5])       # examples/test_tracking.py:163-165
        ctrl = CBFQP(robot, spec, num_obs=NR)

        class Problem3:
            status = "optimal"

            def solve(self, **_):
                G = np.vstack([ctrl.A1.value, np.eye(3), -np.eye(3)])
                c = np.concatenate([ctrl.b1.value.reshape(-1), np.full(6, spec["w_max"])])
                u, st = oqp.solve_qpn(G, c, np.asarray(ctrl.u_ref.value, dtype=float).reshape(-1))
                self.status = "optimal" if st == 0 else "infeasible"
                ctrl.u.value = None if u is None else u.reshape(3, 1)

        ctrl.cbf_controller = Problem3()
        rec = {k: [] for k in ["X", "goal", "U", "u_ref", "obs", "k", "step", "ee", "jac", "nominal", "circles", "h0", "dh0",
                               "A", "b", "u_star_oracle", "status_oracle"]}
        n = 90 if mode == "cbf" else 30
        for i in range(n):
            X = rng.uniform(-np.pi, np.pi, 3)
            goal = np.array([5.0, 3.5]) + rng.uniform(-3, 3, 2)
            U = rng.uniform(-2, 2, 3)
            K = int(rng.integers(1, 5))
            obs = np.zeros((K, 7))
            for r in range(K):                            # obstacles around the arm's workspace; some touch it
                rho, phi = rng.uniform(0.6, 3.6), rng.uniform(-np.pi, np.pi)
                obs[r, 0:3] = [5.0 + rho * np.cos(phi), 3.5 + rho * np.sin(phi), rng.uniform(0.15, 0.5)]
            robot.X = X.reshape(-1, 1).copy()
            u_ref = robot.nominal_input(goal).reshape(-1)
            if i % 5 == 2:
                u_ref = rng.uniform(-3, 3, 3)             # outside the box
            circ = robot.robot.get_link_circles(robot.X, radius=spec["radius"])
            hs, dhs = robot.agent_barrier(obs[0])
            control_ref = {"state_machine": "track", "u_ref": u_ref.reshape(3, 1), "goal": None}
            u = ctrl.solve_control_problem(robot.X, control_ref, list(obs))
            robot.X = X.reshape(-1, 1).copy()
            obs_p = np.full((4, 7), np.nan); obs_p[:K] = obs
            rec["X"].append(X); rec["goal"].append(goal); rec["U"].append(U); rec["u_ref"].append(u_ref)
            rec["obs"].append(obs_p); rec["k"].append(K)
            rec["step"].append(robot.robot.step(robot.X.copy(), U.reshape(-1, 1)).reshape(-1).copy())
            rec["ee"].append(np.asarray(robot.robot.get_end_effector(robot.X), dtype=float).reshape(-1))
            rec["jac"].append(np.asarray(robot.robot.get_jacobian(robot.X), dtype=float))
            rec["nominal"].append(robot.nominal_input(goal).reshape(-1))
            rec["circles"].append(np.array([[c["x"], c["y"], c["link_idx"]] for c in circ]))
            rec["h0"].append(np.array(hs, dtype=float)); rec["dh0"].append(np.array(dhs, dtype=float))
            rec["A"].append(ctrl.A1.value.copy()); rec["b"].append(ctrl.b1.value.reshape(-1).copy())
            rec["u_star_oracle"].append(np.full(3, np.nan) if u is None else np.asarray(u, dtype=float).reshape(-1))
            rec["status_oracle"].append(0 if ctrl.status == "optimal" else 1)
        for k, v in rec.items():
            out[f"{mode}/{k}"] = np.array(v)
        print("Manipulator2D", mode, "cases", n, "infeasible", int(np.sum(rec["status_oracle"])), "circles/obstacle", len(circ))
    out["meta"] = np.array([2.0, 5.0, 0.25, 5.0, 3.5, NR, DT])          # w_max, Kp, radius, base_x, base_y, num_rows, dt
    np.savez_compressed(os.path.join(HERE, "manipulator2d.npz"), **out)


def gen_linear_models():
    """Quad3D (linear 12-state model, MPC-CBF only): f, g and step from the reference's robot class -- what oracle/mpc_lin.py builds its Euler prediction (x + (f + g u) dt, mpc_cbf.py:135-141)
    and the barrier's own one-step map (robot.step: Euler for SI, RK4 for Quad3D) from."""
    rng = np.random.default_rng(31)
    out = {}
    for name, nx, nu in (("Quad3D", 12, 4),):                # SingleIntegrator2D's f / g / step are in integrators.npz
        spec = {"model": name, "radius": 0.25}
        robot = BaseRobot(np.zeros((nx, 1)), spec, DT, NullAxes())
        rec = {k: [] for k in ["X", "U", "f", "g", "step"]}
        for i in range(40):
            X = rng.uniform(-1, 1, nx)
            if name == "Quad3D":
                X[0:3] = rng.uniform(0, 10, 3); X[3:6] = rng.uniform(-0.4, 0.4, 3)     # angles away from the wrap
            else:
                X = rng.uniform(0, 10, 2)
            U = rng.uniform(-3, 3, nu)
            robot.X = X.reshape(-1, 1).copy()
            rec["X"].append(X); rec["U"].append(U)
            rec["f"].append(np.asarray(robot.f(), dtype=float).reshape(-1))
            rec["g"].append(np.asarray(robot.g(), dtype=float))
            rec["step"].append(np.asarray(robot.robot.step(robot.X.copy(), U.reshape(-1, 1)), dtype=float).reshape(-1))
        for k, v in rec.items():
            out[f"{name}/{k}"] = np.array(v)
        out[f"{name}/spec"] = np.array([float(robot.robot_spec.get(k, np.nan)) for k in
                                        ("mass", "Ix", "Iy", "Iz", "L", "nu", "u_max", "u_min", "v_max", "radius")])
        print(name, "cases", len(rec["X"]))
    np.savez_compressed(os.path.join(HERE, "linear_models.npz"), **out)


def gen_closed_loop_integrators():
    """examples/test_tracking.py scene with --model si / di --algo cbf_qp, enable_rotation=False (no attitude controller:
    the yaw of the integrators stays at its initial value and only decides the first state machine state)."""
    out = {}
    known = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0],
                      [7.0, 7.0, 3.0], [4.0, 3.5, 1.5], [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6],
                      [11.0, 5.0, 0.8], [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
    known = np.hstack((known, np.zeros((known.shape[0], 4))))
    wps = np.array([[2, 2, np.pi / 2], [2, 12, 0], [12, 12, 0], [12, 2, 0]], dtype=np.float64)
    cases = [("si", "SingleIntegrator2D", {"v_max": 1.0}, wps[0].copy()),
             ("di", "DoubleIntegrator2D", {"v_max": 1.0, "a_max": 1.0}, np.array([2.0, 2.0, 0.0, 0.0, np.pi / 2])),
             ("di_back", "DoubleIntegrator2D", {"v_max": 1.0, "a_max": 1.0}, np.array([2.0, 2.0, 0.3, -0.4, -np.pi / 2]))]
    for tag, model, extra, x0 in cases:
        spec = dict(extra, model=model, radius=0.25)
        ctl = LocalTrackingController(x0, spec, controller_type={"pos": "cbf_qp"}, dt=DT, env=ref_env.Env(),
                                      enable_rotation=False)
        hi = np.array([spec["v_max"]] * 2) if model == "SingleIntegrator2D" else np.array([spec["a_max"]] * 2)
        ctl.pos_controller.cbf_controller = OracleProblem(ctl.pos_controller, -hi, hi)
        ctl.obs = known.copy()
        ctl.set_waypoints(wps)
        Xs, Us, rets, sms = [ctl.robot.X.reshape(-1).copy()], [], [], [["idle", "track", "stop", "rotate"].index(ctl.state_machine)]
        for _ in range(1200):
            ret = ctl.control_step()
            rets.append(ret); sms.append(["idle", "track", "stop", "rotate"].index(ctl.state_machine))
            if ret == -2:
                break
            Xs.append(ctl.robot.X.reshape(-1).copy()); Us.append(ctl.get_control_input().reshape(-1).copy())
            if ret == -1:
                break
        out[f"{tag}/obs"] = known; out[f"{tag}/waypoints"] = wps; out[f"{tag}/x0"] = x0
        out[f"{tag}/X"] = np.array(Xs); out[f"{tag}/U"] = np.array(Us)
        out[f"{tag}/ret"] = np.array(rets); out[f"{tag}/sm"] = np.array(sms)
        print(tag, "steps", len(rets), "last ret", rets[-1], "first sm", sms[0])
    np.savez_compressed(os.path.join(HERE, "closed_loop_integrators.npz"), **out)


def gen_closed_loop_manipulator():
    """examples/test_tracking.py --model ma --algo cbf_qp (base moved to (5, 3.5), three known obstacles, two waypoints), plus
    a scene with an obstacle in the arm's sweep and a first waypoint outside the field of view (starts in 'stop')."""
    out = {}
    base = np.array([5.0, 3.5])
    scenes = {
        "example": (np.array([[6.0, 4.5, 0.3], [4.0, 1.0, 0.3], [1.0, 4.0, 0.3]]), np.array([[6.5, 4.0, 0.0], [2.0, 4.5, 0.0]]), np.zeros(3)),
        "sweep": (np.array([[6.4, 5.3, 0.25], [3.4, 5.6, 0.3], [7.6, 2.0, 0.3]]), np.array([[5.5, 6.0, 0.0], [3.0, 4.8, 0.0]]), np.array([0.2, 0.4, -0.3])),
        "behind": (np.array([[6.0, 1.6, 0.3], [2.6, 2.2, 0.3]]), np.array([[3.2, 3.0, 0.0], [6.0, 5.2, 0.0]]), np.array([0.0, 0.3, 0.3])),
    }
    for tag, (known, wps, q0) in scenes.items():
        known = np.hstack((known, np.zeros((known.shape[0], 4))))
        spec = {"model": "Manipulator2D", "w_max": 2.0, "Kp": 5.0, "radius": 0.25, "reached_threshold": 0.5}
        ctl = LocalTrackingController(q0.copy(), spec, controller_type={"pos": "cbf_qp"}, dt=DT, env=ref_env.Env())
        ctl.robot.robot.base_pos = base.copy()                     # examples/test_tracking.py:163-165
        pc = ctl.pos_controller

        class Problem3:
            status = "optimal"

            def solve(self, **_):
                G = np.vstack([pc.A1.value, np.eye(3), -np.eye(3)])
                c = np.concatenate([pc.b1.value.reshape(-1), np.full(6, spec["w_max"])])
                u, st = oqp.solve_qpn(G, c, np.asarray(pc.u_ref.value, dtype=float).reshape(-1))
                self.status = "optimal" if st == 0 else "infeasible"
                pc.u.value = None if u is None else u.reshape(3, 1)

        pc.cbf_controller = Problem3()
        ctl.obs = known.copy()
        ctl.set_waypoints(wps)
        names = ["idle", "track", "stop", "rotate"]
        Xs, Us, rets, sms = [ctl.robot.X.reshape(-1).copy()], [], [], [names.index(ctl.state_machine)]
        for _ in range(600):
            ret = ctl.control_step()
            rets.append(ret); sms.append(names.index(ctl.state_machine))
            if ret == -2:
                break
            Xs.append(ctl.robot.X.reshape(-1).copy()); Us.append(ctl.get_control_input().reshape(-1).copy())
            if ret == -1:
                break
        out[f"{tag}/obs"] = known; out[f"{tag}/waypoints"] = wps; out[f"{tag}/q0"] = q0
        out[f"{tag}/filtered_waypoints"] = np.asarray(ctl.waypoints, dtype=float)
        out[f"{tag}/X"] = np.array(Xs); out[f"{tag}/U"] = np.array(Us)
        out[f"{tag}/ret"] = np.array(rets); out[f"{tag}/sm"] = np.array(sms)
        print(tag, "steps", len(rets), "last ret", rets[-1], "first sm", names[sms[0]], "constrained steps",
              int(np.sum(np.abs(np.array(Us) - np.clip(np.array(Us), -2, 2)).sum(axis=1) >= 0)))
    out["base"] = base
    np.savez_compressed(os.path.join(HERE, "closed_loop_manipulator.npz"), **out)



# ---------------------------------------------------------------------------------------------------------------
# MPC-CBF problem functions, executed from the reference's own source
# ---------------------------------------------------------------------------------------------------------------
MPC_MODELS = {
    # name: (BaseRobot X0 rows, nx, nu, spec)  -- specs are the ones examples/test_tracking.py passes per model
    "SingleIntegrator2D": (3, 2, 2, {"v_max": 1.0, "radius": 0.25}),
    "Unicycle2D": (3, 3, 2, {"v_max": 1.0, "w_max": 0.5, "radius": 0.25}),
    "DynamicUnicycle2D": (4, 4, 2, {"a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25}),
    "DoubleIntegrator2D": (5, 4, 2, {"v_max": 1.0, "a_max": 1.0, "radius": 0.25}),
    "KinematicBicycle2D": (4, 4, 2, {"a_max": 0.5, "radius": 0.5}),
    "KinematicBicycle2D_C3BF": (4, 4, 2, {"a_max": 0.5, "radius": 0.5}),
    "KinematicBicycle2D_DPCBF": (4, 4, 2, {"a_max": 0.5, "radius": 0.5}),
    "Quad2D": (6, 6, 2, {"f_min": 3.0, "f_max": 10.0, "radius": 0.25}),
    "Quad3D": (12, 12, 4, {"radius": 0.25}),
    "VTOL2D": (6, 6, 4, {"radius": 0.6}),
}


def _draw_mpc_state(rng, name):
    if name == "SingleIntegrator2D":
        return rng.uniform(0, 14, 2)
    if name == "Unicycle2D":
        return np.array([*rng.uniform(0, 14, 2), rng.uniform(-4.0, 4.0)])           # heading past +-pi: no wrap in the model
    if name == "DynamicUnicycle2D":
        return np.array([*rng.uniform(0, 14, 2), rng.uniform(-4.0, 4.0), rng.uniform(-1.0, 1.0)])
    if name == "DoubleIntegrator2D":
        return np.array([*rng.uniform(0, 14, 2), *rng.uniform(-1.2, 1.2, 2)])        # some speeds above v_max (rescaled in step)
    if name.startswith("KinematicBicycle2D"):
        return np.array([*rng.uniform(0, 14, 2), rng.uniform(-4.0, 4.0), rng.uniform(-0.3, 4.0)])   # both sides of the v clip
    if name == "Quad2D":
        return np.array([*rng.uniform(0, 14, 2), rng.uniform(-0.6, 0.6), *rng.uniform(-1.5, 1.5, 2), rng.uniform(-1, 1)])
    if name == "Quad3D":
        X = rng.uniform(-1, 1, 12); X[0:3] = rng.uniform(0, 10, 3); X[3:6] = rng.uniform(-0.4, 0.4, 3)
        return X
    if name == "VTOL2D":
        return np.array([rng.uniform(0, 60), rng.uniform(2, 14), rng.uniform(-0.3, 0.3), rng.uniform(3, 15),
                         rng.uniform(-3, 3), rng.uniform(-0.5, 0.5)])
    raise KeyError(name)


def _draw_mpc_input(rng, name, spec):
    if name == "SingleIntegrator2D":
        return rng.uniform(-1.2, 1.2, 2)
    if name == "Unicycle2D":
        return np.array([rng.uniform(-1.2, 1.2), rng.uniform(-0.6, 0.6)])
    if name == "DynamicUnicycle2D":
        return np.array([rng.uniform(-1.2, 1.2), rng.uniform(-0.6, 0.6)])
    if name == "DoubleIntegrator2D":
        return rng.uniform(-1.2, 1.2, 2)
    if name.startswith("KinematicBicycle2D"):
        return np.array([rng.uniform(-6, 6), rng.uniform(-0.35, 0.35)])
    if name == "Quad2D":
        return rng.uniform(2.0, 11.0, 2)
    if name == "Quad3D":
        return rng.uniform(-3, 12, 4)
    if name == "VTOL2D":
        return np.array([*rng.uniform(0, 1, 3), rng.uniform(-0.5, 0.5)])
    raise KeyError(name)


def _draw_mpc_obstacles(rng, name, X, K, R, i):
    """K rows of 7: circles near the robot, superellipsoids where the model's DT barrier has that branch, moving
    circles for the C3BF / DPCBF barriers, and the [1000, 1000, 0, ...] dummy row update_tvp pads with."""
    pos = np.array([X[0], X[1], 0.0, 0.0])
    obs = draw_circles(rng, pos, K, R, moving=name in ("KinematicBicycle2D_C3BF", "KinematicBicycle2D_DPCBF"),
                       overlap=(i % 7 == 0))
    obs = np.asarray(obs, dtype=float)
    if name in ("SingleIntegrator2D", "DynamicUnicycle2D", "DoubleIntegrator2D") and i % 3 == 1:
        se = np.asarray(draw_superellipsoids(rng, pos, K, R), dtype=float)
        pick = rng.random(K) < 0.6
        obs[pick] = se[pick]
    if i % 5 == 2:
        obs[-1] = [1000.0, 1000.0, 0, 0, 0, 0, 0]
    return obs


def gen_mpc_functions():
    """Executes, under the numeric casadi / do_mpc recorder (tests/golden/_casadi_numeric.py), the reference's OWN
    MPCCBF.__init__ (weights, horizon, gains, overrides: position_control/mpc_cbf.py:7-100), create_model (:108-160:
    x_next, stage cost), create_mpc (:162-259: n_horizon, rterm, bounds), set_cbf_constraint / compute_cbf_constraint
    (:295-325) with each robot's agent_barrier_dt, and update_tvp / tvp_fun (:261-293, :338-364) -- for every model the
    class accepts.  Stored per model: the tables, and on seeded (x, u, goal, obs) draws the values x_next, cost,
    (h_k, d_h[, dd_h]) per obstacle and the registered constraint expressions (-cbf <= 0)."""
    import _casadi_numeric as CN
    CN.install_casadi()
    CN.install_do_mpc()
    import importlib
    for modname in [m for m in list(sys.modules) if m.startswith("safe_control.") and ("robots." in m or "dynamic_env." in m or m.endswith("mpc_cbf"))]:
        del sys.modules[modname]                   # re-import the robot modules against the numeric casadi
    robot_mod = importlib.import_module("safe_control.robots.robot")
    importlib.reload(robot_mod)
    mpc_mod = importlib.import_module("safe_control.position_control.mpc_cbf")
    BaseRobotN, MPCCBFRef = robot_mod.BaseRobot, mpc_mod.MPCCBF
    import casadi as ca

    rng = np.random.default_rng(4242)
    out = {}
    K = 4
    for name, (rows0, nx, nu, spec0) in MPC_MODELS.items():
        spec = dict(spec0, model=name)
        robot = BaseRobotN(np.zeros((rows0, 1)), spec, DT, NullAxes())
        spec = robot.robot_spec
        CN.POINT.clear()
        ctrl = MPCCBFRef(robot, spec, num_obs=K)
        # ---- tables
        out[f"{name}/Q"] = np.asarray(ctrl.Q, dtype=float)
        out[f"{name}/R"] = np.asarray(ctrl.R, dtype=float)
        out[f"{name}/horizon"] = np.array(ctrl.horizon)
        out[f"{name}/n_horizon_param"] = np.array(ctrl.mpc.params["n_horizon"])
        out[f"{name}/t_step"] = np.array(ctrl.mpc.params["t_step"])
        out[f"{name}/n_states"] = np.array(ctrl.n_states)
        out[f"{name}/n_controls"] = np.array(ctrl.n_controls)
        out[f"{name}/goal_init"] = np.asarray(ctrl.goal, dtype=float)
        out[f"{name}/rterm_u"] = ctrl.mpc.rterm["u"]
        for key, val in ctrl.cbf_param.items():
            out[f"{name}/cbf_param/{key}"] = np.array(float(val))
        lo_u = np.full(nu, -np.inf); hi_u = np.full(nu, np.inf); lo_x = np.full(nx, -np.inf); hi_x = np.full(nx, np.inf)
        for key, val in ctrl.mpc.bounds.items():
            side, vt = key[0], key[1]
            tgt = {("lower", "_u"): lo_u, ("upper", "_u"): hi_u, ("lower", "_x"): lo_x, ("upper", "_x"): hi_x}[(side, vt)]
            if len(key) == 4:
                tgt[key[3]] = float(val)
            else:
                tgt[:] = np.asarray(val, dtype=float).reshape(-1)
        out[f"{name}/u_lo"], out[f"{name}/u_hi"], out[f"{name}/x_lo"], out[f"{name}/x_hi"] = lo_u, hi_u, lo_x, hi_x
        out[f"{name}/spec_keys"] = np.array(sorted(k for k, v in spec.items() if isinstance(v, (int, float)) and k != "model"))
        out[f"{name}/spec_vals"] = np.array([float(spec[k]) for k in out[f"{name}/spec_keys"]])
        out[f"{name}/robot_radius"] = np.array(float(robot.robot_radius))
        # gain overrides (mpc_cbf.py:90-95)
        ov = dict(spec, mpc_cbf_alpha=0.31, mpc_cbf_alpha1=0.27, mpc_cbf_alpha2=0.19, mpc_horizon=7)
        c2 = MPCCBFRef(BaseRobotN(np.zeros((rows0, 1)), dict(ov), DT, NullAxes()), dict(ov), num_obs=2)
        out[f"{name}/override/keys"] = np.array(sorted(c2.cbf_param))
        out[f"{name}/override/vals"] = np.array([float(c2.cbf_param[k]) for k in sorted(c2.cbf_param)])
        out[f"{name}/override/horizon"] = np.array(c2.horizon)
        # ---- values on draws
        rec = {k: [] for k in ["x", "u", "goal", "obs", "x_next", "cost", "cost_next", "f", "g", "hk", "dh", "ddh", "cons", "step"]}
        n_cases = 48
        for i in range(n_cases):
            x = _draw_mpc_state(rng, name)
            u = _draw_mpc_input(rng, name, spec)
            goal = rng.uniform(0, 14, 3 if name == "Quad3D" else 2)
            obs = _draw_mpc_obstacles(rng, name, x, K, float(robot.robot_radius), i)
            goal_pad = np.concatenate([goal, np.zeros(nx - goal.shape[0])])
            CN.POINT.clear()
            CN.POINT.update(x=x.reshape(nx, 1), u=u.reshape(nu, 1), goal=goal_pad.reshape(nx, 1), obs=obs,
                            alpha=np.array([[ctrl.cbf_param.get("alpha", 0.0)]]),
                            alpha1=np.array([[ctrl.cbf_param.get("alpha1", 0.0)]]),
                            alpha2=np.array([[ctrl.cbf_param.get("alpha2", 0.0)]]))
            ctrl.setup_control_problem()                      # re-traces create_model / create_mpc at this point
            rec["x"].append(x); rec["u"].append(u); rec["goal"].append(goal); rec["obs"].append(obs)
            rec["x_next"].append(ctrl.model.rhs["x"].v.reshape(-1))
            rec["cost"].append(float(ctrl.model.aux["cost"]))
            rec["f"].append(np.array(CN._val(robot.f_casadi(ca.SX(x.reshape(nx, 1))))).reshape(-1))
            rec["g"].append(np.array(CN._val(robot.g_casadi(ca.SX(x.reshape(nx, 1))))))
            cons = []
            for j in range(K):
                expr, ub = ctrl.mpc.nl_cons[f"cbf_{j}"]
                assert ub == 0
                cons.append(float(expr))
            rec["cons"].append(cons)
            hk, dh, ddh = [], [], []
            for j in range(K):
                r = robot.agent_barrier_dt(ca.SX(x.reshape(nx, 1)), ca.SX(u.reshape(nu, 1)), ca.SX(obs[j].reshape(1, 7)))
                hk.append(float(r[0])); dh.append(float(r[1])); ddh.append(float(r[2]) if len(r) > 2 else np.nan)
            rec["hk"].append(hk); rec["dh"].append(dh); rec["ddh"].append(ddh)
            CN.POINT["x"] = rec["x_next"][-1].reshape(nx, 1)      # the stage cost at the predicted state (what lterm / mterm see next)
            ctrl.setup_control_problem()
            rec["cost_next"].append(float(ctrl.model.aux["cost"]))
            # the robot's numpy step() -- what the DT barrier composes, minus casadi's fmod/clip spelling
            try:
                rec["step"].append(np.asarray(robot.robot.step(x.reshape(nx, 1).copy(), u.reshape(nu, 1)), dtype=float).reshape(-1))
            except Exception:
                rec["step"].append(np.full(nx, np.nan))
        for k, v in rec.items():
            out[f"{name}/{k}"] = np.array(v, dtype=float)
        # ---- update_tvp / tvp_fun (mpc_cbf.py:261-293, :338-364)
        tv = {}
        cases = {"none": None, "empty": [], "three_wide": [[1.0, 2.0, 0.3], [4.0, 5.0, 0.6]],
                 "seven_wide": [[1.0, 2.0, 0.3, 0.1, -0.2, 0.0, 0.0], [3.0, 3.0, 0.5, 0.8, 4.0, 0.3, 1.0]],
                 "too_many": [[float(j), float(j) + 1.0, 0.2 + 0.1 * j] for j in range(K + 2)]}
        for cname, ob in cases.items():
            goal = np.array([3.5, -1.25, 2.0])[: (3 if name == "Quad3D" else 2)]
            ctrl.update_tvp(goal, None if ob is None else [np.array(o) for o in ob])
            tpl = ctrl.mpc.tvp_fun(0.0)
            tv[cname] = tpl
            out[f"{name}/tvp/{cname}/obs_attr"] = np.asarray(ctrl.obs, dtype=float)
            out[f"{name}/tvp/{cname}/obs"] = tpl.values["obs"]
            out[f"{name}/tvp/{cname}/goal"] = tpl.values["goal"].reshape(-1)
            for key in ("alpha", "alpha1", "alpha2"):
                if key in tpl.values:
                    out[f"{name}/tvp/{cname}/{key}"] = tpl.values[key]
        raised = False
        try:
            ctrl.update_tvp(np.zeros(2), [np.array([1.0, 2.0, 0.3, 0.0, 0.0])])
        except ValueError:
            raised = True
        out[f"{name}/tvp/five_wide_raises"] = np.array(raised)
        print(name, "mpc cases", n_cases, "cons[0] =", rec["cons"][0])
    np.savez_compressed(os.path.join(HERE, "mpc_functions.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "closed_loop_manipulator":
        gen_closed_loop_manipulator()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "closed_loop_integrators":
        gen_closed_loop_integrators()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mpc_functions":
        gen_mpc_functions()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "linear_models":
        gen_linear_models()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "manipulator2d":
        gen_manipulator()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "unicycle2d":
        gen_unicycle2d()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "quad2d":
        gen_quad2d()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "integrators":
        gen_integrators()
        sys.exit(0)
    rng = np.random.default_rng(20240611)
    gen_callbacks(rng)
    gen_cbfqp(rng)
    gen_nearest(rng)
    gen_closed_loop()
    gen_integrators()
    gen_quad2d()
    gen_unicycle2d()
    gen_manipulator()
    gen_linear_models()
    gen_closed_loop_integrators()
    gen_closed_loop_manipulator()
    gen_mpc_functions()        # last: it swaps the inert casadi / do_mpc stand-ins for the numeric ones
