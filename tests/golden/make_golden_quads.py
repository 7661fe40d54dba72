#!/usr/bin/env python3
"""Golden vectors for the closed loop of Quad2D / Quad3D (SURVEY 8f-1 over the 8f-3 models) from the reference's own code.

Run ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_quads.py

Executed verbatim from the reference: ``Quad2D`` / ``Quad3D`` ``nominal_input``, ``stop``, ``has_stopped``, ``rotate_to``,
``step`` (robots/quad2D.py:83-164, robots/quad3D.py:100-257) and ``LocalTrackingController.control_step`` / ``set_waypoints``
/ ``update_goal`` / ``get_nearest_unpassed_obs`` (tracking.py:197-249,345-403,497-535,559-668) with these robots on the scene
of examples/test_tracking.py.  NOT from the reference: the position controller's solve -- do-mpc / casadi / IPOPT are
absent, so a stand-in ``MPCCBF`` that follows the reference's call protocol (mpc_cbf.py:366-402: pass u_ref through outside
'track', goal and padded obstacles from control_ref / nearest_obs, previous MPC input as u0) hands the NLP to this repo's
numpy oracle (oracle/mpc_gn.py, oracle/mpc_lin.py).  The fixtures therefore pin everything AROUND the solve.

Writes tests/golden/closed_loop_quads.npz."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402

_ref_import.install()

from oracle import mpc_gn as OG, mpc_lin as OL, mpc_cbf as OM  # noqa: E402

N_HORIZON = 10
LOG = {"solves": 0}


class PlugMPC:
    """Stands where position_control.mpc_cbf.MPCCBF would (mpc_cbf.py:7-402): same constructor, same call protocol."""

    def __init__(self, robot, robot_spec, show_mpc_traj=False, num_obs=5):
        self.robot, self.robot_spec, self.num_obs = robot, robot_spec, num_obs
        self.status = "optimal"                                # mpc_cbf.py:10
        self.nu = 4 if robot_spec["model"] == "Quad3D" else 2
        self.u_prev = np.zeros(self.nu)
        if robot_spec["model"] == "Quad2D":
            self.u_prev = None                                 # first call: the initial guess of do-mpc is u0 = 0 -> clipped into the box
        self.records = []

    def solve_control_problem(self, robot_state, control_ref, nearest_obs):
        if control_ref["state_machine"] != "track":            # mpc_cbf.py:379-381
            return control_ref["u_ref"]
        x = np.asarray(robot_state, dtype=float).reshape(-1)
        goal = np.asarray(control_ref["goal"], dtype=float).reshape(-1)
        obs = np.tile(OM.DUMMY_OBS, (self.num_obs, 1))         # update_tvp, mpc_cbf.py:338-364
        if nearest_obs is not None:
            for j, ob in enumerate(list(nearest_obs)[: self.num_obs]):
                ob = np.asarray(ob, dtype=float).reshape(-1)
                obs[j, : min(7, ob.shape[0])] = ob[:7]
        up = np.zeros(self.nu) if self.u_prev is None else self.u_prev
        if self.robot_spec["model"] == "Quad2D":
            u, st, it = OG.solve(OG.quad2d_model(dict(self.robot_spec), dt=self.robot.dt), x, up, goal[:2], obs, N=N_HORIZON)[:3]
        else:
            u, st, it = OL.solve(OL.quad3d_model(dict(self.robot_spec), dt=self.robot.dt), x, up, goal[:3], obs, N=N_HORIZON)[:3]
        LOG["solves"] += 1
        self.records.append((x.copy(), up.copy(), goal.copy(), obs.copy(), np.asarray(u).copy(), int(st)))
        self.u_prev = np.asarray(u, dtype=float).copy()
        return np.asarray(u, dtype=float).reshape(-1, 1)


_m = types.ModuleType("safe_control.position_control.mpc_cbf")
_m.MPCCBF = PlugMPC
sys.modules["safe_control.position_control.mpc_cbf"] = _m

from safe_control.tracking import LocalTrackingController  # noqa: E402
from safe_control.utils import env as ref_env  # noqa: E402
from safe_control.robots.quad2D import Quad2D  # noqa: E402
from safe_control.robots.quad3D import Quad3D  # noqa: E402

DT = 0.05
NAMES = ["idle", "track", "stop", "rotate"]
KNOWN = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0],
                  [4.0, 3.5, 1.5], [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8], [13.5, 11.0, 0.6],
                  [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])                      # examples/test_tracking.py:52-54


def gen_functions(out):
    rng = np.random.default_rng(20261002)
    q2 = Quad2D(DT, {"f_min": 3.0, "f_max": 10.0, "radius": 0.25})
    q3 = Quad3D(DT, {"radius": 0.25})
    n = 48
    X2 = np.column_stack([rng.uniform(0, 14, n), rng.uniform(0, 14, n), rng.uniform(-1.0, 1.0, n), rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(-1, 1, n)])
    X2[:6, 3:5] *= 0.01                                                       # a few nearly stopped states
    G2 = rng.uniform(0, 14, (n, 2))
    U2 = rng.uniform(3.0, 10.0, (n, 2))
    out["q2/X"], out["q2/goal"], out["q2/U"] = X2, G2, U2
    out["q2/nominal"] = np.array([q2.nominal_input(X2[i].reshape(-1, 1), G2[i].reshape(-1, 1)).flatten() for i in range(n)])
    out["q2/stop"] = np.array([np.asarray(q2.stop(X2[i].reshape(-1, 1))).flatten() for i in range(n)])
    out["q2/has_stopped"] = np.array([bool(q2.has_stopped(X2[i].reshape(-1, 1))) for i in range(n)])
    out["q2/step"] = np.array([q2.step(X2[i].reshape(-1, 1).copy(), U2[i].reshape(-1, 1)).flatten() for i in range(n)])
    X3 = np.zeros((n, 12))
    X3[:, 0:2] = rng.uniform(0, 14, (n, 2)); X3[:, 2] = rng.uniform(0, 3, n); X3[:, 3:5] = rng.uniform(-0.3, 0.3, (n, 2))
    X3[:, 5] = rng.uniform(-3.1, 3.1, n); X3[:, 6:9] = rng.uniform(-1.5, 1.5, (n, 3)); X3[:, 9:12] = rng.uniform(-0.5, 0.5, (n, 3))
    X3[:6, 6:12] *= 0.01
    G3 = np.column_stack([rng.uniform(0, 14, (n, 2)), rng.uniform(0, 3, n)])
    U3 = rng.uniform(-10, 10, (n, 4))
    ang = rng.uniform(-3.1, 3.1, n)
    out["q3/X"], out["q3/goal"], out["q3/U"], out["q3/ang"] = X3, G3, U3, ang
    out["q3/nominal"] = np.array([q3.nominal_input(X3[i].reshape(-1, 1), G3[i]).flatten() for i in range(n)])
    out["q3/stop"] = np.array([np.asarray(q3.stop(X3[i].reshape(-1, 1))).flatten() for i in range(n)])
    out["q3/has_stopped"] = np.array([bool(q3.has_stopped(X3[i].reshape(-1, 1))) for i in range(n)])
    out["q3/rotate_to"] = np.array([np.asarray(q3.rotate_to(X3[i].reshape(-1, 1), ang[i])).flatten() for i in range(n)])
    out["q3/step"] = np.array([q3.step(X3[i].reshape(-1, 1).copy(), U3[i].reshape(-1, 1)).flatten() for i in range(n)])
    out["q3/pinvB2"] = np.linalg.pinv(q3.B2)


def run_loop(tag, model, spec, x0, wps, steps, out, enable_rotation=True):
    known = np.hstack((KNOWN, np.zeros((KNOWN.shape[0], 4))))
    ctl = LocalTrackingController(np.asarray(x0, dtype=float), dict(spec, model=model), controller_type={"pos": "mpc_cbf"}, dt=DT,
                                  env=ref_env.Env(), enable_rotation=enable_rotation)
    ctl.obs = known.copy()
    ctl.set_waypoints(np.asarray(wps, dtype=float))
    Xs, Us, rets, sms = [ctl.robot.X.reshape(-1).copy()], [], [], [NAMES.index(ctl.state_machine)]
    idx = [ctl.current_goal_index]
    for _ in range(steps):
        ret = ctl.control_step()
        rets.append(ret); sms.append(NAMES.index(ctl.state_machine)); idx.append(ctl.current_goal_index)
        if ret == -2:
            break
        Xs.append(ctl.robot.X.reshape(-1).copy()); Us.append(ctl.get_control_input().reshape(-1).copy())
        if ret == -1:
            break
    out[f"{tag}/obs"] = known; out[f"{tag}/waypoints"] = np.asarray(wps, dtype=float); out[f"{tag}/x0"] = np.asarray(x0, dtype=float)
    out[f"{tag}/filtered_waypoints"] = np.asarray(ctl.waypoints, dtype=float)
    out[f"{tag}/X"] = np.array(Xs); out[f"{tag}/U"] = np.array(Us); out[f"{tag}/ret"] = np.array(rets); out[f"{tag}/sm"] = np.array(sms)
    out[f"{tag}/goal_index"] = np.array(idx)
    rec = ctl.pos_controller.records
    out[f"{tag}/mpc_x"] = np.array([r[0] for r in rec]); out[f"{tag}/mpc_u_prev"] = np.array([r[1] for r in rec])
    out[f"{tag}/mpc_goal"] = np.array([r[2] for r in rec]); out[f"{tag}/mpc_obs"] = np.array([r[3] for r in rec])
    out[f"{tag}/mpc_u"] = np.array([r[4] for r in rec]); out[f"{tag}/mpc_status"] = np.array([r[5] for r in rec])
    print(tag, "steps", len(rets), "last ret", rets[-1], "first sm", NAMES[sms[0]], "sm seen", sorted(set(sms)), "MPC solves", len(rec),
          "non-optimal", int(np.sum(out[f"{tag}/mpc_status"] != 0)), "final", np.round(Xs[-1][:3], 2))


def gen():
    out = {}
    gen_functions(out)
    wps = np.array([[2, 2, np.pi / 2], [2, 12, 0], [12, 12, 0], [12, 2, 0]], dtype=np.float64)     # examples/test_tracking.py:43-48
    q2 = {"f_min": 3.0, "f_max": 10.0, "radius": 0.25}
    q3 = {"radius": 0.25}
    steps = int(os.environ.get("QUAD_GOLDEN_STEPS", "260"))
    run_loop("q2_example", "Quad2D", q2, wps[0], wps, steps, out)                       # --model quad (x_init = waypoints[0])
    run_loop("q2_behind", "Quad2D", q2, np.array([6.0, 2.0, 0.0, 0.5, 0.3, 0.0]), np.array([[2.5, 2.2, 0.0], [2.0, 6.0, 0.0]]), steps, out)
    run_loop("q3_example", "Quad3D", q3, wps[0], wps, steps, out)                       # --model quad3d: the third column is the z goal
    run_loop("q3_behind", "Quad3D", q3, np.array([10.0, 2.5, 1.0, 0.3]), np.array([[8.0, 5.0, 1.5], [3.0, 3.0, 1.0]]), steps, out)
    np.savez_compressed(os.path.join(HERE, "closed_loop_quads.npz"), **out)
    print("wrote closed_loop_quads.npz", os.path.getsize(os.path.join(HERE, "closed_loop_quads.npz")), "bytes")


if __name__ == "__main__":
    gen()
