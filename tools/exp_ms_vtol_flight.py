"""CPU: the reference's examples/test_vtol.py scene (:12-64: 20 m/s at (2, 10), 24 discs, waypoints (70, 10) -> (70, 0.5)) flown by the
oracle loop (oracle/tracking_quad.py) with the do-mpc-faithful MULTIPLE-SHOOTING solver (oracle/ms_ipopt.py) as position controller.
    python3 tools/exp_ms_vtol_flight.py [steps] [out.npz] [solver: ms | condensed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ms_ipopt as MS, mpc_vtol as OV
from oracle.tracking_quad import QuadTrackingOracle

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out = sys.argv[2] if len(sys.argv) > 2 else None
which = sys.argv[3] if len(sys.argv) > 3 else "ms"           # ms | riccati (same algorithm, the kernel's linear algebra) | condensed
p1, p2 = 67.0, 73.0
obs = np.array([[p1, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[p2, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]])
obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
spec = dict(radius=0.6, v_max=20.0, reached_threshold=1.0)
mdl = MS.vtol_model(dict(radius=0.6, v_max=20.0))
# ms: dense LDL', IPOPT's restoration;  riccati: the same with the stage-wise linear algebra;  kernel: what csrc/mpc_vtol_ms.hip runs (riccati +
# a restoration that keeps the dynamics rows hard)
OPTS = {"ms": None, "riccati": dict(linear_solver="riccati"), "kernel": dict(linear_solver="riccati", resto_elastic="ineq"),
        "hybrid": dict(linear_solver="riccati", restoration="none"),
        "hybrid_nosoc": dict(linear_solver="riccati", restoration="none", max_soc=0),
        "kernel_nosoc": dict(linear_solver="riccati", resto_elastic="ineq", max_soc=0),
        "profile": dict(MS.KERNEL_PROFILE)}.get(which)                       # profile: exactly what csrc/mpc_vtol_ms.hip is held to (stall / floor rules included)   # hybrid: regular phase here, the condensed oracle when it asks for a restoration
log = []


rec = []                                                                    # REC=<file.npz>: the inputs and results of every solve (replayed on the GPU: tools/exp_vtol_lost_replay.py)


def solve_ms(X, up, goal, ob):
    t0 = time.time()
    u, st, it, info = MS.solve(mdl, X, up, goal, ob, return_info=True, opts=OPTS)
    if os.environ.get("REC"):
        rec.append(dict(X=np.array(X, float), up=np.array(up, float), goal=np.array(goal, float), ob=np.array(ob, float), u=np.array(u, float), st=st, it=it, status=info["status"]))
        np.savez(os.environ["REC"], **{k: np.array([r[k] for r in rec]) for k in rec[0]})
    if info["status"] == "needs_resto":
        u2, st2, it2 = OV.solve(X, up, goal[:2], ob, N=30, spec=dict(spec))
        log.append(dict(st="resto->condensed:%d" % st2, it=it + it2, dt=time.time() - t0, viol=0.0, u=u2.copy()))
        return u2
    log.append(dict(st=info["status"], it=it, dt=time.time() - t0, viol=float(max(np.abs(info["c"]).max(), info["d"].max())), u=u.copy()))
    return u


def solve_cond(X, up, goal, ob):
    t0 = time.time()
    u, st, it = OV.solve(X, up, goal[:2], ob, N=30, spec=dict(spec))
    log.append(dict(st={0: "optimal", 1: "local_infeasibility", 2: "inaccurate"}[st], it=it, dt=time.time() - t0, viol=0.0, u=u.copy()))
    return u


_start = [float(v) for v in os.environ["START"].split(",")] if os.environ.get("START") else [2.0, 10.0, 20.0]        # START="x,z,speed": another start (tools/exp_vtol_fleet.py)
o = QuadTrackingOracle("VTOL2D", np.array([_start[0], _start[1], 0.0, _start[2], 0.0, 0.0]), spec=spec, obs=obs7, num_constraints=10,
                       solve_fn=solve_cond if which == "condensed" else solve_ms)
o.set_waypoints(np.array([[70.0, 10.0], [70.0, 0.5]]) if os.environ.get("START") else np.array([[2.0, 10.0], [70.0, 10.0], [70.0, 0.5]]))
traj = []
ret = 0
for k in range(steps):
    n0 = len(log)
    ret = o.control_step()
    l = log[-1] if len(log) > n0 else dict(st="-", it=0, dt=0.0, viol=0.0, u=np.zeros(4))
    traj.append(np.concatenate([o.X, l["u"], [o.current_goal_index, l["it"], {"optimal": 0, "acceptable": 0, "local_infeasibility": 1}.get(l["st"], 2)]]))
    print(f"step {k:3d} ret {ret:2d} {l['st']:22s} it {l['it']:4d} {l['dt']:6.1f}s viol {l['viol']:.1e} x {o.X[0]:7.2f} z {o.X[1]:6.2f} pitch {np.degrees(o.X[2]):6.1f} vx {o.X[3]:6.2f} vz {o.X[4]:6.2f} u {np.round(l['u'], 3)} goal {o.current_goal_index}", flush=True)
    if out and k % 10 == 0:
        np.savez(out, traj=np.array(traj), ret=ret)
    if ret != 0:
        break
print("flight ended: ret", ret, "after", len(traj), "control steps; goal index", o.current_goal_index)
if out:
    np.savez(out, traj=np.array(traj), ret=ret)
