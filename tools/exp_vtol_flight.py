"""GPU: VTOL2D closed loops with the reference solver's budget (max_iter 3000, continuation launches): per-step MPC status and
iteration count, and where the flight ends.  Scenes: "own" = the scene of tests/test_tracking_vtol_gpu.py, "ref" = the reference's
examples/test_vtol.py:21-64 (20 m/s at (2, 10), 24 discs, goal (70, 10)), "easy" = cruise past one disc.
    python3 tools/exp_vtol_flight.py [scene] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca

scene = sys.argv[1] if len(sys.argv) > 1 else "own"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
if scene == "own":
    spec = {"model": "VTOL2D", "num_constraints": 4, "reached_threshold": 3.0}
    obs = np.array([[80.0, 10.5, 1.5], [95.0, 7.0, 1.0], [-20.0, 10.0, 1.0], [40.0, 30.0, 1.0], [-5.0, 12.0, 0.5], [130.0, 12.0, 1.0]])
    x0 = np.array([0.0, 10.0, 0.0, 12.0, 0.0, 0.0]); wps = np.array([[0.0, 10.0], [60.0, 10.0], [120.0, 10.0]])
elif scene == "easy":
    spec = {"model": "VTOL2D", "num_constraints": 4, "reached_threshold": 3.0}
    obs = np.array([[60.0, 13.5, 1.0], [120.0, 6.0, 1.0]])
    x0 = np.array([0.0, 10.0, 0.0, 12.0, 0.0, 0.0]); wps = np.array([[0.0, 10.0], [150.0, 10.0]])
else:
    p1, p2 = 67.0, 73.0
    obs = np.array([[p1, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[p2, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]])
    spec = {"model": "VTOL2D", "radius": 0.6, "v_max": 20.0, "reached_threshold": 1.0, "num_constraints": 10}
    x0 = np.array([2.0, 10.0, 0.0, 20.0, 0.0, 0.0]); wps = np.array([[2.0, 10.0], [70.0, 10.0], [70.0, 0.5]])
obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
ctl = sca.BatchedTrackingController(x0[None, :], dict(spec), obs=obs7, device="cuda:0")
ctl.set_waypoints(wps)
print("mpc budget", ctl.mpc.max_iter, "slices", ctl.mpc.iter_slices)
names = {0: "opt", 1: "infeas", 2: "inacc"}
log = []
dump = []
for k in range(steps):
    ret = int(ctl.control_step(1)[0].item())
    X = ctl.X[0].cpu().numpy()
    st = int(ctl.mpc_status[0].item())
    it = int(getattr(ctl, "mpc_iters", torch.zeros(1))[0].item()) if hasattr(ctl, "mpc_iters") else -1
    log.append((k, st, it, X[0], X[1], X[2], X[3]))
    dump.append(dict(k=k, X_after=X.copy(), u_prev=ctl.u_prev[0].cpu().numpy().copy(), goal=ctl.goal[0].cpu().numpy().copy(), st=st, it=it))
    if k < 8 or k % 10 == 0 or ret != 0:
        print(f"step {k:3d} ret {ret:2d} mpc {names.get(st, st):6s} it {it:5d}  x {X[0]:7.2f} z {X[1]:6.2f} pitch {np.degrees(X[2]):6.1f} deg vx {X[3]:6.2f}  u {ctl.u_pos[0].cpu().numpy().round(3)}")
    if ret != 0:
        break
sts = np.array([l[1] for l in log])
print("steps flown", len(log), "ret", ret, "statuses opt/infeas/inacc", [(sts == s).sum() for s in (0, 1, 2)], "goal index", int(ctl.current_goal_index[0].item()))
print("status per step:", "".join(str(l[1]) for l in log))
print("iterations per step:", [l[2] for l in log])

if len(sys.argv) > 3:
    np.savez(sys.argv[3], X=np.array([d["X_after"] for d in dump]), u_prev=np.array([d["u_prev"] for d in dump]), goal=np.array([d["goal"] for d in dump]),
             st=np.array([d["st"] for d in dump]), it=np.array([d["it"] for d in dump]), obs=obs7, spec_radius=spec.get("radius", 0.0), v_max=spec.get("v_max", 0.0))
