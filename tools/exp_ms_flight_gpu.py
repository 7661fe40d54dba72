"""GPU: the reference's examples/test_vtol.py scene through the drop-in loop with the multiple-shooting kernel (default) or the condensed one.
    python3 tools/exp_ms_flight_gpu.py [multiple_shooting|condensed] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca

form = sys.argv[1] if len(sys.argv) > 1 else "multiple_shooting"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
p1, p2 = 67.0, 73.0
obs = np.array([[p1, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[p2, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]])
obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
spec = {"model": "VTOL2D", "radius": 0.6, "v_max": 20.0, "reached_threshold": 1.0, "num_constraints": 10, "mpc_formulation": form}
_start = [float(v) for v in os.environ["START"].split(",")] if os.environ.get("START") else [2.0, 10.0, 20.0]        # START="x,z,speed"
ctl = sca.BatchedTrackingController(np.array([[_start[0], _start[1], 0.0, _start[2], 0.0, 0.0]]), spec, obs=obs7, device="cuda:0", **({"io_dtype": os.environ["IO"]} if os.environ.get("IO") else {}))
ctl.set_waypoints(np.array([[70.0, 10.0], [70.0, 0.5]]) if os.environ.get("START") else np.array([[2.0, 10.0], [70.0, 10.0], [70.0, 0.5]]))
print("io dtype", ctl.tdtype)
print(type(ctl.mpc).__name__)
names = {0: "opt", 1: "infeas", 2: "inacc", 4: "resto"}
sts, t0 = [], time.time()
for k in range(steps):
    ret = int(ctl.control_step(1)[0].item())
    X = ctl.X[0].cpu().numpy(); st = int(ctl.mpc_status[0].item()); it = int(ctl.mpc_iters[0].item())
    nf = getattr(ctl.mpc, "n_fallback", 0)
    sts.append(st)
    if k < 8 or k % 10 == 0 or ret != 0 or os.environ.get("ALL"):
        print(f"step {k:3d} ret {ret:2d} mpc {names.get(st, st):6s} it {it:5d} fb {nf} x {X[0]:7.2f} z {X[1]:6.2f} pitch {np.degrees(X[2]):6.1f} vx {X[3]:6.2f} vz {X[4]:6.2f} u {ctl.u_pos[0].cpu().numpy().round(3)} goal {int(ctl.current_goal_index[0].item())}", flush=True)
    if ret != 0:
        break
sts = np.array(sts)
print("flight ended: ret", ret, "after", len(sts), "control steps; statuses opt/infeas/inacc", [(sts == s).sum() for s in (0, 1, 2)], "%.1f s" % (time.time() - t0))
