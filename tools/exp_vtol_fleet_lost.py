"""GPU: which aircraft of bench.py's 256-aircraft VTOL2D fleet (vtol_fleet_closed_loop_leg) are lost, where and when -- starts, return codes,
the step of the return and the states along the way, for tools/exp_vtol_lost_oracle.py (the f64 oracle flown from the same starts).
    python3 tools/exp_vtol_fleet_lost.py [B] [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/vtol_fleet_lost.json"
obs = np.array([[67.0, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[73.0, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]])
obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
rng = np.random.default_rng(0)
X0 = np.zeros((B, 6))
X0[:, 0] = 2.0 + 10.0 * rng.uniform(size=B); X0[:, 1] = 10.0 + rng.uniform(-0.5, 0.5, B); X0[:, 3] = rng.uniform(18.0, 20.0, B)
X0[0] = [2.0, 10.0, 0.0, 20.0, 0.0, 0.0]
spec = {"model": "VTOL2D", "radius": 0.6, "v_max": 20.0, "reached_threshold": 1.0, "num_constraints": 10}
ctl = sca.BatchedTrackingController(X0, spec, obs=obs7, device="cuda:0")
ctl.set_waypoints(np.array([[70.0, 10.0], [70.0, 0.5]]))
done = torch.zeros(B, dtype=torch.int32, device="cuda:0"); when = torch.zeros(B, dtype=torch.int32, device="cuda:0")
traj = []
for n in range(1, 451):
    ret = ctl.control_step(1)
    new = (done == 0) & (ret != 0)
    done = torch.where(new, ret.to(torch.int32), done); when = torch.where(new, torch.full_like(when, n), when)
    traj.append(ctl.X.detach().double().cpu().numpy().copy())
    if bool((done != 0).all()):
        break
done, when, traj = done.cpu().numpy(), when.cpu().numpy(), np.array(traj)
lost = np.nonzero(done != -1)[0]
res = {"aircraft": B, "control_steps": int(n), "landed": int((done == -1).sum()), "lost_indices": lost.tolist(),
       "lost": [{"index": int(i), "start": [float(np.float32(v)) for v in X0[i]], "return_code": int(done[i]), "step": int(when[i]),
                 "state_at_return": traj[min(int(when[i]), len(traj)) - 1, i].tolist(),
                 "states_every_10_steps": traj[:max(1, int(when[i])):10, i].round(3).tolist()} for i in lost],
       "landed_starts_x_range": [float(X0[done == -1, 0].min()), float(X0[done == -1, 0].max())],
       "lost_starts_x": X0[lost, 0].round(3).tolist(), "lost_starts_speed": X0[lost, 3].round(3).tolist()}
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "lost"}))
