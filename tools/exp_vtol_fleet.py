"""GPU: the reference's VTOL2D example scene flown by B aircraft at once from perturbed starts (bench.py: vtol_fleet_closed_loop_leg), with
per-step solver statistics.   python3 tools/exp_vtol_fleet.py [B] [steps] [formulation]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 450
form = sys.argv[3] if len(sys.argv) > 3 else "multiple_shooting"
obs = np.array([[67.0, z, 0.5] for z in (6.0, 7.0, 8.0, 9.0)] + [[73.0, float(z), 0.5] for z in range(1, 16)] + [[60.0, 12.0, 1.5]])
obs7 = np.hstack([obs, np.zeros((len(obs), 4))])
spec = {"model": "VTOL2D", "radius": 0.6, "v_max": 20.0, "reached_threshold": 1.0, "num_constraints": 10, "mpc_formulation": form}
rng = np.random.default_rng(0)
X0 = np.zeros((B, 6))
X0[:, 0] = 2.0 + 10.0 * rng.uniform(size=B); X0[:, 1] = 10.0 + rng.uniform(-0.5, 0.5, B); X0[:, 3] = rng.uniform(18.0, 20.0, B)
X0[0] = [2.0, 10.0, 0.0, 20.0, 0.0, 0.0]
ctl = sca.BatchedTrackingController(X0, spec, obs=obs7, device="cuda:0")
ctl.set_waypoints(np.array([[70.0, 10.0], [70.0, 0.5]]))
done = torch.zeros(B, dtype=torch.int32, device="cuda:0"); when = torch.zeros(B, dtype=torch.int32, device="cuda:0")
# keep the inputs of solves that run to the iteration limit (replayed on the CPU: tools/dbg_ms_crawler.py)
_solve, crawlers = ctl.mpc.solve, []
def solve(X, up, g, ob, **kw):
    r = _solve(X, up, g, ob, **kw)
    bad = torch.nonzero(r[2] >= 1000).flatten()
    for i in bad[:2].tolist():
        if len(crawlers) < 8:
            crawlers.append(dict(X=X[i].cpu().numpy(), up=up[i].cpu().numpy(), g=g[i].cpu().numpy(), ob=ob[i].cpu().numpy(), it=int(r[2][i]), st=int(r[1][i]), u=r[0][i].cpu().numpy()))
    return r
ctl.mpc.solve = solve
t0 = time.time()
for n in range(1, steps + 1):
    ts = time.time()
    ret = ctl.control_step(1)
    torch.cuda.synchronize()
    new = (done == 0) & (ret != 0)
    done = torch.where(new, ret.to(torch.int32), done); when = torch.where(new, torch.full_like(when, n), when)
    st, it = ctl.mpc_status.cpu().numpy(), ctl.mpc_iters.cpu().numpy()
    live = (done == 0).cpu().numpy()
    if n <= 5 or n % 25 == 0 or not live.any():
        X = ctl.X.cpu().numpy()
        print(f"step {n:3d} {1e3 * (time.time() - ts):7.1f} ms live {int(live.sum()):4d} statuses 0/1/2 {[(st[live] == s).sum() for s in (0, 1, 2)]} iters max {it[live].max() if live.any() else 0} "
              f"x {X[live, 0].min() if live.any() else 0:6.1f}..{X[live, 0].max() if live.any() else 0:6.1f} z {X[live, 1].min() if live.any() else 0:5.1f}..{X[live, 1].max() if live.any() else 0:5.1f} goal idx {np.bincount(ctl.current_goal_index.cpu().numpy(), minlength=3).tolist()}", flush=True)
    if not live.any():
        break
d = done.cpu().numpy(); w = when.cpu().numpy()
print(f"{B} aircraft, {n} steps, {time.time() - t0:.1f} s: landed {int((d == -1).sum())}, lost {int((d == -2).sum())}, still flying {int((d == 0).sum())}; return steps {np.sort(w[d != 0])[:5]} .. {np.sort(w[d != 0])[-5:]}")
X = ctl.X.cpu().numpy()
for i in np.flatnonzero(d == 0)[:6]:
    print("  flying", i, "start", X0[i, [0, 1, 3]].round(2), "now", X[i].round(2), "goal", int(ctl.current_goal_index[i].item()))
for i in np.flatnonzero(d == -2)[:6]:
    print("  lost  ", i, "start", X0[i, [0, 1, 3]].round(2), "at step", w[i], "state", X[i].round(2))

if crawlers:
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/vtol_crawlers.npz", **{f"{k}_{j}": v for j, c in enumerate(crawlers) for k, v in c.items()})
    print("saved", len(crawlers), "long solves:", [(c["it"], c["st"]) for c in crawlers])
