"""GPU: the closed-loop MPC leg of bench.py (4096 DynamicUnicycle2D agents, 14 circles) step by step: status counts, iteration statistics
and the wall time of each control step -- which solves make a step long.   python3 tools/exp_closed_loop_iters.py [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import safe_control_amd as sca

T = int(sys.argv[1]) if len(sys.argv) > 1 else 22
MODE = sys.argv[2] if len(sys.argv) > 2 else "default"          # "plain": one launch per solve (no continuation schedule)
B = 4096
obs = np.array([[2.2, 5.0, 0.2], [3.0, 5.0, 0.2], [4.0, 9.0, 0.3], [1.5, 10.0, 0.5], [9.0, 11.0, 1.0], [7.0, 7.0, 3.0],
                [4.0, 3.5, 1.5], [10.0, 7.3, 0.4], [6.0, 13.0, 0.7], [5.0, 10.0, 0.6], [11.0, 5.0, 0.8],
                [13.5, 11.0, 0.6], [2.0, 7.0, 0.7], [2.0, 8.0, 0.5]])
rng = np.random.default_rng(0)
P = rng.uniform(0.5, 13.5, (4 * B, 2))
clear = (np.hypot(P[:, None, 0] - obs[None, :, 0], P[:, None, 1] - obs[None, :, 1]) - obs[None, :, 2]).min(axis=1) > 0.85
P = P[clear][:B]
X0 = np.column_stack([P, rng.uniform(-np.pi, np.pi, B), rng.uniform(0, 1, B)])
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25, "num_constraints": 8}
ctl = sca.BatchedTrackingController(X0, spec, controller_type={"pos": "mpc_cbf"}, obs=obs, io_dtype="f32", device="cuda:0")
ctl.set_waypoints(np.array([[2.0, 2.0], [2.0, 12.0], [12.0, 12.0], [12.0, 2.0]]))
if MODE == "plain":
    ctl.mpc.iter_slices, ctl.mpc.classify_first = (), False
print("mode", MODE, "max_iter", ctl.mpc.max_iter, "slices", ctl.mpc.iter_slices, "classify", ctl.mpc.classify_first)
for k in range(T):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctl.control_step(1)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
    st, it = ctl.mpc_status.cpu().numpy(), ctl.mpc_iters.cpu().numpy()
    live = (ctl.ret == 0).cpu().numpy()
    raw, trk = ctl._raw_iters.cpu().numpy(), ctl._raw_track.cpu().numpy() != 0
    extra = f"  | launch: track {int(trk.sum())} max it {raw[trk].max() if trk.any() else 0}, other slots max it {raw[~trk].max() if (~trk).any() else 0} mean {raw[~trk].mean() if (~trk).any() else 0:.1f}"
    print(f"step {k:2d}: {ms:7.2f} ms  running {int(live.sum())}  status 0/1/2 = {[int((st == s).sum()) for s in (0, 1, 2)]}  iterations mean {it.mean():.1f} p99 {np.percentile(it, 99):.0f} "
          f"max {it.max()} (status {int(st[np.argmax(it)])})  beyond 100: {int((it > 100).sum())}" + extra, flush=True)
