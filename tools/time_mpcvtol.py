"""Launch time of the VTOL2D MPC-CBF kernel on the 4096-problem vtol batch for several lanes-per-block settings (SC_VTOL_LANES).
MI355X only.   python tools/time_mpcvtol.py [B]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_control_amd as sca
from safe_control_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
X, up, goal, obs = W.mpc_family_batch("vtol", B, 8, seed=0)
tt = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
X, up, goal, obs = tt(X), tt(up), tt(goal), tt(obs)
ctl = sca.BatchedVtolMPCCBF(io_dtype="f64")
for lanes in (os.environ.get("LANES", "64,32,16,8,4").split(",")):
    os.environ["SC_VTOL_LANES"] = lanes
    u, st, it = ctl.solve(X, up, goal, obs); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        u, st, it = ctl.solve(X, up, goal, obs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"lanes {lanes:>2s}: {ms:8.2f} ms per {B} problems = {B / ms * 1e3:9.0f} solves/s | optimal {float((st == 0).float().mean()):.4f} "
          f"infeasible {float((st == 1).float().mean()):.4f} | iterations mean {float(it.float().mean()):.1f} max {int(it.max())}", flush=True)
