"""Launch time of the VTOL2D MPC-CBF wave kernel on the vtol bench batch: one launch with the round-3 limit of 100 iterations, and the
reference solver's budget (3000) as continuation launches.  MI355X only.   python tools/time_mpcvtol.py [B] [f32|f64] [limit100]
(`limit100`: the one-launch configuration only -- counter passes want one kind of dispatch; `ms`: the multiple-shooting kernel of
round 5, csrc/mpc_vtol_ms.hip, one launch with IPOPT's budget)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_control_amd as sca
from safe_control_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
io = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = "cuda:0"
X, up, goal, obs = W.mpc_family_batch("vtol", B, 8, seed=0)
td = torch.float32 if io == "f32" else torch.float64
tt = lambda a: torch.tensor(a, dtype=td, device=dev)
X, up, goal, obs = tt(X), tt(up), tt(goal), tt(obs)
CONFIGS = (("limit 100, one launch", dict(max_iter=100, iter_slices=(), classify_first=False)), ("budget 3000, sliced", dict()))
if len(sys.argv) > 3 and sys.argv[3] == "limit100":
    CONFIGS = CONFIGS[:1]
MS = len(sys.argv) > 3 and sys.argv[3] == "ms"
if MS:
    CONFIGS = (("multiple shooting, budget", dict()),)
for label, kw in CONFIGS:
    ctl = sca.BatchedVtolMSMPCCBF(io_dtype=io, fallback=False) if MS else sca.BatchedVtolMPCCBF(io_dtype=io, **kw)
    u, st, it = ctl.solve(X, up, goal, obs); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        u, st, it = ctl.solve(X, up, goal, obs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    print(f"{label:24s}: {ms:8.2f} ms per {B} problems = {B / ms * 1e3:9.0f} solves/s | optimal {float((st == 0).float().mean()):.4f} "
          f"inaccurate {float((st == 2).float().mean()):.4f} | iterations mean {float(it.float().mean()):.1f} max {int(it.max())}", flush=True)
