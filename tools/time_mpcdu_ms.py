"""GPU: launch time of the multiple-shooting DynamicUnicycle2D kernel (csrc/mpc_du_ms.hip, kernel 13) on the 4096 configs[2] problems, next to the
condensed kernel on the same batch.   python tools/time_mpcdu_ms.py [B] [f32|f64] [reps] [du|di|kb|uni|si] [max_iter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
io = sys.argv[2] if len(sys.argv) > 2 else "f32"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
fam = sys.argv[4] if len(sys.argv) > 4 else "du"
SPEC = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25} if fam == "du" else {"model": W.MPC_FAMILIES[fam]}
MAXIT = int(sys.argv[5]) if len(sys.argv) > 5 else None
dt = torch.float32 if io == "f32" else torch.float64
X, up, goal, obs = W.mpc_family_batch(fam, B, 8, seed=0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
args = (t(X), t(up), t(goal), t(obs))
for name, ctl in (("multiple shooting (kernel 13)", sca.BatchedMSMPCCBF(SPEC, io_dtype=io, check_circles=False, max_iter=MAXIT)), ("condensed (kernel 3)" if fam in ("du", "uni") else ("condensed (mpclin)" if fam == "si" else "condensed (mpcgn)"),
                   sca.BatchedMPCCBF(SPEC, io_dtype=io) if fam in ("du", "uni") else (sca.BatchedLinearMPCCBF(SPEC, io_dtype=io) if fam == "si" else sca.BatchedGnMPCCBF(SPEC, io_dtype=io)))):
    u, st, it = ctl.solve(*args)[:3]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); ctl.solve(*args); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    itf = it.double()
    print(f"{name}: {B} problems {io}: {min(ts):.3f} ms (median {np.median(ts):.3f}) = {B / min(ts) * 1e3:.0f} solves/s; optimal {(st == 0).double().mean().item():.4f} "
          f"infeasible {(st == 1).double().mean().item():.4f} inaccurate {(st == 2).double().mean().item():.4f}; iterations mean {itf.mean().item():.1f} max {int(it.max())} sum {int(it.sum())}")
