"""GPU: launch time of csrc/mpc_vtol_ms.hip on the VTOL2D bench batch, next to the condensed kernel.
    python3 tools/exp_ms_bench.py [B] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
X, up, goal, obs = W.mpc_family_batch("vtol", B, K, seed=0)
for io in ("f64", "f32"):
    dt = torch.float64 if io == "f64" else torch.float32
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    args = (t(X), t(up), t(goal), t(obs))
    for name, ctl in (("ms", sca.BatchedVtolMSMPCCBF(io_dtype=io, fallback=False)), ("ms+fallback", sca.BatchedVtolMSMPCCBF(io_dtype=io)),
                      ("condensed", sca.BatchedVtolMPCCBF(io_dtype=io))):
        ctl.solve(*args); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.time(); r = ctl.solve(*args); torch.cuda.synchronize(); ts.append(time.time() - t0)
        st, it = r[1].cpu().numpy(), r[2].cpu().numpy()
        print(f"{io} {name:12s}: {min(ts) * 1e3:8.2f} ms per {B}  ({B / min(ts):9.0f} solves/s)  status 0/1/2/4 = {[(st == s).sum() for s in (0, 1, 2, 4)]}  iterations mean {it.mean():.1f} max {it.max()}")
