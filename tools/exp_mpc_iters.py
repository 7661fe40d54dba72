"""GPU experiment: iteration counts of the config-3 batch against cheap start-of-solve features (is an
longest-first launch order predictable?).  Writes gpurun_out/mpc_iters.npz.
    python3 tools/exp_mpc_iters.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ctl = sca.BatchedMPCCBF({"model": "DynamicUnicycle2D"}, io_dtype="f32", horizon=10)
Xn, gn, _, on = W.du_cbfqp_batch(B, 8, seed=0)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
X, g, ob = t(Xn), t(gn), t(on)
up = torch.zeros((B, 2), dtype=torch.float32, device=dev)
u, st, it = ctl.solve(X, up, g, ob)
torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/mpc_iters.npz", X=Xn, goal=gn, obs=on, st=st.cpu().numpy(), it=it.cpu().numpy(), u=u.cpu().numpy())
itn = it.cpu().numpy()
print("iters: mean", itn.mean(), "p50", np.percentile(itn, 50), "p90", np.percentile(itn, 90), "p99", np.percentile(itn, 99), "max", itn.max())
