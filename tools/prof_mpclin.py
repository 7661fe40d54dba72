"""A few launches of the linear-model MPC-CBF kernel at one configuration (driver for rocprofv3):
    python3 tools/prof_mpclin.py MODEL B n_launches [horizon]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

model, B, nl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
ctl = sca.BatchedLinearMPCCBF({"model": model}, io_dtype="f32", horizon=N)
Xn, gn, on = W.linear_mpc_batch(model, B, 8, seed=0)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
X, g, ob = t(Xn), t(gn), t(on)
up = torch.zeros((B, 4 if model == "Quad3D" else 2), dtype=torch.float32, device=dev)
for _ in range(nl):
    ctl.solve(X, up, g, ob)
torch.cuda.synchronize()
