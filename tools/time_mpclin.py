"""GPU timing of the linear-model MPC-CBF kernel at one configuration (SC_EXP_LIB selects a variant build):
    python3 tools/time_mpclin.py MODEL B HORIZON"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_amd import _lib as _L
if os.environ.get("SC_EXP_LIB"):
    _L.LIB_PATH = os.environ["SC_EXP_LIB"]
import safe_control_amd as sca
from safe_control_amd import workloads as W

model, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
ctl = sca.BatchedLinearMPCCBF({"model": model}, io_dtype="f32", horizon=N)
Xn, gn, on = W.linear_mpc_batch(model, B, 8, seed=0)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
X, g, ob = t(Xn), t(gn), t(on)
up = torch.zeros((B, 4 if model == "Quad3D" else 2), dtype=torch.float32, device=dev)
u, st, it = ctl.solve(X, up, g, ob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    u, st, it = ctl.solve(X, up, g, ob)
e1.record()
torch.cuda.synchronize()
print(model, "B", B, "N", N, "ms", e0.elapsed_time(e1) / 3, "optimal", float((st == 0).double().mean()), "iters", float(it.double().mean()))
