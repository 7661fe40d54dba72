"""Experiment: per-iteration trace (SC_EXP_TRACE builds) of N = 20 MPC problems for two variant libraries."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from safe_control_amd import _lib as _L
_L.LIB_PATH = os.environ["SC_EXP_LIB"]
import safe_control_amd as sca
from safe_control_amd import workloads as W
dev = torch.device("cuda:0")
TD = torch.float32 if os.environ.get("SC_EXP_IO") == "f32" else torch.float64
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
ctl = sca.BatchedMPCCBF(dict(spec), io_dtype=os.environ.get("SC_EXP_IO","f64"), horizon=20)
B = 4
Xn, goal, un, on = W.du_cbfqp_batch(B, 8, seed=0)
t = lambda a: torch.tensor(a, dtype=TD, device=dev)
u, st, it, z = ctl.solve(t(Xn), torch.zeros((B, 2), dtype=TD, device=dev), t(goal), t(on), want_z=True)
torch.cuda.synchronize()
np.set_printoptions(linewidth=200, precision=6)
print("iters", it.cpu().numpy(), "status", st.cpu().numpy())
for b in range(2):
    print(z[b].cpu().numpy().reshape(20, 2).T)
