"""GPU timing of the DynamicUnicycle2D MPC-CBF kernel at one horizon (run-time sizes unless N is 10 or 20; SC_EXP_LIB
selects a variant build):
    python3 tools/time_mpccbf.py B HORIZON"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_amd import _lib as _L
if os.environ.get("SC_EXP_LIB"):
    _L.LIB_PATH = os.environ["SC_EXP_LIB"]
import safe_control_amd as sca
from safe_control_amd import workloads as W

B, N = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
ctl = sca.BatchedMPCCBF({"model": "DynamicUnicycle2D"}, io_dtype="f32", horizon=N)
Xn, gn, _, on = W.du_cbfqp_batch(B, 8, seed=0)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
X, g, ob = t(Xn), t(gn), t(on)
up = torch.zeros((B, 2), dtype=torch.float32, device=dev)
out = ctl.solve(X, up, g, ob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    out = ctl.solve(X, up, g, ob)
e1.record()
torch.cuda.synchronize()
st, it = out[1], out[2]
print("DU MPC B", B, "N", N, "ms", e0.elapsed_time(e1) / 3, "optimal", float((st == 0).double().mean()), "iters", float(it.double().mean()))
