"""GPU timing of the streaming CBF-QP kernel (f32 storage, f64 arithmetic) at large batches; SC_EXP_LIB selects a variant
build of the library:   python3 tools/exp_cbfqp_time.py [log2B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_amd import _lib as _L
if os.environ.get("SC_EXP_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["SC_EXP_LIB"])
import safe_control_amd as sca
from safe_control_amd import workloads as W

dev = "cuda:0"
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
ctl = sca.BatchedCBFQP(dict(spec), io_dtype="f32", compute_dtype="f64")
X, goal, ur, obs = W.du_cbfqp_batch(1 << 20, 8, seed=0)
for lb in [int(a) for a in sys.argv[1:]] or [20, 24]:
    B = 1 << lb
    rep = max(1, B >> 20)
    a = torch.tensor(X[:B], dtype=torch.float32, device=dev).repeat(rep, 1)
    b = torch.tensor(ur[:B], dtype=torch.float32, device=dev).repeat(rep, 1)
    c = torch.tensor(obs[:B], dtype=torch.float32, device=dev).repeat(rep, 1, 1)
    out = (torch.empty((B, 2), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
           torch.empty((B, 8), dtype=torch.float32, device=dev))
    for _ in range(3):
        ctl.solve(a, b, c, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ctl.solve(a, b, c, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    print(os.environ.get("SC_EXP_LIB", "default"), "B=2^%d" % lb, "%.1f us" % us, "frac %.3f" % (292.0 * B / (us * 1e-6) / 8e12),
          "checksum", float(out[0].nan_to_num().double().sum()), int((out[1] == 0).sum()))
    del a, b, c, out
