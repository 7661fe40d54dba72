"""As tools/exp_cbfqp_time.py with f32 arithmetic (the variant whose bound is HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W
dev = "cuda:0"
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
ctl = sca.BatchedCBFQP(dict(spec), io_dtype="f32", compute_dtype="f32")
X, goal, ur, obs = W.du_cbfqp_batch(1 << 20, 8, seed=0)
for lb in [int(a) for a in sys.argv[1:]] or [20, 24]:
    B = 1 << lb
    rep = max(1, B >> 20)
    a = torch.tensor(X[:B], dtype=torch.float32, device=dev).repeat(rep, 1)
    b = torch.tensor(ur[:B], dtype=torch.float32, device=dev).repeat(rep, 1)
    c = torch.tensor(obs[:B], dtype=torch.float32, device=dev).repeat(rep, 1, 1)
    out = (torch.empty((B, 2), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev), torch.empty((B, 8), dtype=torch.float32, device=dev))
    for _ in range(3):
        ctl.solve(a, b, c, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctl.solve(a, b, c, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    print("f32 arithmetic", os.environ.get("SC_CBFQP_STREAM", "stream"), "B=2^%d" % lb, "%.1f us" % us, "frac %.3f" % (292.0 * B / (us * 1e-6) / 8e12), "checksum", float(out[0].nan_to_num().double().sum()))
    del a, b, c, out
