"""GPU: the register CBF-QP kernel at 2^24 agents (f32 storage; f64 and f32 arithmetic): time per launch and fraction of the 8 TB/s HBM peak.
   python tools/time_cbfqp_big.py [log2 B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B, K, dev = 1 << lg, 8, "cuda:0"
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
X, goal, ur, obs = W.du_cbfqp_batch(1 << 20, K, seed=0)
rep = B >> 20
for comp in ("f64", "f32"):
    ctl = sca.BatchedCBFQP(dict(spec), io_dtype="f32", compute_dtype=comp)
    a = torch.tensor(X, dtype=torch.float32, device=dev).repeat(rep, 1); b = torch.tensor(ur, dtype=torch.float32, device=dev).repeat(rep, 1)
    c = torch.tensor(obs, dtype=torch.float32, device=dev).repeat(rep, 1, 1)
    out = (torch.empty((B, 2), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev), torch.empty((B, K), dtype=torch.float32, device=dev))
    for _ in range(3): ctl.solve(a, b, c, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ctl.solve(a, b, c, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    import hashlib
    hh = hashlib.sha256(); [hh.update(o.cpu().numpy().tobytes()) for o in out]
    print("  outputs sha", hh.hexdigest()[:16], "stream_min", os.environ.get("SC_CBFQP_STREAM_MIN", "default"))
    print(f"2^{lg} agents, f32 storage, {comp} arithmetic: {ms:.3f} ms = {B / ms / 1e6:.2f} G solves/s = {292 * B / ms / 1e9:.2f} TB/s = {292 * B / ms / 1e9 / 8:.3f} of the HBM peak")
    del a, b, c, out
