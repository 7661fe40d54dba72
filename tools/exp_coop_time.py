"""GPU timing of the cooperative CBF-QP kernel inside a hipGraph (the headline launch shape): B agents, K obstacle rows (K <= 8: 8 lanes
per agent, K <= 16: 16).  SC_EXP_LIB selects a variant build.    python3 tools/exp_coop_time.py [B K] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_amd import _lib as _L
if os.environ.get("SC_EXP_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["SC_EXP_LIB"])
import safe_control_amd as sca
from safe_control_amd import workloads as W

dev = "cuda:0"
args = [int(a) for a in sys.argv[1:]] or [4096, 8, 16384, 16]
for B, K in zip(args[0::2], args[1::2]):
    ctl = sca.BatchedCBFQP({"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}, io_dtype="f32", compute_dtype="f64")
    X, goal, ur, obs = W.du_cbfqp_batch(B, K, seed=0)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    a, b, c = t(X), t(ur), t(obs)
    out = (torch.empty((B, 2), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
           torch.empty((B, K), dtype=torch.float32, device=dev))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ctl.solve(a, b, c, out=out)
        g = torch.cuda.CUDAGraph()
        n = 200
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                ctl.solve(a, b, c, out=out)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s)
        torch.cuda.synchronize()
    print(os.environ.get("SC_EXP_LIB", "default"), f"B={B} K={K}", "%.2f us per launch" % (1e3 * e0.elapsed_time(e1) / n),
          "optimal", int((out[1] == 0).sum()), "checksum", float(out[0].nan_to_num().double().sum()))
