#!/usr/bin/env python3
"""Run the CBF-QP kernel a few times at one configuration (driver for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W
B = int(sys.argv[1]); K = int(sys.argv[2]); io = sys.argv[3]; comp = sys.argv[4]; n = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = "cuda:0"
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
X, goal, ur, obs = W.du_cbfqp_batch(min(B, 1 << 20), K, seed=0)
rep = max(1, B // X.shape[0])
ctl = sca.BatchedCBFQP(dict(spec), io_dtype=io, compute_dtype=comp)
td = ctl.torch_dtype
a = torch.tensor(X, dtype=td, device=dev).repeat(rep, 1)
b = torch.tensor(ur, dtype=td, device=dev).repeat(rep, 1)
c = torch.tensor(obs, dtype=td, device=dev).repeat(rep, 1, 1)
out = (torch.empty((B, 2), dtype=td, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
       torch.empty((B, K), dtype=td, device=dev))
for _ in range(n):
    ctl.solve(a, b, c, out=out)
torch.cuda.synchronize()
