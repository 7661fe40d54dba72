#!/usr/bin/env python3
"""Kernel-time explorer for the CBF-QP kernel: compute dtype x batch size (GPU box only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

dev = "cuda:0"
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
Ks = [int(k) for k in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8"])]
for K in Ks:
    for B in (4096, 65536, 1 << 20, 1 << 23):
        X, goal, ur, obs = W.du_cbfqp_batch(min(B, 1 << 20), K, seed=0)
        rep = B // X.shape[0]
        tX = torch.tensor(X, dtype=torch.float32, device=dev).repeat(rep, 1)
        tu = torch.tensor(ur, dtype=torch.float32, device=dev).repeat(rep, 1)
        to = torch.tensor(obs, dtype=torch.float32, device=dev).repeat(rep, 1, 1)
        for io, comp in (("f32", "f32"), ("f32", "f64"), ("f64", "f64")):
            ctl = sca.BatchedCBFQP(dict(spec), io_dtype=io, compute_dtype=comp)
            td = ctl.torch_dtype
            a, b, c = tX.to(td), tu.to(td), to.to(td)
            out = (torch.empty((B, 2), dtype=td, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
                   torch.empty((B, K), dtype=td, device=dev))
            for _ in range(3):
                ctl.solve(a, b, c, out=out)
            torch.cuda.synchronize()
            n = 50 if B <= 65536 else 10
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                ctl.solve(a, b, c, out=out)
                with torch.cuda.graph(g, stream=s):
                    for _ in range(n):
                        ctl.solve(a, b, c, out=out)
            torch.cuda.synchronize()
            g.replay(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / n
            es = 4 if io == "f32" else 8
            nbytes = ((6 + 7 * K) * es + (2 + K) * es + 4) * B
            print(f"K={K:2d} B={B:8d} io={io} comp={comp}: {us:9.2f} us  {B/us/1e3:8.2f} Gsolves/s  {nbytes/us/1e3:8.1f} GB/s", flush=True)
