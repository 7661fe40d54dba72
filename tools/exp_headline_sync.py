"""Where do the microseconds of a SHORT headline run go?  bench.py's timed region at --steps 20 reports ~4.5-5 us per step against
3.1 at --steps 200.  Times the same 4096-agent hipGraph with (a) torch.cuda.synchronize() as the completion wait, (b) a host
spin on the closing event before it, (c) each of them right after 50 ms of GPU work (clock state).  MI355X only.
  python tools/exp_headline_sync.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_control_amd as sca
from safe_control_amd import workloads as W

dev = torch.device("cuda:0")
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
ctl = sca.BatchedCBFQP(dict(spec), dt=0.05, io_dtype="f32", compute_dtype="f64")
B, K = 4096, 8
Xn, goal, un, on = W.du_cbfqp_batch(B, K, seed=0)
X, ur, ob = (torch.tensor(v, dtype=torch.float32, device=dev) for v in (Xn, un, on))
out = (torch.empty((B, 2), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
       torch.empty((B, K), dtype=torch.float32, device=dev))


def make_graph(steps):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        ctl.solve(X, ur, ob, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(steps):
                ctl.solve(X, ur, ob, out=out)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    return g


def timed(g, steps, spin, busy):
    if busy:
        a = torch.randn(4096, 4096, device=dev)
        for _ in range(busy):
            a = a @ a * 1e-4
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(); g.replay(); e1.record()
    if spin:
        while not e1.query():
            pass
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return 1e6 * (t1 - t0) / steps, 1e3 * e0.elapsed_time(e1) / steps


for steps in (20, 200):
    g = make_graph(steps)
    for spin in (False, True):
        for busy in (0, 20):
            r = [timed(g, steps, spin, busy) for _ in range(7)]
            host = sorted(x[0] for x in r); ev = sorted(x[1] for x in r)
            print(f"steps {steps:3d} spin {int(spin)} busy {busy:2d}: host first {r[0][0]:.2f} median {host[3]:.2f} min {host[0]:.2f} us/step | events first {r[0][1]:.2f} median {ev[3]:.2f} min {ev[0]:.2f}", flush=True)
