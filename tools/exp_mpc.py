"""GPU experiment: MPC-CBF iteration histogram and batch-size scaling (run through gpurun)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import safe_control_amd as sca
from safe_control_amd import _lib as _L
import os
if os.environ.get('SC_EXP_LIB'):            # experiment only: time a variant build of the library
    _L.LIB_PATH = os.environ['SC_EXP_LIB']
from safe_control_amd import workloads as W

dev = torch.device("cuda:0")
spec = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}
NH = int(os.environ.get('SC_EXP_N', 10)); KO = int(os.environ.get('SC_EXP_K', 8))
OD = os.environ.get('SC_EXP_OD') == '1'
ctl = (sca.BatchedOptimalDecayMPCCBF if OD else sca.BatchedMPCCBF)(dict(spec), io_dtype="f32", horizon=NH)
for B in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384, 65536]:
    Xn, goal, un, on = W.du_cbfqp_batch(B, KO, seed=0)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    X, g, ob = t(Xn), t(goal), t(on)
    up = torch.zeros((B, 2), dtype=torch.float32, device=dev)
    r_ = ctl.solve(X, up, g, ob)
    u, st, it = (r_[0], r_[2], r_[3]) if OD else r_[:3]
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ctl.solve(X, up, g, ob)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    itn = it.cpu().numpy()
    print(f"B={B} ms={ms:.3f} solves/s={B/ms*1e3:.0f} iters mean={itn.mean():.2f} max={itn.max()} "
          f"p50={np.percentile(itn,50)} p90={np.percentile(itn,90)} p99={np.percentile(itn,99)} "
          f"status={np.bincount(st.cpu().numpy(), minlength=4)}", flush=True)
    if B == 4096:
        print(" hist", np.bincount(itn // 5)[:25])
        # sorted by iteration count (oracle ordering) to see the tail effect
        order = torch.tensor(np.argsort(-itn, kind="stable"), device=dev)
        Xs, gs, obs_, ups = X[order].contiguous(), g[order].contiguous(), ob[order].contiguous(), up[order].contiguous()
        ctl.solve(Xs, ups, gs, obs_); torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            ctl.solve(Xs, ups, gs, obs_)
        e1.record(); torch.cuda.synchronize()
        print(f" longest-first order: ms={e0.elapsed_time(e1)/3:.3f}")
