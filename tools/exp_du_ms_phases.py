"""GPU: cycles per phase of kernel 13's iteration (a -DSC_DUMS_PROF build of csrc/mpc_du_ms.hip: SC_EXP_LIB / SAFE_CONTROL_AMD_LIB points at it), lone
waves (256 problems = one per CU) and a full machine (4096).   SAFE_CONTROL_AMD_LIB=exp_libs/lib_prof.so python tools/exp_du_ms_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

SPEC = {"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25}
NAMES = ["eval2 (errors)", "errors + mu", "eval2 (build)", "riccati backward", "finish_step", "step lengths + barrier", "line search", "update"]
for B in (256, 4096):
    X, up, goal, obs = W.mpc_family_batch("du", B, 8, seed=0)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    ctl = sca.BatchedMSMPCCBF(SPEC, io_dtype="f64", check_circles=False, max_iter=200)
    u, st, it, tr = ctl.solve(t(X), t(up), t(goal), t(obs), want_trace=True)
    torch.cuda.synchronize()
    prof = tr[:, 200, :].cpu().numpy(); its = it.cpu().numpy().astype(float)
    per = prof.sum(axis=0) / its.sum()
    print(f"B = {B}: {its.sum():.0f} iterations; cycles per iteration {per.sum():.0f}")
    for n, v in zip(NAMES, per):
        print(f"   {n:24s} {v:9.0f}  {100 * v / per.sum():5.1f} %")
    j = int(its.argmax())
    pj = prof[j] / its[j]
    print(f"   the longest solve (problem {j}, {its[j]:.0f} iterations, status {int(st[j])}): {pj.sum():.0f} cycles per iteration = {pj.sum() * its[j] / 2.4e6:.2f} ms at 2.4 GHz: " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, pj)))
