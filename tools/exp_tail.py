"""GPU experiment: what bounds a 4096-problem interior-point launch -- the work or its slowest problem -- and what a better launch
order could buy.  For each family: iteration histogram at the shipped limit and at max_iter = 3000, then the launch time of
  original order | longest first (the best any scheduler can do with this kernel) | shortest first | only the problems above p90.
    python3 tools/exp_tail.py [family ...]      families: du kb c3bf dpcbf di quad2d quad3d si vtol"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

dev = torch.device("cuda:0")
B, K = 4096, 8
t32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def make(fam, max_iter):
    if fam == "du":
        ctl = sca.BatchedMPCCBF({"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}, io_dtype="f32", horizon=10,
                                max_iter=max_iter)
        Xn, gn, _, on = W.du_cbfqp_batch(B, K, seed=0)
        return ctl, (Xn, np.zeros((B, 2)), gn, on)
    if fam in ("quad3d", "si"):
        model = {"quad3d": "Quad3D", "si": "SingleIntegrator2D"}[fam]
        ctl = sca.BatchedLinearMPCCBF({"model": model}, io_dtype="f32", horizon=10, max_iter=max_iter)
        Xn, gn, on = W.linear_mpc_batch(model, B, K, seed=0)
        return ctl, (Xn, np.zeros((B, 4 if fam == "quad3d" else 2)), gn, on)
    if fam == "vtol":
        ctl = sca.BatchedVtolMPCCBF(io_dtype="f32", max_iter=max_iter)
        Xn, up0, gn, on = W.mpc_family_batch("vtol", B, K, seed=0)
        return ctl, (Xn, up0, gn, on)
    model = W.MPC_FAMILIES[fam]
    ctl = sca.BatchedGnMPCCBF({"model": model}, io_dtype="f32", horizon=10, max_iter=max_iter)
    Xn, up0, gn, on = W.mpc_family_batch(fam, B, K, seed=0)
    return ctl, (Xn, up0, gn, on)


def run(ctl, arrs, idx=None, reps=3):
    X, up, g, ob = [t32(a if idx is None else a[idx]) for a in arrs]
    out = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = ctl.solve(X, up, g, ob)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out[1].cpu().numpy(), out[2].cpu().numpy()


for fam in ((sys.argv[1:] or ["du", "kb", "c3bf", "vtol", "quad3d"]) if __name__ == "__main__" else []):
    try:
        ctl, arrs = make(fam, 100)
    except TypeError as e:
        print(fam, "constructor has no max_iter:", e)
        continue
    ms, st, it = run(ctl, arrs)
    pct = lambda a: [float(np.percentile(a, q)) for q in (50, 90, 99)] + [int(a.max())]
    print(f"[{fam}] limit 100: {ms:.3f} ms  status 0/1/2 = {[(st == s).mean().round(4) for s in (0, 1, 2)]}  it mean {it.mean():.1f} p50/p90/p99/max {pct(it)}")
    desc = np.argsort(-it, kind="stable")
    for name, idx in (("longest first", desc), ("shortest first", desc[::-1].copy()), ("above p90 only", np.nonzero(it > np.percentile(it, 90))[0]),
                      ("above p99 only", np.nonzero(it > np.percentile(it, 99))[0])):
        if len(idx) == 0:
            continue
        ms_, _, it_ = run(ctl, arrs, idx)
        print(f"    {name:16s} {len(idx):5d} problems  {ms_:8.3f} ms   (work-only bound: sum it / slots; max it {it_.max()})")
    ctl3, _ = make(fam, 3000)
    ms3, st3, it3 = run(ctl3, arrs, reps=1)
    ch = np.nonzero(st3 != st)[0]
    print(f"    limit 3000: {ms3:.3f} ms  status 0/1/2 = {[(st3 == s).mean().round(4) for s in (0, 1, 2)]}  it mean {it3.mean():.1f} p50/p90/p99/max {pct(it3)}"
          f"  status changed on {len(ch)} problems; its of those: {sorted(it3[ch].tolist())[:40]}")
    still = np.nonzero(st3 == 2)[0]
    print(f"    still inaccurate at 3000: {len(still)}  its: {sorted(it3[still].tolist())[:60]}")
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez(f"gpurun_out/tail_{fam}.npz", st=st, it=it, st3=st3, it3=it3)
