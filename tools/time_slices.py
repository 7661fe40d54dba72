"""GPU timing of continuation-launch schedules on the 4096-problem bench batches (f32 storage):
    python3 tools/time_slices.py [family ...]     families: du kb c3bf dpcbf di quad2d quad3d si vtol"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from exp_tail import make as _make, t32, B, K                              # noqa: E402  (runs nothing: exp_tail's loop is under argv)

def timed(ctl, arrs, reps=5):
    X, up, g, ob = [t32(a) for a in arrs]
    out = ctl.solve(X, up, g, ob)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = ctl.solve(X, up, g, ob); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, out[1].cpu().numpy(), out[2].cpu().numpy()

if __name__ == "__main__":
    for fam in (sys.argv[1:] or ["du"]):
        base, arrs = _make(fam, 100)
        ms0, st0, it0 = timed(base, arrs)
        print(f"[{fam}] one launch, limit 100: {ms0:.3f} ms   (it max {it0.max()}, mean {it0.mean():.1f})")
        for label, kw in (("classify first", dict(classify_first=True)),
                          ("caps (100,) limit 3000", dict(iter_slices=(100,), max_iter=3000)),
                          ("classify + caps (100,) limit 3000", dict(classify_first=True, iter_slices=(100,), max_iter=3000)),
                          ("caps (16, 32, 64)", dict(iter_slices=(16, 32, 64))),
                          ("classify + caps (24,)", dict(classify_first=True, iter_slices=(24,))),
                          ("one launch limit 3000", dict(max_iter=3000))):
            ctl, _ = _make(fam, kw.pop("max_iter", 100))
            ctl.init_slices(kw.get("iter_slices"), kw.get("classify_first", False), True)
            ms, st, it = timed(ctl, arrs)
            print(f"    {label:36s} {ms:9.3f} ms   status equal {bool((st == st0).all())}  iters equal {bool((it == it0).all())}  pending {int((st < 0).sum())}  it max {it.max()}")
