"""GPU timing of the MPC kernels with and without the restoration phase (max_entries = 0) on the bench batches, and the
iteration-count distribution.   python tools/exp_resto_time.py [family ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_control_amd as sca
from safe_control_amd import _lib, workloads as W

def run(fam, resto):
    name = W.MPC_FAMILIES[fam]
    X, up, goal, obs = W.mpc_family_batch(fam, 4096, 8, 0)
    cls = sca.BatchedMPCCBF if fam == "du" else (sca.BatchedLinearMPCCBF if fam in ("si", "quad3d") else sca.BatchedGnMPCCBF)
    spec = {"model": name, "a_max": 1.0, "w_max": 0.5, "radius": 0.25} if fam == "du" else {"model": name}   # bench.py: mpc_leg
    ctl = cls(spec, io_dtype="f32", horizon=10)
    ctl.resto = resto
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    a = (t(X), t(up), t(goal), t(obs))
    u, st, it = ctl.solve(*a)[:3]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctl.solve(*a)
    e1.record(); torch.cuda.synchronize()
    it = it.cpu().numpy(); st = st.cpu().numpy()
    return e0.elapsed_time(e1) / 5, it, st

for fam in (sys.argv[1:] or ["du", "quad3d", "quad2d", "kb"]):
    for label, r in (("resto", _lib.default_resto()), ("off", _lib.default_resto(max_entries=0)), ("no-handover", _lib.default_resto(small_iter=1000))):
        ms, it, st = run(fam, r)
        srt = np.sort(it)[::-1]
        print(f"{fam:7s} {label:12s} {ms:7.3f} ms  iters mean {it.mean():.1f} p99 {np.percentile(it, 99):.0f} max {it.max()} top8 {srt[:8]}  status {np.bincount(st, minlength=3) / len(st)}")
