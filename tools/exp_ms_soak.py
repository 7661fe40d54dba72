"""GPU: kernel 13 soak -- 32768 fresh draws per robot and seed with the reference solver's budget, every instantiation: launch time, status shares,
longest solve; a launch that does not come back within the timeout of the caller is the failure this looks for.   python tools/exp_ms_soak.py [seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca
from safe_control_amd import workloads as W
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 32768
for fam in ("du", "uni", "si", "di", "kb"):
    name = W.MPC_FAMILIES[fam]
    spec = dict({"model": name}, **({"a_max": 1.0, "w_max": 0.5, "v_max": 1.0, "radius": 0.25} if fam == "du" else {}))
    ctl = sca.BatchedMSMPCCBF(spec, io_dtype="f32")
    for seed in range(10, 10 + seeds):
        X, up, goal, obs = W.mpc_family_batch(fam, B, 8, seed=seed)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
        a = (t(X), t(up), t(goal), t(obs))
        torch.cuda.synchronize(); t0 = time.time()
        u, st, it = ctl.solve(*a)
        torch.cuda.synchronize(); dt = time.time() - t0
        ok = st == 0
        print(f"{fam} seed {seed}: {B} problems in {dt * 1e3:.1f} ms; optimal {ok.double().mean().item():.4f} infeasible {(st == 1).double().mean().item():.4f} "
              f"inaccurate {(st == 2).double().mean().item():.4f}; iterations mean {it.double().mean().item():.1f} max {int(it.max())}; finite inputs on the optimal ones: {bool(torch.isfinite(u[ok]).all())}", flush=True)
