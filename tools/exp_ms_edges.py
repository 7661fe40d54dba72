import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import safe_control_amd as sca
from safe_control_amd import workloads as W
X, up, g, ob = W.mpc_family_batch("vtol", 64, 16, seed=1)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
for K in (0, 1, 9, 16):
    for cls in (sca.BatchedVtolMSMPCCBF, sca.BatchedOptimalDecayVtolMSMPCCBF):
        c = cls(io_dtype="f64", fallback=False)
        o = torch.zeros((64, K, 7), dtype=torch.float64, device="cuda:0") if K == 0 else t(ob[:, :K])
        try:
            r = c.solve(t(X), t(up), t(g), o)
            st = r[-2].cpu().numpy(); it = r[-1].cpu().numpy()
            print(cls.__name__, "K", K, "status", np.bincount(st, minlength=5).tolist(), "iters mean", it.mean(), "max", it.max())
        except Exception as e:
            print(cls.__name__, "K", K, "ERR", repr(e)[:150])
c = sca.BatchedVtolMSMPCCBF(io_dtype="f64", fallback=False)
r = c.solve(t(X[:0]), t(up[:0]), t(g[:0]), t(ob[:0, :8]))
print("B=0", [tuple(a.shape) for a in r])
r = c.solve(t(X[:1]), t(up[:1]), t(g[:1]), t(ob[:1, :8]))
print("B=1", r[1].tolist(), r[2].tolist())
