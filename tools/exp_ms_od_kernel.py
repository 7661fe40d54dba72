"""GPU: the optimal-decay instantiation of csrc/mpc_vtol_ms.hip against oracle/ms_ipopt.py (vtol_od_model, kernel profile), and against the
condensed optimal-decay kernel on the bench batch.   python3 tools/exp_ms_od_kernel.py [n_parity] [B_bench]"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multiprocessing import Pool
import safe_control_amd as sca
from safe_control_amd import workloads as W
from oracle import ms_ipopt as MS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
Bb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096


def od_batch(B, K=8, seed=0):
    Xn, up0, gn, on = W.mpc_family_batch("vtol", B, K, seed=seed)
    on = on.copy()
    rng = np.random.default_rng(seed + 100)
    r = rng.uniform(0.8, 1.6, B); d = 10.0 + 20.0 * rng.uniform(size=B); off = rng.uniform(-1.0, 1.0, B)
    on[::2, 0, 0], on[::2, 0, 1], on[::2, 0, 2] = (Xn[:, 0] + d + r)[::2], (Xn[:, 1] + off)[::2], r[::2]
    return Xn, up0, gn, on


X, up, goal, obs = od_batch(max(Bb, n))
PROFILE = dict(linear_solver="riccati", max_soc=0, restoration="none")
mdl = MS.vtol_od_model()


def one(i):
    u, st, it, info = MS.solve(mdl, X[i], up[i], goal[i], obs[i], return_info=True, opts=PROFILE)
    return u[:4], st, it, info["U"][:, 4:].reshape(-1)


if __name__ == "__main__":
    t = lambda a, m=None: torch.tensor(np.ascontiguousarray(a[:m]), dtype=torch.float64, device="cuda:0")
    ctl = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False)
    u, rho, st, it = ctl.solve(t(X, n), t(up, n), t(goal, n), t(obs, n))
    torch.cuda.synchronize()
    u, rho, st, it = u.cpu().numpy(), rho.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    with Pool(min(32, os.cpu_count() or 4)) as p:
        res = p.map(one, range(n))
    bad = 0
    for i, (uo, so, ito, ro) in enumerate(res):
        du, dr = np.abs(u[i] - uo).max(), np.abs(rho[i] - ro).max()
        flag = "" if (st[i] == so and abs(int(it[i]) - ito) <= 1 and (so != 0 or du < 1e-6)) else "  <<<<"
        bad += bool(flag)
        if i < 10 or flag:
            print(f"{i:4d} kernel st {st[i]} it {it[i]:4d} | oracle st {so} it {ito:4d} | du {du:.2e} drho {dr:.2e} rho_max {np.abs(ro - 1).max():.3f}{flag}")
    print(f"status equal {np.mean([st[i] == r[1] for i, r in enumerate(res)]):.4f}  iterations equal {np.mean([it[i] == r[2] for i, r in enumerate(res)]):.4f}  mismatching {bad}")
    # bench batch
    for name, c in (("ms", sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False)), ("ms+fallback", sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64")),
                    ("condensed", sca.BatchedOptimalDecayVtolMPCCBF(io_dtype="f64"))):
        args = (t(X, Bb), t(up, Bb), t(goal, Bb), t(obs, Bb))
        c.solve(*args); torch.cuda.synchronize()
        t0 = time.time(); r = c.solve(*args); torch.cuda.synchronize(); dt = time.time() - t0
        s_, i_ = r[2].cpu().numpy(), r[3].cpu().numpy()
        print(f"{name:12s}: {dt * 1e3:8.1f} ms per {Bb}  status 0/1/2/4 = {[int((s_ == s).sum()) for s in (0, 1, 2, 4)]}  iterations mean {i_.mean():.1f} max {i_.max()}  "
              f"decay moved {float(((r[1] - 1).abs().max(dim=1).values > 1e-3).double().mean()):.3f}")
