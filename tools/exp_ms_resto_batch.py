"""GPU + host cores: the problems of the optimal-decay bench batch that need IPOPT's restoration phase (SC_STATUS_NEEDS_RESTO without a
workspace), solved with the in-kernel restoration and by oracle/ms_ipopt.py (KERNEL_PROFILE).   python3 tools/exp_ms_resto_batch.py [B]"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiprocessing import Pool
import safe_control_amd as sca
from safe_control_amd import workloads as W
from oracle import ms_ipopt as MS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Xn, up0, gn, on = W.mpc_family_batch("vtol", B, 8, seed=0)
on = on.copy()
rng = np.random.default_rng(100)
r = rng.uniform(0.8, 1.6, B); d = 10.0 + 20.0 * rng.uniform(size=B); off = rng.uniform(-1.0, 1.0, B)
on[::2, 0, 0], on[::2, 0, 1], on[::2, 0, 2] = (Xn[:, 0] + d + r)[::2], (Xn[:, 1] + off)[::2], r[::2]
mdl = MS.vtol_od_model()


def one(i):
    tr = []
    u, st, it, info = MS.solve(mdl, Xn[i], up0[i], gn[i], on[i], return_info=True, opts=dict(MS.KERNEL_PROFILE), trace=tr)
    return u[:4], st, it, sum(1 for q in tr if q["resto"]), info["status"]


if __name__ == "__main__":
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    raw = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False, restoration=False)
    st0 = raw.solve(t(Xn), t(up0), t(gn), t(on))[2].cpu().numpy()
    idx = np.flatnonzero(st0 == 4)
    ctl = sca.BatchedOptimalDecayVtolMSMPCCBF(io_dtype="f64", fallback=False)
    u, rho, st, it, tr = ctl.solve(t(Xn[idx]), t(up0[idx]), t(gn[idx]), t(on[idx]), want_trace=True)
    u, st, it, tr = u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), tr.cpu().numpy()
    with Pool(min(32, os.cpu_count() or 4)) as p:
        res = p.map(one, [int(i) for i in idx], chunksize=1)
    same_st = same_it = same_u = 0
    for k, (i, q) in enumerate(zip(idx, res)):
        nr = int((tr[k, :it[k] + 1, 7] < 0).sum())
        du = np.abs(u[k] - q[0]).max()
        same_st += st[k] == q[1]; same_it += it[k] == q[2]; same_u += du < 1e-6
        print(f"{i:5d} kernel st {st[k]} it {it[k]:4d} resto {nr:3d} | oracle st {q[1]} ({q[4]}) it {q[2]:4d} resto {q[3]:3d} | du {du:.1e}")
    print(f"{len(idx)} problems: status equal {same_st}, iterations equal {same_it}, same input {same_u}")
