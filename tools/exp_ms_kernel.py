"""GPU: csrc/mpc_vtol_ms.hip against oracle/ms_ipopt.py in its kernel profile (riccati, no SOC, no restoration), with iteration traces.
    python3 tools/exp_ms_kernel.py [n] [seed] [K]"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multiprocessing import Pool
import safe_control_amd as sca
from safe_control_amd import workloads as W
from oracle import ms_ipopt as MS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("vtol", max(n, 64), K, seed=seed))
PROFILE = dict(linear_solver="riccati", max_soc=0, restoration="none")
mdl = MS.vtol_model()


def one(i):
    tr = []
    u, st, it, info = MS.solve(mdl, X[i], up[i], goal[i], obs[i], return_info=True, opts=PROFILE, trace=tr)
    return u, st, it, np.array([[t["E0"], t["dinf"], t["pinf"], t["comp"], t["mu"], t["theta"], t["delta"], t["alpha"]] for t in tr]), np.concatenate([info["X"].reshape(-1), info["U"].reshape(-1)])


if __name__ == "__main__":
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    ctl = sca.BatchedVtolMSMPCCBF(io_dtype="f64", fallback=False)
    t0 = time.time()
    u, st, it, plan, trace = ctl.solve(t(X), t(up), t(goal), t(obs), want_plan=True, want_trace=True)
    torch.cuda.synchronize()
    print("kernel: %.3f s for %d problems" % (time.time() - t0, n))
    u, st, it, plan, trace = u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), plan.cpu().numpy(), trace.cpu().numpy()
    with Pool(min(32, os.cpu_count() or 4)) as p:
        res = p.map(one, range(n))
    bad = 0
    for i, (uo, so, ito, tro, po) in enumerate(res):
        du = np.abs(u[i] - uo).max()
        dp = np.abs(plan[i] - po).max()
        flag = "" if (st[i] == so and it[i] == ito and du < 1e-7) else "  <<<<"
        bad += bool(flag)
        if i < 12 or flag:
            print(f"{i:4d} kernel st {st[i]} it {it[i]:4d} | oracle st {so} it {ito:4d} | du {du:.2e} dplan {dp:.2e}{flag}")
    print(f"status equal {np.mean([st[i] == r[1] for i, r in enumerate(res)]):.4f}  iterations equal {np.mean([it[i] == r[2] for i, r in enumerate(res)]):.4f}  "
          f"max du on equal status {max([np.abs(u[i] - r[0]).max() for i, r in enumerate(res) if st[i] == r[1]] + [0]):.2e}  mismatching {bad}")
    # first problem whose trace parts: show where
    for i, (uo, so, ito, tro, po) in enumerate(res):
        m = min(len(tro), it[i] + 1)
        rel = np.abs(trace[i, :m] - tro[:m]) / np.maximum(1e-12, np.abs(tro[:m]))
        w = np.argwhere(rel > 1e-6)
        if len(w):
            r = w[0][0]
            print(f"problem {i}: traces part at iteration {r} (columns E0 dinf pinf comp mu theta delta alpha)")
            for q in range(max(0, r - 1), min(m, r + 3)):
                print("   k", np.array2string(trace[i, q], precision=6, max_line_width=200))
                print("   o", np.array2string(tro[q], precision=6, max_line_width=200))
            break
    else:
        print("every trace equal to 1e-6 relative")
