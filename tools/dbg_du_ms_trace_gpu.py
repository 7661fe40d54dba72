import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_mpccbf_ms_gpu as T
import safe_control_amd as sca
from safe_control_amd import workloads as W
n = 384
X, up, goal, obs = (a[:n] for a in W.mpc_family_batch("du", 4096, 8, seed=0))
ctl = sca.BatchedMSMPCCBF(T.SPEC, io_dtype="f64")
u, st, it, plan, trace = ctl.solve(T.t(X), T.t(up), T.t(goal), T.t(obs), want_plan=True, want_trace=True)
torch.cuda.synchronize()
u, st, it, plan, trace = (a.cpu().numpy() for a in (u, st, it, plan, trace))
res = T.oracle_many(X, up, goal, obs)
np.set_printoptions(linewidth=220, precision=4)
worst = []
for i, r in enumerate(res):
    m = min(len(r[3]), it[i] + 1, 12)
    ir = np.flatnonzero(np.signbit(r[3][:, 7]))
    m = min(m, ir[0]) if len(ir) else m
    if m == 0: continue
    rel = np.abs(trace[i, :m, :6] - r[3][:m, :6]) / np.maximum(1e-7, np.abs(r[3][:m, :6]))
    worst.append((rel.max(), i, np.unravel_index(rel.argmax(), rel.shape)))
worst.sort(reverse=True)
for w, i, pos in worst[:3]:
    print("problem", i, "worst", w, "at", pos, "status", st[i], res[i][1], "it", it[i], res[i][2])
    print(trace[i, : it[i] + 1]); print(res[i][3])
