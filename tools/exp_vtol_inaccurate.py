"""GPU + host cores: the VTOL2D bench draws that end `optimal_inaccurate` with the reference solver's budget -- does an independent
phase-1 (tests/test_oracle_mpc_resto.py: phase_one on the oracle's problem functions) find a feasible plan for them?
    python3 tools/exp_vtol_inaccurate.py [first] [count] [starts]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import safe_control_amd as sca
from safe_control_amd import workloads as W
from _oracle_pool import phase_one_many

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
starts = int(sys.argv[3]) if len(sys.argv) > 3 else 4
X, up, goal, obs = (a[first:first + count] for a in W.mpc_family_batch("vtol", 4096, 8, seed=0))
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
ctl = sca.BatchedVtolMPCCBF(io_dtype="f64")
u, st, it, z = (a.cpu().numpy() for a in ctl.solve(t(X), t(up), t(goal), t(obs), want_z=True))
bad = np.flatnonzero(st == 2)
print(f"vtol draws {first}..{first + count - 1}: optimal {np.mean(st == 0):.4f} infeasible {np.mean(st == 1):.4f} inaccurate {len(bad)} ({np.mean(st == 2):.4f}); "
      f"their iteration counts {sorted(it[bad].tolist())}")
best = phase_one_many("vtol", X[bad], up[bad], goal[bad], obs[bad], z[bad], starts=starts, timeout=6000)
feas = best >= -1e-7
print(f"  a feasible plan exists for {int(feas.sum())} of them: draws {(first + bad[feas]).tolist()} (iterations {it[bad[feas]].tolist()});")
print(f"  none found for {int((~feas).sum())}: best min g {np.round(np.sort(best[~feas])[::-1][:12], 4).tolist()} ...")
