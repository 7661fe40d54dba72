"""GPU + host cores: the check of tests/test_mpc_full_batch_gpu.py::test_no_feasible_plan_for_what_the_kernel_labels_infeasible on a wider
range of bench draws.   python3 tools/exp_phase1_wide.py FAMILY FIRST COUNT [starts]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from safe_control_amd import workloads as W
from _oracle_pool import family_solve_many, phase_one_many
import test_mpc_full_batch_gpu as T

fam, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
starts = int(sys.argv[4]) if len(sys.argv) > 4 else 6
X, up, goal, obs = (a[first:first + count] for a in W.mpc_family_batch(fam, 4096, 8, seed=0))
u, st, it, z = T.gpu_solve(fam, X, up, goal, obs)
o = family_solve_many(fam, X, up, goal, obs)
inf = np.flatnonzero(st == 1)
best = phase_one_many(fam, X[inf], up[inf], goal[inf], obs[inf], z[inf], starts=starts)
found = best >= -1e-7
stall = (o["stalled"][inf] == 1) & (o["st"][inf] == 1)
print(f"{fam} draws {first}..{first + count - 1}: status equal to the oracle's on {np.mean(st == o['st']):.4f}; {len(inf)} labelled infeasible "
      f"({int(stall.sum())} by the stall certificate); inaccurate {np.mean(st == 2):.4f}")
print("  feasible plan found for draws", (first + inf[found]).tolist(), "min g", best[found].tolist(), "stall-certified among them:", (first + inf[found & stall]).tolist())
m = np.sort(best[stall])[::-1][:5] if stall.any() else []
print("  the stall certificates closest to feasibility (best min g):", [float(f"{v:.3g}") for v in m])
