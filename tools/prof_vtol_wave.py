"""Cycle counters of the phases of mpcvtol_wave_kernel (a build with -DSC_VTOL_PROF writes them into z_out): where an iteration goes.
  (cd safe_control_amd/csrc && touch mpc_vtol_wave.hip && make EXTRA=-DSC_VTOL_PROF) ; python tools/prof_vtol_wave.py ; rebuild without
MI355X only."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_control_amd import _lib as _L
if os.environ.get("SC_EXP_LIB"):                     # tools/build_variants.sh mpc_vtol_wave VP="-DSC_VTOL_PROF -mllvm -disable-machine-licm"
    _L.LIB_PATH = os.path.abspath(os.environ["SC_EXP_LIB"])
import safe_control_amd as sca
from safe_control_amd import workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
X, up, goal, obs = W.mpc_family_batch("vtol", B, 8, seed=0)
tt = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda:0")
ctl = sca.BatchedVtolMPCCBF(io_dtype="f64", max_iter=100, iter_slices=(), classify_first=False)
u, st, it, z = ctl.solve(tt(X), tt(up), tt(goal), tt(obs), want_z=True); torch.cuda.synchronize()
pr = z.cpu().numpy()[:, :8]; it = it.cpu().numpy()
names = ["eval (main)", "linearise", "adjoint", "stage blocks", "riccati", "lq forward", "rows + rest", "line search trials"]
tot = pr.sum(axis=1)
print(f"{B} problems, iterations mean {it.mean():.1f}; cycles per iteration (mean over problems): {np.mean(tot / it):.0f}")
for i, nme in enumerate(names):
    print(f"  {nme:20s} {np.mean(pr[:, i] / it):10.0f} cycles / iteration   {100 * pr[:, i].sum() / tot.sum():5.1f} %")
