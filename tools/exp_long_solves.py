"""GPU: the longest solves of a bench leg -- which problems they are, how they end -- written to gpurun_out/long_<name>.npz with their
inputs, so that the numpy oracle can trace them on the CPU afterwards.   python3 tools/exp_long_solves.py c3bf_loop|kb|c3bf|dpcbf|vtol [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import safe_control_amd as sca
from safe_control_amd import workloads as W
import bench

name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
if name.endswith("_loop"):
    model = {"c3bf_loop": "KinematicBicycle2D_C3BF", "dpcbf_loop": "KinematicBicycle2D_DPCBF"}[name]
    X, up, g, ob, G, every = bench.bicycle_loop_states(dev, model)
    ctl = sca.BatchedGnMPCCBF({"model": model, "a_max": 5.0, "radius": 0.3}, io_dtype="f32", horizon=10)
elif name == "vtol":
    X, up, g, ob = (torch.tensor(a, dtype=torch.float32, device=dev) for a in W.mpc_family_batch("vtol", 4096, 8, seed=0))
    ctl = sca.BatchedVtolMPCCBF(io_dtype="f32")
else:
    X, up, g, ob = (torch.tensor(a, dtype=torch.float32, device=dev) for a in W.mpc_family_batch(name, 4096, 8, seed=0))
    ctl = sca.BatchedGnMPCCBF({"model": W.MPC_FAMILIES[name]}, io_dtype="f32", horizon=10)
u, st, it = ctl.solve(X, up, g, ob)
torch.cuda.synchronize()
it_, st_ = it.cpu().numpy(), st.cpu().numpy()
order = np.argsort(-it_)[:n]
print(name, "problems", len(it_), "status 0/1/2", [int((st_ == s).sum()) for s in (0, 1, 2)], "iterations mean %.1f" % it_.mean(),
      "histogram (<=50, 100, 200, 500, 1000, 3000):", np.histogram(it_, bins=[0, 51, 101, 201, 501, 1001, 3001])[0])
print("longest:", [(int(i), int(it_[i]), int(st_[i])) for i in order])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"long_{name}.npz"), idx=order, it=it_[order], st=st_[order], X=X[order].cpu().numpy().astype(np.float64),
         up=up[order].cpu().numpy().astype(np.float64), goal=g[order].cpu().numpy().astype(np.float64), obs=ob[order].cpu().numpy().astype(np.float64),
         u=u[order].cpu().numpy().astype(np.float64))
