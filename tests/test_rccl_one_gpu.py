"""GPU: the RCCL path on the one GPU the box has.  A child `torch.distributed.run --nproc-per-node 1` (started before this process touches
the GPU API in any way that matters to it: it is a separate process) initialises the process group with backend `nccl` -- which IS RCCL on
ROCm --, runs the collective of the design on device tensors (`NeighborExchange.step`: all_gather_into_tensor of the agent states, then the
neighbour kernels), `sharding.max_over_ranks` (all_reduce MAX on a device tensor), `gather_agents` / `broadcast_obstacle_table`, and one
step of BASELINE configs[3] (gather -> neighbours -> C3BF CBF-QP); the results are compared with the same computation without a process
group.  World size one moves no bytes over xGMI, but it does go through RCCL's communicator setup and its device-tensor code paths, which
the gloo tests on CPU tensors (tests/test_sharding_gloo.py) cannot see.  No scaling curve is claimed from this."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
import safe_control_amd as sca
from safe_control_amd import sharding, workloads as W
n, K = 4096, 16
spec = {"model": "KinematicBicycle2D_C3BF", "a_max": 5.0, "radius": 0.3}
Xn, goal, un, on = W.kb_c3bf_batch(n, K, seed=0, spec=spec)
Xn[:, :2] *= 10.0
dev = torch.device("cuda", 0)
X = torch.tensor(Xn, dtype=torch.float32, device=dev); ur = torch.tensor(un, dtype=torch.float32, device=dev)
ex = sharding.NeighborExchange(n, K, 0.3, nx=4, dtype=torch.float32, device=dev)
assert ex.ws == 1
# force the collective even at world size one (NeighborExchange.gather copies when there is one rank)
out = torch.empty_like(ex.X_all)
dist.all_gather_into_tensor(out, X)
assert torch.equal(out, X)
# ... and the other two collectives sharding.py issues when there are several ranks, on device tensors: all_reduce MAX (max_over_ranks), broadcast
tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
assert tt.item() == 1.25
tb = torch.tensor(on[0], dtype=torch.float32, device=dev)
dist.broadcast(tb, src=0)
assert torch.equal(tb, torch.tensor(on[0], dtype=torch.float32, device=dev))
obs = ex.step(X).clone()
ctl = sca.BatchedCBFQP(dict(spec), dt=0.05, io_dtype="f32", compute_dtype="f64")
u, st, h = ctl.solve(X, ur, obs)
t = sharding.max_over_ranks(1.25, device=dev)
tab = sharding.broadcast_obstacle_table(torch.tensor(on[0], dtype=torch.float32, device=dev))
g = sharding.gather_agents(u, n)
dist.barrier()
torch.cuda.synchronize()
np.savez(sys.argv[2], obs=obs.cpu().numpy(), u=u.cpu().numpy(), st=st.cpu().numpy(), t=np.array([t]), tab=tab.cpu().numpy(),
         g=(g.cpu().numpy() if g is not None else np.zeros(0)))
dist.destroy_process_group()
"""


def test_world_size_one_nccl_group_runs_the_exchange_and_the_config4_step(tmp_path):
    script, out = tmp_path / "child.py", tmp_path / "out.npz"
    script.write_text(CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script), ROOT, str(out)], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    # the same step with no process group in this process
    import safe_control_amd as sca
    from safe_control_amd import sharding, workloads as W
    n, K = 4096, 16
    spec = {"model": "KinematicBicycle2D_C3BF", "a_max": 5.0, "radius": 0.3}
    Xn, goal, un, on = W.kb_c3bf_batch(n, K, seed=0, spec=spec)
    Xn[:, :2] *= 10.0
    X = torch.tensor(Xn, dtype=torch.float32, device="cuda:0"); ur = torch.tensor(un, dtype=torch.float32, device="cuda:0")
    obs = sharding.neighbor_obstacles(X, n, K, 0.3)
    u, st, h = sca.BatchedCBFQP(dict(spec), dt=0.05, io_dtype="f32", compute_dtype="f64").solve(X, ur, obs)
    assert np.array_equal(got["obs"], obs.cpu().numpy()) and np.array_equal(got["u"], u.cpu().numpy(), equal_nan=True) and np.array_equal(got["st"], st.cpu().numpy())
    assert got["t"][0] == 1.25 and np.array_equal(got["tab"], on[0].astype(np.float32)) and np.array_equal(got["g"], u.cpu().numpy(), equal_nan=True)
