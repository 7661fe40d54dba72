"""GPU: the config-4 neighbour search (16384 agents, K = 16): the uniform-grid cell list (sc_neighbor_obstacles_batch_ws) against the plain scan
(sc_neighbor_obstacles_batch), and the whole kb_c3bf step.   python tools/time_neighbors.py [B] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from safe_control_amd import _lib, workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
X = W.kb_c3bf_batch(B, K, seed=0)[0]
tX = torch.tensor(X, dtype=torch.float32, device="cuda:0")
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
a = torch.empty((B, K, 7), dtype=torch.float32, device="cuda:0"); b = torch.empty_like(a)
nbytes = int(lib.sc_neighbor_workspace_bytes(0, B, B, K))
ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda:0")
def run_scan(): assert lib.sc_neighbor_obstacles_batch(0, B, 0, B, K, 0.3, tX.data_ptr(), a.data_ptr(), stream) == 0
def run_cells(): assert lib.sc_neighbor_obstacles_batch_ws(0, B, 0, B, K, 0.3, tX.data_ptr(), b.data_ptr(), ws.data_ptr(), nbytes, stream) == 0
for name, f in (("scan", run_scan), ("cell list", run_cells)):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {B} agents, K = {K}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
print("equal:", torch.equal(a, b))
