"""A few launches of the Manipulator2D CBF-QP kernel (driver for rocprofv3):  python3 tools/prof_manip.py B K n_launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca
B, K, nl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
X = rng.uniform(-np.pi, np.pi, (B, 3)); ur = rng.uniform(-2.5, 2.5, (B, 3))
rho, phi = rng.uniform(0.8, 3.8, (B, K)), rng.uniform(-np.pi, np.pi, (B, K))
obs = np.zeros((B, K, 7)); obs[..., 0], obs[..., 1], obs[..., 2] = rho * np.cos(phi), rho * np.sin(phi), rng.uniform(0.15, 0.5, (B, K))
ctl = sca.BatchedManipulatorCBFQP({"model": "Manipulator2D", "w_max": 2.0, "radius": 0.25}, io_dtype="f32", num_rows=150)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
a = (t(X), t(ur), t(obs))
for _ in range(nl):
    ctl.solve(*a)
torch.cuda.synchronize()
