"""GPU experiment: config-5-like batch (optimal-decay MPC-CBF, N = 20, superellipsoid obstacles) on the DU model."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import safe_control_amd as sca
from safe_control_amd import workloads as W
dev = torch.device("cuda:0")
B, K, N = 4096, 8, 20
Xn, goal, un, on = W.du_cbfqp_batch(B, K, seed=5)
rng = np.random.default_rng(5)
on[:, :, 3] = on[:, :, 2] * rng.uniform(0.6, 1.4, (B, K))        # semi-axes a, b from the circle radius
on[:, :, 4] = rng.choice([4.0, 6.0], (B, K))                      # exponent
on[:, :, 5] = rng.uniform(-np.pi, np.pi, (B, K))                   # orientation
on[:, :, 6] = 1.0                                                  # superellipsoid flag
ctl = sca.BatchedOptimalDecayMPCCBF({"model": "DynamicUnicycle2D", "a_max": 1.0, "w_max": 0.5, "radius": 0.25}, io_dtype="f32", horizon=N)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
X, g, ob = t(Xn), t(goal), t(on); up = torch.zeros((B, 2), dtype=torch.float32, device=dev)
u, rho, st, it = ctl.solve(X, up, g, ob); torch.cuda.synchronize()
t0 = time.perf_counter(); u, rho, st, it = ctl.solve(X, up, g, ob); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"B={B} N={N} K={K} superellipsoids: {1e3*dt:.1f} ms, {B/dt:.0f} solves/s, status {np.bincount(st.cpu().numpy(), minlength=4)}, iters mean {it.double().mean().item():.1f}")
