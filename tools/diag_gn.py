"""Developer diagnostic: Quad2D / DoubleIntegrator2D MPC-CBF batch of tests/test_mpcgn_gpu.py through whatever library is installed
at safe_control_amd/lib (status, iterations, first inputs).    python3 tools/diag_gn.py [Quad2D|DoubleIntegrator2D]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import safe_control_amd as sca
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_mpcgn_gpu as T

name = sys.argv[1] if len(sys.argv) > 1 else "Quad2D"
N, K, B = 10, 8, 20
mdl = T.MODELS[name]()
rng = np.random.default_rng(N * 10 + K)
X = np.zeros((B, mdl["nx"])); Gl = np.zeros((B, 2)); O = np.zeros((B, K, 7))
for i in range(B):
    X[i], Gl[i], O[i] = T.draw(mdl, rng, K)
up = np.tile(T.u_start(mdl), (B, 1))
ctl = sca.BatchedGnMPCCBF({"model": name}, io_dtype="f64", horizon=N)
u, st, it, z = ctl.solve(T.t(X), T.t(up), T.t(Gl), T.t(O), want_z=True)
torch.cuda.synchronize()
print("st", st.cpu().numpy().tolist())
print("it", it.cpu().numpy().tolist())
print("u0", np.round(u.cpu().numpy()[:4], 6).tolist())
print("z0", np.round(z.cpu().numpy()[0], 5).tolist())
