"""GPU: the long closed-loop solves saved by tools/exp_vtol_fleet.py (gpurun_out/vtol_crawlers.npz), solved alone by kernel 12 in f64 and f32
storage, with and without the other saved problems in the batch.   python3 tools/dbg_ms_crawler_gpu.py [file]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_control_amd as sca
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "tools/data/vtol_crawlers.npz")
n = len([k for k in d.files if k.startswith("X_")])
spec = {"model": "VTOL2D", "radius": 0.6, "v_max": 20.0}
X = np.stack([d[f"X_{j}"] for j in range(n)]); up = np.stack([d[f"up_{j}"] for j in range(n)]); g = np.stack([d[f"g_{j}"][:2] for j in range(n)]); ob = np.stack([d[f"ob_{j}"] for j in range(n)])
print("saved: iterations", [int(d[f"it_{j}"]) for j in range(n)], "status", [int(d[f"st_{j}"]) for j in range(n)])
for io, dt in (("f64", torch.float64), ("f32", torch.float32)):
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    ctl = sca.BatchedVtolMSMPCCBF(spec, io_dtype=io, fallback=False)
    u, st, it = ctl.solve(t(X), t(up), t(g), t(ob))
    print(io, "batch : iterations", it.tolist(), "status", st.tolist())
    one = [ctl.solve(t(X[j:j + 1]), t(up[j:j + 1]), t(g[j:j + 1]), t(ob[j:j + 1])) for j in range(n)]
    print(io, "alone : iterations", [int(o[2][0]) for o in one], "status", [int(o[1][0]) for o in one])
