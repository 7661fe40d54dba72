"""GPU experiment: the two halves of the BASELINE configs[4] fleet (bench.py --workload hetero_fleet) one at a time and together --
how much of the fleet step is overlap.    python3 tools/exp_fleet_solo.py [n_total]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_amd as sca
from safe_control_amd import workloads as W

dev = torch.device("cuda:0")
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N, K = 20, 8
uspec = {"model": "Unicycle2D", "v_max": 1.0, "w_max": 0.5, "radius": 0.25}
Xu, gu, _, ou = W.du_cbfqp_batch(n_total // 2, K, seed=0)
Xq, gq, oq = W.linear_mpc_batch("Quad3D", n_total // 2, K, seed=1)
uni = sca.BatchedOptimalDecayMPCCBF(uspec, io_dtype="f32", horizon=N, extension=True)
quad = sca.BatchedOptimalDecayLinearMPCCBF({"model": "Quad3D"}, io_dtype="f32", horizon=N)
Xu[:, 3] = 0.0
ou = W.superellipsoid_obstacles(Xu[:, :2], K, seed=1000)
oq = W.superellipsoid_obstacles(Xq[:, :2], K, seed=1001)
t = lambda arr: torch.tensor(arr, dtype=torch.float32, device=dev)
tXu, tgu, tou, tXq, tgq, toq = t(Xu), t(gu), t(ou), t(Xq), t(gq), t(oq)
upu = torch.zeros((n_total // 2, 2), dtype=torch.float32, device=dev)
upq = torch.zeros((n_total // 2, 4), dtype=torch.float32, device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def run(do_u, do_q):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if do_u:
        with torch.cuda.stream(s1):
            uni.solve(tXu, upu, tgu, tou)
    if do_q:
        with torch.cuda.stream(s2):
            quad.solve(tXq, upq, tgq, toq)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


run(True, True)
print("unicycle alone %.1f ms, quad3d alone %.1f ms, together %.1f ms" % (run(True, False), run(False, True), run(True, True)))
