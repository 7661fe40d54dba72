import sys, json
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for f in (bench.vtol_ms_closed_loop_leg, bench.vtol_fleet_closed_loop_leg):
    r = f(dev)
    print(json.dumps({k: v for k, v in r.items() if not isinstance(v, (dict, list))})[:900])
