import sys; sys.path.insert(0,'/root/repo')
import torch, bench, json
r = bench.vtol_fleet_closed_loop_leg(torch.device("cuda:0"), B=1024)
print(json.dumps({k: v for k, v in r.items() if k != "workload"}, indent=1))
