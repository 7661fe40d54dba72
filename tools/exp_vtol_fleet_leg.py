import sys; sys.path.insert(0,'/root/repo')
import torch, bench, json
print(json.dumps(bench.vtol_fleet_closed_loop_leg(torch.device("cuda:0")), indent=1))
