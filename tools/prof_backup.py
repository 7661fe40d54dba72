#!/usr/bin/env python3
"""Run the Backup-CBF QP kernel a few times on the bench fleet (driver for rocprofv3):  prof_backup.py B n [--prepare]
--prepare (run WITHOUT the profiler first): builds the fleet with the closed-loop rollouts of bench.backup_cbf_leg and stores it under
gpurun_out/; the profiled run then loads it, so the only launches of backupcbf_kernel it sees are the n + 1 full-batch solves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
path = os.path.join(ROOT, "gpurun_out", f"backup_fleet_{B}.pt")
if "--prepare" in sys.argv:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tX, bx = bench.backup_cbf_leg(dev, B=B, steps=0)
    torch.save({"X": tX.cpu(), "bx": bx.cpu()}, path)
    print("fleet of", tX.shape[0], "agents ->", path)
else:
    f = torch.load(path)
    print(bench.backup_cbf_leg(dev, B=B, steps=n, split=False, fleet=(f["X"].to(dev), f["bx"].to(dev))))
