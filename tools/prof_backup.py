#!/usr/bin/env python3
"""Run the Backup-CBF QP kernel a few times on the bench batch (driver for rocprofv3):  prof_backup.py B n"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print(bench.backup_cbf_leg(torch.device("cuda:0"), B=B, steps=n, split=False))
