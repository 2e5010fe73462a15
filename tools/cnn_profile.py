#!/usr/bin/env python
"""Run the CNN a few times at IIW size (for rocprofv3 --kernel-trace --stats / --pmc)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import reflectance_filtering_amd as rf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
scene, _ = bench.synth_batch(torch, n, 333, 500, 5002, dev)
for _ in range(3):
    rf.get_reflectance_batch(scene)
torch.cuda.synchronize()
