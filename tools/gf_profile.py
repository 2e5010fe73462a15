#!/usr/bin/env python
"""Run the guided filter a few times at 4K (for rocprofv3 --kernel-trace --stats / --pmc).

    python3 tools/gf_profile.py [batch] [h w] [grey|colour] [iterations]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import reflectance_filtering_amd as rf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
kind = sys.argv[4] if len(sys.argv) > 4 else "grey"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 1
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
scene, grey = bench.synth_batch(torch, n, h, w, 5000, dev)
flat = (scene // 32) * 32 + 16
src = grey if kind == "grey" else scene
dst = torch.empty_like(src)
for _ in range(3):
    rf.ops.guided_filter_u8(flat, src, 45, 3.0, iterations=iters, out=dst)
torch.cuda.synchronize()
