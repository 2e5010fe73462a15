#!/usr/bin/env python
"""Guided filter, one pass over 8 x 3840x2160 (grey CNN-style map as a 3-channel src), ms per call by
radius - across the boundaries between the forms of stage 2 (round 6: fused up to 128, float kernels
beyond):    gpurun -- python tools/gf_radius_time.py
"""
import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import reflectance_filtering_amd as rf
dev = torch.device("cuda:0")
n, h, w = 8, 2160, 3840
scene, grey = bench.synth_batch(torch, n, h, w, 77, dev)
guide = bench.flat_guide(scene)
out = torch.empty_like(grey)
res = {}
for r in (45, 64, 80, 96, 97, 110, 120, 121, 128, 200):
    for _ in range(2):
        rf.ops.guided_filter_u8(guide, grey, r, 3.0, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rf.ops.guided_filter_u8(guide, grey, r, 3.0, out=out); e1.record(); torch.cuda.synchronize()
    res[r] = round(e0.elapsed_time(e1), 3)
print(json.dumps(res))
