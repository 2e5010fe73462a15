#!/usr/bin/env python
"""Joint bilateral, colour src, 8 x 1080p: MP/s and G taps/s by sigma_spatial around the radii where
the one-pass colour tile stops fitting (radius 43..52) - gpurun -- python tools/jbf_colour_radius_time.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import reflectance_filtering_amd as rf

dev = torch.device("cuda:0")
n, h, w = 8, 1080, 1920
scene, grey = bench.synth_batch(torch, n, h, w, 5005, dev)
joint = bench.flat_guide(scene)
colour = scene.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
out = torch.empty_like(colour)
res = {}
for ss in (22.0, 28.0, 29.0, 30.0, 32.0, 34.5, 36.0):
    radius = int(round(1.5 * ss))
    taps = sum(1 for i in range(-radius, radius + 1) for j in range(-radius, radius + 1)
               if (i * i + j * j) ** 0.5 <= radius)
    for name, src in (("colour", colour), ("grey", grey)):
        for _ in range(2):
            rf.ops.joint_bilateral_u8(joint, src, -1, 15.0, ss, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rf.ops.joint_bilateral_u8(joint, src, -1, 15.0, ss, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        res["s%g r%d %s" % (ss, radius, name)] = {"ms": round(ms, 2), "mp_per_s": round(n * h * w / 1e3 / ms, 1),
                                                   "gtaps_per_s": round(n * h * w * taps / 1e6 / ms, 1)}
print(json.dumps(res, indent=1))
