#!/usr/bin/env python
"""One guided-filter pass at 8 x 3840x2160 (radius 45, eps 3, flat guide), grey and colour src:
median of 7 HIP-event timings of the whole call, on one stream and with the default two halves.

    python3 tools/gf_pass_time.py [radius]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import reflectance_filtering_amd as rf
from reflectance_filtering_amd import _ffi

radius = int(sys.argv[1]) if len(sys.argv) > 1 else 45
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
n, h, w = 8, 2160, 3840
scene, grey = bench.synth_batch(torch, n, h, w, 5000, dev)
flat = bench.flat_guide(scene)
for one in (1, 0):
    for tag, src in (("grey", grey), ("colour", scene)):
        dst = torch.empty_like(src)
        ts = []
        with _ffi.debug_options(gf_one_stream=one):
            for _ in range(3):
                rf.ops.guided_filter_u8(flat, src, radius, 3.0, out=dst)
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rf.ops.guided_filter_u8(flat, src, radius, 3.0, out=dst)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
        print("%s src, %s: %.3f ms per pass (%.1f GP/s)"
              % (tag, "one stream" if one else "two halves", sorted(ts)[3],
                 n * h * w / 1e6 / sorted(ts)[3]), flush=True)
