#!/usr/bin/env python
"""Single-image latency of the guided filter with and without the exact-row stage 2 (round 6).

    gpurun -- python tools/gf_latency_exact.py > gpurun_out/lat.json

1 / 2 / 4 images at 256x256, 1920x1080 and 3840x2160, one pass at radius 45, grey (3-channel) and
colour src: best of 10 event-timed calls, ms per call, debug option gf_exact = 0 / 1
(profiles/r06_gf_exact.md, "Single images").
"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import reflectance_filtering_amd as rf
from reflectance_filtering_amd import _ffi
dev = torch.device("cuda:0")
res = {}
for (h, w) in ((256, 256), (1080, 1920), (2160, 3840)):
    for n in (1, 2, 4):
        scene, grey = bench.synth_batch(torch, n, h, w, 77, dev)
        guide = bench.flat_guide(scene)
        for kind, src in (("grey3", grey), ("colour", scene.roll(shifts=(37, 91), dims=(1, 2)).contiguous())):
            out = torch.empty_like(src)
            for opt in (0, 1):
                with _ffi.debug_options(gf_exact=opt):
                    for _ in range(3):
                        rf.ops.guided_filter_u8(guide, src, 45, 3.0, out=out)
                    torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(10):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); rf.ops.guided_filter_u8(guide, src, 45, 3.0, out=out); e1.record()
                        torch.cuda.synchronize()
                        best = min(best, e0.elapsed_time(e1))
                res["%dx%d n=%d %s exact=%d" % (w, h, n, kind, opt)] = round(best, 4)
print(json.dumps(res, indent=1))
