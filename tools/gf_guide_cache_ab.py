#!/usr/bin/env python
"""Guided filter, experiment switch gf_guide_cache (guide statistics kept across the passes of an
iterated call): identical bytes over shapes x radii x src kinds, then 3 passes at 8 x 4K with and
without it.   python3 tools/gf_guide_cache_ab.py"""
import sys, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bench
import reflectance_filtering_amd as rf
from reflectance_filtering_amd import _ffi
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
bad = []
for (n, h, w) in [(1, 7, 5), (2, 91, 300), (3, 200, 517), (2, 333, 500), (1, 1080, 1920)]:
    scene, grey = bench.synth_batch(torch, n, h, w, 100 + h + w, dev)
    flat = (scene // 32) * 32 + 16
    mixed = scene.clone()
    if n > 1: mixed[0] = grey[0]
    for radius in (3, 9, 45, 52, 97):
        for tag, src in (("grey", grey), ("colour", scene), ("mixed", mixed), ("1ch", grey[..., :1].contiguous())):
            for iters in (2, 3):
                a = rf.ops.guided_filter_u8(flat, src, radius, 3.0, iterations=iters)
                with _ffi.debug_options(gf_guide_cache=1):
                    b = rf.ops.guided_filter_u8(flat, src, radius, 3.0, iterations=iters)
                if not torch.equal(a, b):
                    bad.append((n, h, w, radius, tag, iters, int((a != b).sum())))
print("mismatches", bad)
n, h, w = 8, 2160, 3840
scene, grey = bench.synth_batch(torch, n, h, w, 5000, dev)
flat = (scene // 32) * 32 + 16
for tag, src in (("grey", grey), ("colour", scene)):
    dst = torch.empty_like(src)
    for opt in ({}, {"gf_guide_cache": 1}):
        ts = []
        with _ffi.debug_options(**opt):
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rf.ops.guided_filter_u8(flat, src, 45, 3.0, iterations=3, out=dst); e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(tag, opt, sorted(ts)[1])
