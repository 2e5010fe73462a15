#!/usr/bin/env python
"""Guided filter stage 1: run time of a whole pass against the rows per segment (debug option
gf_seg_rows), 4K, grey and colour src, two batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import reflectance_filtering_amd as rf
from reflectance_filtering_amd import _ffi
dev = torch.device("cuda", 0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for n in (8, 13):
    scene, grey = bench.synth_batch(torch, n, 2160, 3840, 5000, dev)
    flat = bench.flat_guide(scene); dst = torch.empty_like(scene)
    for tag, src in (("grey", grey), ("colour", scene)):
        row = []
        for sr in (0, 23, 34, 45, 54, 68, 90, 108, 135, 180, 270, 540):
            with _ffi.debug_options(gf_seg_rows=sr):
                row.append("%d:%.3f" % (sr, timed(lambda: rf.ops.guided_filter_u8(flat, src, 45, 3.0, out=dst))))
        print("batch %d %s  " % (n, tag) + "  ".join(row), flush=True)
    del scene, grey, flat, dst
    torch.cuda.empty_cache()
