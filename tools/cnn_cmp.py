#!/usr/bin/env python
"""CNN: the register-resident kernel (default) against the LDS-column kernel of round 1 in one
process: identical values, timing at 256 IIW-size images."""
import sys
sys.path.insert(0, ".")
import torch
import bench
import reflectance_filtering_amd as rf

dev = torch.device("cuda", 0)
scene, _ = bench.synth_batch(torch, 256, 333, 500, 5002, dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


r_new, r8_new = rf.get_reflectance_batch(scene)
ms_new = timed(lambda: rf.get_reflectance_batch(scene))
with rf._ffi.debug_options(cnn_lds_columns=1):
    r_old, r8_old = rf.get_reflectance_batch(scene)
    ms_old = timed(lambda: rf.get_reflectance_batch(scene))
print("register kernel %.3f ms (%.0f MP/s), LDS-column kernel %.3f ms; equal r: %s, equal r8: %s"
      % (ms_new, 256 * 333 * 500 / 1e3 / ms_new, ms_old, torch.equal(r_new, r_old),
         torch.equal(r8_new, r8_old)))
