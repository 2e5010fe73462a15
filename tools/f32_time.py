#!/usr/bin/env python
"""CV_32F joint bilateral at 1080p (radius 33): whole-call time (value range, host-built table,
kernel) for 3/3, 3/1 and 1/1 joint/src channels."""
import sys; sys.path.insert(0, ".")
import torch, bench
import reflectance_filtering_amd as rf
dev = torch.device("cuda", 0)
sc, gr = bench.synth_batch(torch, 2, 1080, 1920, 5003, dev)
jf = sc.float().div_(255.0).contiguous(); sf = gr.float().div_(255.0).contiguous()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for tag, j, s in (("3ch/3ch", jf[:1], sf[:1]), ("3ch/1ch", jf[:1], sf[:1, :, :, :1].contiguous()), ("1ch/1ch", sf[:1, :, :, :1].contiguous(), sf[:1, :, :, :1].contiguous())):
    ms = timed(lambda: rf.ops.joint_bilateral_f32(j, s, -1, 20 / 255.0, 22.0))
    print(tag, "%.2f ms  %.0f MP/s" % (ms, 1080 * 1920 / 1e3 / ms))
