#!/usr/bin/env python
"""CV_32F joint bilateral at 1080p (radius 33): whole-call time (value range, host-built table,
kernel) for 3/3, 3/1 and 1/1 joint/src channels.  `python3 tools/f32_time.py [batch]` (default 1:
the latency of one image, host round trip included; a batch amortises it)."""
import sys; sys.path.insert(0, ".")
import torch, bench
import reflectance_filtering_amd as rf
dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc, gr = bench.synth_batch(torch, max(2, nb), 1080, 1920, 5003, dev)
jf = sc.float().div_(255.0).contiguous(); sf = gr.float().div_(255.0).contiguous()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for tag, j, s in (("3ch/3ch", jf[:nb], sf[:nb]), ("3ch/1ch", jf[:nb], sf[:nb, :, :, :1].contiguous()), ("1ch/1ch", sf[:nb, :, :, :1].contiguous(), sf[:nb, :, :, :1].contiguous())):
    ms = timed(lambda: rf.ops.joint_bilateral_f32(j, s, -1, 20 / 255.0, 22.0))
    print(tag, "batch %d: %.2f ms  %.0f MP/s" % (nb, ms, nb * 1080 * 1920 / 1e3 / ms))
