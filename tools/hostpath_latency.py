#!/usr/bin/env python
"""Latency of the numpy-in / numpy-out drop-in calls (upload + kernel + download) and of the PNG codec,
next to the device-resident operators: what a user of the single-image tools sees per call."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import reflectance_filtering_amd as rf
from tests import synth
h, w = 1080, 1920
joint = synth.scene_u8(h, w, seed=1); src = synth.reflectance_like_u8(h, w, seed=2); csrc = synth.scene_u8(h, w, seed=3)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts)//2] * 1e3
print("ximgproc.jointBilateralFilter grey src  : %.2f ms" % t(lambda: rf.ximgproc.jointBilateralFilter(joint, src, -1, 20, 22)))
print("ximgproc.jointBilateralFilter colour src: %.2f ms" % t(lambda: rf.ximgproc.jointBilateralFilter(joint, csrc, -1, 20, 22)))
print("ximgproc.guidedFilter r45 grey src      : %.2f ms" % t(lambda: rf.ximgproc.guidedFilter(joint, src, 45, 3.0)))
print("ximgproc.guidedFilter r45 colour src    : %.2f ms" % t(lambda: rf.ximgproc.guidedFilter(joint, csrc, 45, 3.0)))
jd = torch.from_numpy(joint[None]).cuda(); sd = torch.from_numpy(src[None]).cuda(); cd = torch.from_numpy(csrc[None]).cuda()
print("device-resident jbf grey                : %.2f ms" % t(lambda: rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0)))
print("device-resident gf grey / colour        : %.2f / %.2f ms" % (t(lambda: rf.ops.guided_filter_u8(jd, sd, 45, 3.0)), t(lambda: rf.ops.guided_filter_u8(jd, cd, 45, 3.0))))
print("H2D 6.2MB pageable: %.2f ms ; D2H: %.2f ms" % (t(lambda: torch.from_numpy(joint).cuda()), t(lambda: jd.cpu())))
pin = torch.from_numpy(joint).pin_memory()
print("H2D pinned: %.2f ms" % t(lambda: pin.cuda(non_blocking=True)))
import tempfile, os
d = tempfile.mkdtemp(); f = os.path.join(d, "a.png")
print("imwrite 1080p: %.1f ms; imread: %.1f ms" % (t(lambda: rf.image_utils.imwrite(f, joint), 3), t(lambda: rf.image_utils.imread(f), 3)))
