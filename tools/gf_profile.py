#!/usr/bin/env python
"""Run the guided filter a few times at 4K (for rocprofv3 --kernel-trace --stats / --pmc).

    python3 tools/gf_profile.py [batch] [h w] [grey|colour] [iterations] [wall|nowall] [one|two] [radius] [lib]
(`wall`: also write the wall time of a call to gpurun_out/gf_profile_wall_<kind>.json;
 `one`: the whole batch on one stream, so that kernel durations are those of kernels running alone;
 `lib`: another build of librf_hip.so to load instead of the in-tree one)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import reflectance_filtering_amd as rf
if len(sys.argv) > 9:
    rf._ffi.LIB_PATH = os.path.abspath(sys.argv[9])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
kind = sys.argv[4] if len(sys.argv) > 4 else "grey"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 1
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
scene, grey = bench.synth_batch(torch, n, h, w, 5000, dev)
flat = (scene // 32) * 32 + 16
src = grey if kind == "grey" else scene
dst = torch.empty_like(src)
radius = int(sys.argv[8]) if len(sys.argv) > 8 else 45
if len(sys.argv) > 7 and sys.argv[7] == "one":
    rf._ffi.load_library().rf_debug_option(b"gf_one_stream", 1)
wall = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rf.ops.guided_filter_u8(flat, src, radius, 3.0, iterations=iters, out=dst)
    e1.record()
    torch.cuda.synchronize()
    wall.append(e0.elapsed_time(e1))
# wall time of a call (the two halves of the batch overlap on two streams, so the kernels'
# durations do not add up to it); the first call also uploads nothing new after the warm-up below
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(out) and len(sys.argv) > 6 and sys.argv[6] == "wall":
    import json
    with open(os.path.join(out, "gf_profile_wall_%s.json" % kind), "w") as fh:
        json.dump({"kind": kind, "batch": n, "h": h, "w": w, "iterations": iters,
                   "wall_ms": sorted(wall)[1]}, fh)
