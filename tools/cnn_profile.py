#!/usr/bin/env python
"""Run the CNN at IIW size (for rocprofv3 --kernel-trace --stats / --pmc): 5 warm-up launches, then
25 launches each bracketed by HIP events.  The median event time goes to
gpurun_out/cnn_profile_events.json, so that the profiler's per-kernel average and the event timing
of the SAME launches can be compared (three cold launches, as this tool used to make, ran 15 % slower
than steady state: the clock had not ramped).

    python3 tools/cnn_profile.py [batch] [launches]
"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import reflectance_filtering_amd as rf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 25
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
scene, _ = bench.synth_batch(torch, n, 333, 500, 5002, dev)
for _ in range(5):
    rf.get_reflectance_batch(scene)
torch.cuda.synchronize()
ms = []
for _ in range(launches):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rf.get_reflectance_batch(scene)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
ms.sort()
rec = {"batch": n, "launches": launches, "event_ms_median": ms[len(ms) // 2], "event_ms_min": ms[0],
       "mp_per_s_median": n * 333 * 500 / 1e3 / ms[len(ms) // 2]}
print(json.dumps(rec))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(out):
    with open(os.path.join(out, "cnn_profile_events.json"), "w") as fh:
        json.dump(rec, fh)
