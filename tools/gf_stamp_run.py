#!/usr/bin/env python
"""Runs the stamped column walk (tools/gf_stamp_build.py) and prints cycles per sub-tile and phase.

    python3 tools/gf_stamp_run.py [radius] [batch] [grey|colour]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import reflectance_filtering_amd as rf

radius = int(sys.argv[1]) if len(sys.argv) > 1 else 45
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kind = sys.argv[3] if len(sys.argv) > 3 else "grey"
rf._ffi.LIB_PATH = os.path.join(ROOT, "reflectance_filtering_amd",
                                "librf_hip.so.stamp" + os.environ.get("RF_STAMP_SUFFIX", ""))
lib = rf._ffi.load_library()
lib.rf_debug_option(b"gf_one_stream", 1)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
scene, grey = bench.synth_batch(torch, n, 2160, 3840, 5000, dev)
flat = (scene // 32) * 32 + 16
src = grey if kind == "grey" else scene
dst = torch.empty_like(src)
buf = (ctypes.c_ulonglong * 16)()
rf.ops.guided_filter_u8(flat, src, radius, 3.0, out=dst)
torch.cuda.synchronize()
lib.rf_debug_gf_stamps(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
rf.ops.guided_filter_u8(flat, src, radius, 3.0, out=dst)
e1.record()
torch.cuda.synchronize()
assert lib.rf_debug_gf_stamps(buf) == 0
names = ["loop head + row table", "wait operands + stage", "stores + guide/operand fetch issue",
         "row chains", "wait for SUM hand-off", "column phase + publish", "flush"]
subs = buf[15]
doc = {"radius": radius, "batch": n, "kind": kind, "call_ms": e0.elapsed_time(e1),
       "wave_subtiles": subs,
       "cycles_per_subtile": {names[i]: buf[i] / max(1, subs) for i in range(7)}}
doc["cycles_per_subtile"]["total"] = sum(buf[i] for i in range(7)) / max(1, subs)
print(json.dumps(doc, indent=1))
