#!/usr/bin/env python
"""Diagnostic: per-phase cycle stamps of the guided filter's column walk (needs a stamped build
of rf_gf.hip linked as librf_hip.so.st; not part of the product)."""
import sys, os, ctypes
sys.path.insert(0, ".")
import torch, bench
from reflectance_filtering_amd import _ffi
_ffi.LIB_PATH = os.path.join("reflectance_filtering_amd", "librf_hip.so.st")
import reflectance_filtering_amd as rf
dev = torch.device("cuda", 0)
scene, grey = bench.synth_batch(torch, 8, 2160, 3840, 5000, dev)
flat = bench.flat_guide(scene); dst = torch.empty_like(scene)
lib = _ffi.load_library()
for name, src in (("colour", scene), ("grey", grey)):
    for _ in range(2):
        rf.ops.guided_filter_u8(flat, src, 45, 3.0, out=dst)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.rf_cw_stamps(buf)
    v = list(buf)
    tot = sum(v[:5])
    print(name, "nsub", v[5], "cycles/sub-tile: staging+sync %.0f, fetch-issue %.0f, row chains %.0f, column phase %.0f, flush %.0f, total %.0f"
          % tuple([x / max(1, v[5]) for x in v[:5]] + [tot / max(1, v[5])]))
