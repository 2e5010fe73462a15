#!/usr/bin/env python
"""Where a wave of the guided filter's stage 1 spends its cycles: s_memtime stamps around the phases
of its row loop (diagnostic build -DRF_GF_S1_STAMP of csrc/rf_gf.hip; the product library is untouched).

    make -C reflectance_filtering_amd/csrc && python tools/gf_s1_stamp.py build     # here (no GPU)
    gpurun -- python3 tools/gf_s1_stamp.py run [batch] [grey|colour] [name=value,...]   # on the GPU

`build` compiles rf_gf.hip with the stamps into reflectance_filtering_amd/librf_hip.so.s1stamp (the
other objects are the product's own).  `run` filters `batch` 4K images (one pass, one stream, stage 1
only - debug options gf_one_stream, gf_exp_skip=6 - plus the options given) and prints the cycles per
row and wave by phase.  Every stamp waits for the wave's outstanding scalar / LDS traffic
(s_memtime returns through lgkmcnt), so phases that end in LDS traffic read a little long.
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reflectance_filtering_amd", "csrc")
LIB = os.path.join(ROOT, "reflectance_filtering_amd", "librf_hip.so.s1stamp")
PHASES = ["entering row: loads + accumulate", "wave totals: readlane + LDS atomics (until they land)",
          "barrier 1", "wave bases + prefix stores", "barrier 2",
          "window means + algebra + alpha/beta store", "leaving row: loads + accumulate",
          "column sums + wave scan (DPP)"]


def build():
    obj = os.path.join(CSRC, "_gf_s1_stamp_tmp.o")
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-gpu-rdc",
             "-fno-slp-vectorize", "-DRF_GF_S1_STAMP=1", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CSRC, "rf_gf.hip"), "-o", obj])
    objs = [os.path.join(CSRC, o) for o in ["rf_api.o", "rf_jbf.o", "rf_cnn.o", "rf_colorize.o", "rf_whdr.o"]
            + ["rf_gf_fused_%d.o" % k for k in range(8)]]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj] + objs)
    os.remove(obj)
    print("built", LIB)


def run(argv):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import reflectance_filtering_amd as rf
    n = int(argv[0]) if argv else 32
    kind = argv[1] if len(argv) > 1 else "grey"
    extra = argv[2] if len(argv) > 2 else ""
    rf._ffi.LIB_PATH = LIB
    lib = rf._ffi.load_library()
    lib.rf_debug_gf_s1_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    lib.rf_debug_option(b"gf_one_stream", 1)
    lib.rf_debug_option(b"gf_exp_skip", 6)
    for item in filter(None, extra.split(",")):
        name, _, value = item.partition("=")
        lib.rf_debug_option(name.encode(), int(value or 1))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    scene, grey = bench.synth_batch(torch, n, 2160, 3840, 6234, dev)
    guide = bench.flat_guide(scene)
    src = grey if kind == "grey" else scene.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
    dst = torch.empty_like(src)
    buf = (ctypes.c_ulonglong * 16)()
    rf.ops.guided_filter_u8(guide, src, 45, 3.0, out=dst)
    torch.cuda.synchronize()
    lib.rf_debug_gf_s1_stamps(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rf.ops.guided_filter_u8(guide, src, 45, 3.0, out=dst)
    e1.record()
    torch.cuda.synchronize()
    assert lib.rf_debug_gf_s1_stamps(buf) == 0
    rows = buf[15]
    doc = {"batch": n, "kind": kind, "options": extra, "stage1_ms": e0.elapsed_time(e1), "wave_rows": rows,
           "cycles_per_row_and_wave": {PHASES[i]: round(buf[i] / max(1, rows), 1) for i in range(8)}}
    doc["cycles_per_row_and_wave"]["total"] = round(sum(buf[i] for i in range(8)) / max(1, rows), 1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])
