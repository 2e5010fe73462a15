#!/usr/bin/env python
"""Time rf_cnn_reflectance_u8 of several librf_hip.so builds in one process (interleaved rounds)
and check that they return identical results.

    python tools/cnn_ab.py [--batch 256] [--rounds 7] --libs a.so,b.so
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--libs", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi, weights as W, image_utils as iu

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n, h, w = args.batch, 333, 500
    scene, _ = bench.synth_batch(torch, n, h, w, 5002, dev)
    wts = torch.from_numpy(np.ascontiguousarray(W.load_weights())).to(dev)
    lut = torch.from_numpy(iu.srgb_byte_lut()).to(dev)
    libs = [("default", _ffi.load_library())]
    for path in filter(None, args.libs.split(",")):
        lib = ctypes.CDLL(path)
        lib.rf_cnn_reflectance_u8.argtypes = _ffi.load_library().rf_cnn_reflectance_u8.argtypes
        lib.rf_cnn_reflectance_u8.restype = ctypes.c_int
        libs.append((os.path.basename(path), lib))
    stream = _ffi.current_stream_ptr(torch)
    outs, times = {}, {name: [] for name, _ in libs}
    for name, lib in libs:
        r = torch.empty((n, h, w), dtype=torch.float32, device=dev)
        r8 = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
        outs[name] = (r, r8)
    for rnd in range(args.rounds + 1):
        for name, lib in libs:
            r, r8 = outs[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.rf_cnn_reflectance_u8(scene.data_ptr(), r.data_ptr(), r8.data_ptr(), n, h, w,
                                           wts.data_ptr(), lut.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0, (name, rc)
            if rnd:
                times[name].append(e0.elapsed_time(e1))
    first = outs[libs[0][0]]
    for name, _ in libs:
        same = torch.equal(outs[name][0], first[0]) and torch.equal(outs[name][1], first[1])
        ms = sorted(times[name])[len(times[name]) // 2]
        print("%-24s median %.3f ms -> %.0f MP/s  identical_to_first=%s"
              % (name, ms, n * h * w / 1e6 / (ms * 1e-3), same))


if __name__ == "__main__":
    main()
