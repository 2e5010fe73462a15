#!/usr/bin/env python
"""Time the joint-bilateral kernel variants (rf_jbf_u8 `tune` override) against each other in
one process, interleaved rounds, and check that every variant returns identical bytes.

    python tools/jbf_tune.py [--batch 8] [--rounds 5] [--variants 1,2,3,4,5,6] [--lib path.so]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="1,2,3,4,5,6")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--sigma-color", type=float, default=20.0)
    ap.add_argument("--sigma-spatial", type=float, default=22.0)
    ap.add_argument("--grey", action="store_true", help="src = joint = grey map (BF(CNN,CNN))")
    ap.add_argument("--rgb-src", action="store_true", help="src = a second RGB scene")
    ap.add_argument("--stage-only", action="store_true", help="time tile staging alone")
    ap.add_argument("--extra-flags", type=lambda v: int(v, 0), default=0,
                    help="extra rf_jbf_u8 flag bits for the second pass over the variants")
    ap.add_argument("--libs", default="", help="comma-separated extra librf_hip builds to compare")
    args = ap.parse_args()
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    joint, src = bench.synth_batch(torch, args.batch, args.height, args.width, 4321, dev)
    if args.grey:
        joint = src.clone()
    if args.rgb_src:
        src = bench.synth_batch(torch, args.batch, args.height, args.width, 99, dev)[0]
    libs = [("default", _ffi.load_library())]
    for path in filter(None, args.libs.split(",")):
        lib = ctypes.CDLL(path)
        lib.rf_jbf_u8.argtypes = _ffi.load_library().rf_jbf_u8.argtypes
        lib.rf_jbf_u8.restype = ctypes.c_int
        lib.rf_last_error.restype = ctypes.c_char_p
        lib.rf_debug_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        libs.append((os.path.basename(path), lib))
    variants = [int(v) for v in args.variants.split(",")]
    stream = _ffi.current_stream_ptr(torch)
    n, h, w, _ = src.shape

    def run(lib, tune, out):
        extra = 0
        if tune >= 100:  # variant + 100 = the same variant with --extra-flags set
            tune -= 100
            extra = args.extra_flags
        # kernel variant and stage-only are debug options (include/reflectance_filtering_debug.h);
        # --extra-flags: bit 0x2000 / 0x4000 of older builds = jbf_compiler_loop / jbf_tile64_only
        lib.rf_debug_option(b"jbf_tune", tune)
        lib.rf_debug_option(b"jbf_stage_only", 1 if args.stage_only else 0)
        lib.rf_debug_option(b"jbf_compiler_loop", 1 if extra & 0x2000 else 0)
        lib.rf_debug_option(b"jbf_tile64_only", 1 if extra & 0x4000 else 0)
        rc = lib.rf_jbf_u8(joint.data_ptr(), src.data_ptr(), out.data_ptr(), n, h, w, 3, 3, -1,
                           args.sigma_color, args.sigma_spatial, 4, extra & 7, stream)
        lib.rf_debug_option(b"jbf_tune", 0)
        lib.rf_debug_option(b"jbf_stage_only", 0)
        if rc != 0:
            raise RuntimeError("variant %d: %s" % (tune, lib.rf_last_error()))

    ref = None
    outs = {}
    times = {}
    for name, lib in libs:
        for v in variants:
            out = torch.empty_like(src)
            try:
                run(lib, v, out)
            except RuntimeError as exc:
                print("skip", name, v, exc)
                continue
            torch.cuda.synchronize()
            outs[(name, v)] = out
            if ref is None:
                ref = out
            same = bool(torch.equal(ref, out))
            print("%-24s tune=%d identical_to_first=%s" % (name, v, same), flush=True)
            times[(name, v)] = []
    for _ in range(args.rounds):
        for (name, v), out in outs.items():
            lib = dict(libs)[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(lib, v, out)
            e1.record()
            torch.cuda.synchronize()
            times[(name, v)].append(e0.elapsed_time(e1))
    mp = n * h * w / 1e6
    for (name, v), ts in times.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        print("%-24s tune=%d  median %.3f ms  min %.3f ms  -> %.0f MP/s (median)"
              % (name, v, med, ts[0], mp / (med * 1e-3)))


if __name__ == "__main__":
    main()
