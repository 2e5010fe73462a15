#!/usr/bin/env python
"""Does a light reader of alpha/beta run BESIDE the guided filter's stage 1?  (round 6)

Stage 1 is bound by VALU issue and fills every CU (4 workgroups x 40 KB LDS, 4 x 112 VGPRs per
SIMD); the walk kernels (184 / 216 VGPRs) are not admitted beside it (profiles/r05_c5_overlap.md).
A kernel of <= 64 VGPRs and no LDS would be.  This tool times, with events, on one MI355X:
  a) stage 1 alone       rf_gf_u8 with gf_exp_skip = 6 (no row walk, no column walk), one stream
  b) a reader alone      tools/microbench/corun_read.so, three access patterns, over 16 B per pixel
  c) both, started together on two streams
and prints one JSON line.  Perfect co-running: c = max(a, b); none: c = a + b.

    python tools/gf_corun.py [--images 64] [--out profiles/r06_corun.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi
    so = os.path.join(ROOT, "tools", "microbench", "corun_read.so")
    if not os.path.exists(so):          # (built artefacts are not in the history)
        import subprocess
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                               os.path.join(ROOT, "tools", "microbench", "corun_read.hip")])
    rd = ctypes.CDLL(so)
    rd.corun_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_void_p]
    rd.corun_read.restype = ctypes.c_int
    dev = torch.device("cuda:0")
    n, h, w = args.images, args.height, args.width
    scene, grey = bench.synth_batch(torch, n, h, w, 4321, dev)
    guide = bench.flat_guide(scene)
    src = grey[..., :1].contiguous()
    out = torch.empty_like(src)
    planes = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev).normal_()
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def stage1():
        rf.ops.guided_filter_u8(guide, src, 45, 3.0, iterations=1, out=out)

    def reader(v):
        rc = rd.corun_read(v, planes.data_ptr(), n, h, w, sink.data_ptr(),
                           ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError("corun_read: hip error %d" % rc)

    def timed(fn_a, fn_b):
        """ms from a common start to the end of both (each on its own stream)"""
        best = None
        for _ in range(args.reps):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            ea = torch.cuda.Event(enable_timing=True)
            eb = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
            s_a.wait_event(e0)
            s_b.wait_event(e0)
            with torch.cuda.stream(s_a):
                if fn_a:
                    fn_a()
                ea.record(s_a)
            with torch.cuda.stream(s_b):
                if fn_b:
                    fn_b()
                eb.record(s_b)
            torch.cuda.synchronize()
            t = (e0.elapsed_time(ea), e0.elapsed_time(eb))
            if best is None or max(t) < max(best):
                best = t
        return best

    res = {"images": n, "h": h, "w": w, "radius": 45, "bytes_read": n * h * w * 16}
    with _ffi.debug_options(gf_exp_skip=6, gf_one_stream=1):
        with torch.cuda.stream(s_a):
            stage1()        # workspace of stream s_a, first-use work
        torch.cuda.synchronize()
        a = timed(stage1, None)
        res["stage1_alone_ms"] = a[0]
        for v, name in ((0, "coalesced"), (1, "lane_row"), (2, "lane_row_plane")):
            reader(v)
            torch.cuda.synchronize()
            b = timed(None, lambda: reader(v))
            c = timed(stage1, lambda: reader(v))
            res[name] = {"reader_alone_ms": b[1], "reader_alone_gbs": res["bytes_read"] / b[1] / 1e6,
                         "both_stage1_ms": c[0], "both_reader_ms": c[1], "both_ms": max(c),
                         "sum_alone_ms": a[0] + b[1]}
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(line + "\n")


if __name__ == "__main__":
    main()
