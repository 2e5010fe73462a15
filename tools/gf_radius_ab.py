#!/usr/bin/env python
"""Guided filter, fused stage 2 (row states + column walk) for any radius:

  1. identical bytes to the row-sum / column-sum kernel pair (debug option gf_two_kernel) over a
     sweep of radii x shapes x src kinds x chained passes (the oracle comparison of every radius
     is tests/test_gpu_parity.py::test_gf_fused_stage2_any_radius and tests/test_gpu_fuzz.py);
  2. time per pass at 4K for a list of radii (fused and two-kernel), grey and colour src;
  3. optionally the same call through other builds of librf_hip.so (--libs a.so,b.so),
     interleaved, identical-bytes check against the default build.

    python tools/gf_radius_ab.py [--batch 8] [--rounds 5] [--radii 8,20,30,45,52,60]
                                 [--sweep-radii 1,2,...] [--skip-check] [--libs path,...] [--out f.json]
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--radii", default="8,20,30,45,52,60")
    ap.add_argument("--sweep-radii", default="1,2,3,5,8,13,16,17,20,30,31,32,33,45,47,48,52,60,64,65,77,96,97")
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--libs", default="")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {"mismatches": [], "cases": 0}

    if not args.skip_check:
        shapes = [(1, 7, 5), (2, 44, 17), (3, 45, 16), (1, 90, 33), (2, 91, 300), (1, 135, 130),
                  (1, 1, 64), (1, 64, 1), (2, 333, 500), (1, 1080, 1920)]
        for (n, h, w) in shapes:
            scene, grey = bench.synth_batch(torch, n, h, w, 100 + h + w, dev)
            flat = (scene // 32) * 32 + 16
            mixed = scene.clone()
            if n > 1:
                mixed[0] = grey[0]       # a grey image among colour ones (run-time flag per image)
            for radius in [int(r) for r in args.sweep_radii.split(",")]:
                if h * w > 1e6 and radius not in (8, 20, 30, 45, 60):
                    continue
                eps = 3.0 if radius % 2 else 7.0
                for tag, src, iters in (("grey", grey, 1), ("colour", scene, 1), ("mixed", mixed, 3),
                                        ("1ch", grey[..., :1].contiguous(), 2)):
                    a = rf.ops.guided_filter_u8(flat, src, radius, eps, iterations=iters)
                    with _ffi.debug_options(gf_two_kernel=1):
                        b = rf.ops.guided_filter_u8(flat, src, radius, eps, iterations=iters)
                    out["cases"] += 1
                    bad = None
                    if not torch.equal(a, b):
                        bad = {"vs": "two_kernel", "bad_bytes": int((a != b).sum())}
                    if bad:
                        bad.update({"n": n, "h": h, "w": w, "radius": radius, "src": tag, "iters": iters})
                        out["mismatches"].append(bad)
            del scene, grey, flat, mixed
        torch.cuda.empty_cache()
        print(json.dumps({k: out[k] for k in ("cases", "mismatches")}), flush=True)

    # timing at 4K
    n, h, w = args.batch, 2160, 3840
    scene, grey = bench.synth_batch(torch, n, h, w, 5000, dev)
    flat = (scene // 32) * 32 + 16
    dst = torch.empty_like(grey)
    lib0 = _ffi.load_library()
    libs = [("default", lib0)]
    for path in filter(None, args.libs.split(",")):
        lib = ctypes.CDLL(path)
        lib.rf_gf_u8.argtypes = lib0.rf_gf_u8.argtypes
        lib.rf_gf_u8.restype = ctypes.c_int
        libs.append((os.path.basename(path), lib))
    ws = rf.ops.gf_workspace(n, h, w, 3, 45, dev, torch)
    stream = _ffi.current_stream_ptr(torch)

    def call(lib, src, radius, iters, d):
        rc = lib.rf_gf_u8(flat.data_ptr(), src.data_ptr(), d.data_ptr(), n, h, w, 3, 3, radius,
                          3.0, iters, ws.data_ptr(), ws.numel(), stream)
        assert rc == 0, rc

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    res = {}
    for radius in [int(r) for r in args.radii.split(",")]:
        for tag, src in (("grey", grey), ("colour", scene)):
            t = {}
            ref = None
            for rnd in range(args.rounds + 1):
                for name, lib in libs:
                    t.setdefault(name, []).append(timed(lambda: call(lib, src, radius, 1, dst)))
                    if rnd == 0:
                        if ref is None:
                            ref = dst.clone()
                        elif not torch.equal(ref, dst):
                            out["mismatches"].append({"vs": name, "radius": radius, "src": tag})
                with _ffi.debug_options(gf_two_kernel=1):
                    t.setdefault("two_kernel", []).append(timed(lambda: call(lib0, src, radius, 1, dst)))
                with _ffi.debug_options(gf_one_stream=1):
                    t.setdefault("one_stream", []).append(timed(lambda: call(lib0, src, radius, 1, dst)))
            for k, v in t.items():
                v = sorted(v[1:])
                ms = v[len(v) // 2]
                res["r%d_%s_%s" % (radius, tag, k)] = {"ms": round(ms, 4),
                                                       "mp_per_s": round(n * h * w / 1e6 / (ms * 1e-3))}
    out["timing_4k_batch%d" % n] = res
    print(json.dumps(out, indent=1))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
