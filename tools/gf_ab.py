#!/usr/bin/env python
"""Guided filter: fused stage 2 (row states + column walk) against the row-sum / column-sum
kernel pair, in one process: identical bytes on a sweep of shapes, then interleaved timing.

    python tools/gf_ab.py [--batch 8] [--rounds 5] [--skip-check]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--skip-check", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {"mismatches": [], "cases": 0}

    def both(guide, src, radius, eps, iters):
        a = rf.ops.guided_filter_u8(guide, src, radius, eps, iterations=iters)
        with _ffi.debug_options(gf_two_kernel=1):
            b = rf.ops.guided_filter_u8(guide, src, radius, eps, iterations=iters)
        return a, b

    if not args.skip_check:
        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        shapes = [(1, 7, 5), (2, 44, 17), (3, 45, 16), (2, 46, 15), (1, 90, 33), (2, 91, 300),
                  (1, 135, 130), (3, 200, 517), (1, 1, 64), (1, 64, 1), (2, 333, 500),
                  (1, 1080, 1920), (2, 2160, 3840)]
        for (n, h, w) in shapes:
            scene, grey = bench.synth_batch(torch, n, h, w, 100 + h + w, dev)
            flat = (scene // 32) * 32 + 16
            mixed = scene.clone()
            if n > 1:
                mixed[0] = grey[0]       # a grey image among colour ones (run-time flag per image)
            for radius in (45, 52):
                for src, iters in ((grey, 1), (scene, 1), (mixed, 3), (grey[..., :1].contiguous(), 2)):
                    if h * w > 4e6 and iters > 1 and src is not mixed:
                        continue
                    a, b = both(flat, src, radius, 3.0 if radius == 45 else 7.0, iters)
                    out["cases"] += 1
                    if not torch.equal(a, b):
                        bad = int((a != b).sum())
                        out["mismatches"].append({"n": n, "h": h, "w": w, "radius": radius,
                                                  "scn": src.shape[3], "iters": iters,
                                                  "bad_bytes": bad})
            del scene, grey, flat, mixed
        torch.cuda.empty_cache()
        print(json.dumps({k: out[k] for k in ("cases", "mismatches")}), flush=True)

    # timing at 4K
    n, h, w = args.batch, 2160, 3840
    scene, grey = bench.synth_batch(torch, n, h, w, 5000, dev)
    flat = (scene // 32) * 32 + 16
    dst = torch.empty_like(grey)

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    res = {}
    for tag, src in (("grey", grey), ("colour", scene)):
        for iters in (1, 3):
            t = {"fused": [], "fused_one_stream": [], "two_kernel": []}
            for _ in range(args.rounds + 1):
                t["fused"].append(timed(lambda: rf.ops.guided_filter_u8(flat, src, 45, 3.0,
                                                                        iterations=iters, out=dst)))
                with _ffi.debug_options(gf_one_stream=1):
                    t["fused_one_stream"].append(timed(lambda: rf.ops.guided_filter_u8(
                        flat, src, 45, 3.0, iterations=iters, out=dst)))
                with _ffi.debug_options(gf_two_kernel=1):
                    t["two_kernel"].append(timed(lambda: rf.ops.guided_filter_u8(
                        flat, src, 45, 3.0, iterations=iters, out=dst)))
            for k, v in t.items():
                v = sorted(v[1:])
                ms = v[len(v) // 2]
                res["%s_x%d_%s" % (tag, iters, k)] = {"ms": ms, "mp_per_s": n * h * w / 1e6 / (ms * 1e-3)}
    out["timing_4k_batch%d" % n] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
