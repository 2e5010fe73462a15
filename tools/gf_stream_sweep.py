#!/usr/bin/env python
"""Guided filter (radius 45, 3 passes, 4K): one stream against two streams over the batch size,
grey and colour src - the data behind the library's choice of when to fork its side stream.

    python3 tools/gf_stream_sweep.py [--batches 2,4,8,12,16,24,36] [--rounds 3]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="2,4,8,12,16,24,36")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    h, w = 2160, 3840
    nmax = max(int(b) for b in args.batches.split(","))
    scene, grey = bench.synth_batch(torch, nmax, h, w, 5000, dev)
    flat = (scene // 32) * 32 + 16
    ws = rf.ops.gf_workspace(nmax, h, w, 3, 45, dev, torch)
    out = {}
    for n in [int(b) for b in args.batches.split(",")]:
        for tag, src in (("grey", grey), ("colour", scene)):
            dst = torch.empty_like(src[:n])
            t = {"two_streams": [], "one_stream": []}
            for rnd in range(args.rounds + 1):
                for mode in ("two_streams", "one_stream"):
                    with _ffi.debug_options(gf_one_stream=int(mode == "one_stream"), gf_force_two_streams=int(mode == "two_streams")):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        rf.ops.guided_filter_u8(flat[:n], src[:n], 45, 3.0, iterations=args.iters,
                                                out=dst, workspace=ws)
                        e1.record()
                        torch.cuda.synchronize()
                        if rnd:
                            t[mode].append(e0.elapsed_time(e1))
            rec = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
            rec["two_over_one"] = rec["two_streams"] / rec["one_stream"]
            rec["best_mp_per_s"] = n * h * w / 1e6 / (min(rec["two_streams"], rec["one_stream"]) * 1e-3)
            out["%s_n%d" % (tag, n)] = rec
            print(tag, n, json.dumps(rec), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
