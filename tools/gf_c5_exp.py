#!/usr/bin/env python
"""Guided filter at the C5 shard (128 x 3840x2160, 3 passes, flat guide, grey map): the step time
under a list of debug-option settings, interleaved in one process (same buffers, same box).

    python tools/gf_c5_exp.py [--batch 128] [--rounds 3] [--src grey|colour] \
        "name=value,name=value" "..." ...

Each positional argument is one setting ("" or "base" = the defaults).  Prints one JSON object:
per setting the per-round times (ms per step of 3 passes) and their median, plus MP/s.  Settings
that change results ("gf_exp_skip") are timing experiments; `--check` compares every setting's
output with the first one's and reports the number of differing bytes.
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(setting):
    opts = {}
    if setting in ("", "base"):
        return opts
    for item in setting.split(","):
        name, _, value = item.partition("=")
        opts[name.strip()] = int(value or 1)
    return opts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--radius", type=int, default=45)
    ap.add_argument("--src", choices=("grey", "colour"), default="grey")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--lib", default=None,
                    help="A/B timing: load THIS build of librf_hip.so instead of the package's own")
    ap.add_argument("settings", nargs="*", default=["base"])
    args = ap.parse_args()
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi
    if args.lib:
        _ffi.LIB_PATH = os.path.abspath(args.lib)
        sys.stderr.write("gf_c5_exp: loading %s instead of the package's library\n" % _ffi.LIB_PATH)
        if _ffi.load_library().rf_version() & 0x3fffffff < 100:
            raise SystemExit("gf_c5_exp: %s is not a librf_hip.so" % _ffi.LIB_PATH)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n, h, w = args.batch, 2160, 3840
    scene, grey = bench.synth_batch(torch, n, h, w, 1234 + 5000, dev)
    guide = bench.flat_guide(scene)
    src = grey if args.src == "grey" else scene.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
    del scene
    dst = torch.empty_like(src)
    ws = rf.ops.gf_workspace(n, h, w, 3, args.radius, dev, torch)

    def step():
        rf.ops.guided_filter_u8(guide, src, args.radius, 3.0, iterations=args.passes, out=dst,
                                workspace=ws)

    def timed():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    res = {"batch": n, "passes": args.passes, "radius": args.radius, "src": args.src, "settings": {}}
    times = {s: [] for s in args.settings}
    ref = None
    for s in args.settings:             # warm every variant once (and compare if asked)
        with _ffi.debug_options(**parse(s)):
            step()
            torch.cuda.synchronize()
            if args.check:
                if ref is None:
                    ref = dst.clone()
                else:
                    res["settings"].setdefault(s, {})["bytes_differing_from_first"] = int(
                        (dst != ref).sum())
    for _ in range(args.rounds):
        for s in args.settings:
            with _ffi.debug_options(**parse(s)):
                times[s].append(timed())
    for s in args.settings:
        med = statistics.median(times[s])
        res["settings"].setdefault(s, {}).update(
            {"ms": [round(t, 3) for t in times[s]], "median_ms": round(med, 3),
             "mp_per_s": round(n * h * w * args.passes / 1e6 / (med * 1e-3), 1)})
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
