#!/usr/bin/env python
"""Exactness census of the guided filter's stage 2 (CPU only; the oracle is the instrument).

The box means of the float planes alpha / beta are the part of the filter whose summation ORDER is
contractual (RowSum<float,double> along x, ColumnSum<double,float> down y): that order is what
forces a sequential row walk and a sequential column walk on the GPU.  A double sum of floats that
never rounds is the exact sum, hence order-free.  This tool counts, with a TwoSum check on every
operation of the oracle's chains (oracle/rf_oracle.c rfo_box_mean_census), how often that is the
case on the inputs of BASELINE config C5 and on a natural-image guide.

    python tools/gf_exactness.py [--height 2160 --width 3840] [--out profiles/r06_gf_exactness.md]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ("rows", "rows_rounded", "row_ops", "row_ops_rounded", "cols", "cols_rounded", "col_ops",
         "col_ops_rounded", "rows_pass_order_free_test", "planes", "planes_exact", "blocks64",
         "blocks64_pass_joint_test")


def census_run(L, guide, src, radius, eps, passes):
    """passes of the oracle's guided filter with the census on; returns one dict per pass."""
    from oracle import c_oracle
    out = []
    buf = (ctypes.c_ulonglong * 16)()
    cur = src
    for _ in range(passes):
        L.rfo_census(1, None)
        cur = c_oracle.guided_filter(guide, cur, radius, eps)
        if cur.ndim == 2:
            cur = cur[:, :, None]
        L.rfo_census(0, buf)
        out.append({k: int(buf[i]) for i, k in enumerate(NAMES)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--radius", type=int, default=45)
    ap.add_argument("--eps", type=float, default=3.0)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import bench
    from oracle import c_oracle
    L = c_oracle.lib()
    L.rfo_census.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
    L.rfo_census.restype = None
    h, w = args.height, args.width
    dev = torch.device("cpu")
    scene, grey = bench.synth_batch(torch, 1, h, w, 1234 + 5000, dev)
    flat = bench.flat_guide(scene)
    scene_np = scene[0].numpy().copy()
    grey_np = grey[0].numpy().copy()
    flat_np = flat[0].numpy().copy()
    colour_src = np.roll(scene_np, (37, 91), axis=(0, 1)).copy()
    cases = (("C5: flat (Voronoi) guide, grey src", flat_np, grey_np[:, :, :1].copy()),
             ("C5 guide, colour src", flat_np, colour_src),
             ("natural-image guide (the scene), grey src", scene_np, grey_np[:, :, :1].copy()))
    lines = ["# exactness census of the guided filter's stage 2 (box means of alpha / beta)",
             "",
             "`python tools/gf_exactness.py --height %d --width %d` (CPU; oracle/rf_oracle.c "
             "`rfo_box_mean_census`: the oracle's chains with a TwoSum check on every double "
             "operation).  Radius %d, eps %g, %d passes, one %dx%d image per case, inputs of "
             "`bench.py` (`synth_batch`, `flat_guide`).  An operation is *rounded* when its double "
             "result differs from the exact sum; a row / column is *exact* when none of its "
             "operations rounded - its sums are then order-free." % (h, w, args.radius, args.eps,
                                                                    args.passes, w, h), ""]
    lines.append("| case | pass | planes | planes fully exact | rows exact | rounded row ops | "
                 "columns exact | rounded column ops | rows passing the order-free test | "
                 "64-row blocks passing the joint test (per plane) |")
    lines.append("|---|---|---|---|---|---|---|---|---|---|")
    for name, g, s in cases:
        t0 = time.time()
        res = census_run(L, g, s, args.radius, args.eps, args.passes)
        for k, c in enumerate(res):
            lines.append("| %s | %d | %d | %d | %.4f %% | %.3e of %.3e | %.4f %% | %.3e of %.3e | %.4f %% | %d of %d (%.2f %%) |" % (
                name, k + 1, c["planes"], c["planes_exact"],
                100.0 * (c["rows"] - c["rows_rounded"]) / max(1, c["rows"]),
                c["row_ops_rounded"], c["row_ops"],
                100.0 * (c["cols"] - c["cols_rounded"]) / max(1, c["cols"]),
                c["col_ops_rounded"], c["col_ops"],
                100.0 * c["rows_pass_order_free_test"] / max(1, c["rows"]),
                c["blocks64_pass_joint_test"], c["blocks64"],
                100.0 * c["blocks64_pass_joint_test"] / max(1, c["blocks64"])))
        sys.stderr.write("%s: %.1f s\n" % (name, time.time() - t0))
    txt = "\n".join(lines) + "\n"
    sys.stdout.write(txt)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(txt)


if __name__ == "__main__":
    main()
