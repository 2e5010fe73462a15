#!/usr/bin/env python
"""Randomised cross-checks on the GPU at sizes the CPU oracle cannot cover in test time.

JBF: the LDS-tiled kernels (all tile shapes, strips, grey / colour tiles) against the untiled
one-thread-per-pixel kernel (RF_JBF_FORCE_GENERIC), an independent implementation of the same
arithmetic.  GF: a grey 3-channel src (one-channel fast path) against the same plane passed as a
1-channel src, and a colour src against its three planes filtered separately.

    python tools/fuzz_parity.py [--seconds 120] [--seed 1]
Prints one summary line (JSON).  Exit code 1 on any mismatch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    stats = {"jbf_cases": 0, "jbf_pixels": 0, "gf_cases": 0, "gf_pixels": 0, "mismatches": []}
    while time.time() < t_end:
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 1300)), int(rng.integers(1, 2000))
        scene, grey3 = bench.synth_batch(torch, n, h, w, int(rng.integers(1 << 30)), dev)
        if rng.random() < 0.3:   # hard edges: zero colour weights, reflected borders stand out
            scene = (scene // 64) * 64
        # ---------------- JBF
        jcn, scn = int(rng.choice([1, 3])), int(rng.choice([1, 3]))
        kind = rng.random()
        src = grey3 if kind < 0.5 else bench.synth_batch(torch, n, h, w, 7, dev)[0]
        joint = scene if jcn == 3 else scene[..., 1:2].contiguous()
        src = src if scn == 3 else src[..., :1].contiguous()
        ss = float(rng.choice([22.0, 28.0, 5.0, 12.3, 34.0]))
        sc = float(rng.choice([20.0, 15.0, 4.0, 60.0]))
        border = int(rng.choice([0, 1, 2, 3, 4]))
        as_bgr = bool(jcn == 1 and rng.random() < 0.5)
        a = rf.ops.joint_bilateral_u8(joint, src, -1, sc, ss, border=border, grey_as_bgr=as_bgr)
        b = rf.ops.joint_bilateral_u8(joint, src, -1, sc, ss, border=border, grey_as_bgr=as_bgr,
                                      flags=_ffi.JBF_FORCE_GENERIC)
        stats["jbf_cases"] += 1
        stats["jbf_pixels"] += n * h * w
        if not torch.equal(a, b):
            stats["mismatches"].append(["jbf", n, h, w, jcn, scn, sc, ss, border, as_bgr])
        # ---------------- GF
        r = int(rng.choice([1, 9, 45, 52, 100]))
        eps = float(rng.choice([3.0, 7.0, 0.5, 1e-3]))
        iters = int(rng.choice([1, 3]))
        g3 = rf.ops.guided_filter_u8(scene, grey3, r, eps, iterations=iters)
        g1 = rf.ops.guided_filter_u8(scene, grey3[..., :1].contiguous(), r, eps, iterations=iters)
        ok = torch.equal(g3, g1.expand(-1, -1, -1, 3))
        col = bench.synth_batch(torch, n, h, w, 11, dev)[0]
        c3 = rf.ops.guided_filter_u8(scene, col, r, eps, iterations=iters)
        for ch in range(3):
            one = rf.ops.guided_filter_u8(scene, col[..., ch:ch + 1].contiguous(), r, eps,
                                          iterations=iters)
            ok = ok and torch.equal(c3[..., ch:ch + 1], one)
        if r in (45, 52):   # fused stage 2 (default for these radii) against the two-kernel form
            with _ffi.debug_options(gf_two_kernel=1):
                ok = ok and torch.equal(g3, rf.ops.guided_filter_u8(scene, grey3, r, eps,
                                                                    iterations=iters))
                ok = ok and torch.equal(c3, rf.ops.guided_filter_u8(scene, col, r, eps,
                                                                    iterations=iters))
            stats["gf_fused_vs_two_kernel"] = stats.get("gf_fused_vs_two_kernel", 0) + 1
        stats["gf_cases"] += 1
        stats["gf_pixels"] += n * h * w
        if not ok:
            stats["mismatches"].append(["gf", n, h, w, r, eps, iters])
    stats["seconds"] = args.seconds
    stats["seed"] = args.seed
    stats["n_mismatches"] = len(stats["mismatches"])
    stats["mismatches"] = stats["mismatches"][:10]      # a sample is enough to reproduce
    print(json.dumps(stats))
    return 1 if stats["n_mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
