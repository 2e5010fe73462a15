#!/usr/bin/env python
"""Extreme shapes on the GPU: one very large image, and a very long batch of tiny images.
Cross-checks: tiled vs untiled JBF, guided-filter grey path vs 1-channel src, CNN/colourise of a
large image against the same pixels processed as many small images (all per-pixel operators).
Prints one JSON line; exit code 1 on a mismatch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import _ffi
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    res = {}
    # ---- one 8192 x 6000 image
    h, w = 6000, 8192
    scene, grey = bench.synth_batch(torch, 1, h, w, 77, dev)
    a = rf.ops.joint_bilateral_u8(scene, grey, -1, 20.0, 22.0)
    crop = (slice(0, 1), slice(h - 700, h), slice(w - 900, w))
    # untiled kernel on the bottom-right corner with its halo (the filter is local)
    sub_j = scene[:, h - 800:, w - 1000:].contiguous()
    sub_s = grey[:, h - 800:, w - 1000:].contiguous()
    b = rf.ops.joint_bilateral_u8(sub_j, sub_s, -1, 20.0, 22.0, flags=_ffi.JBF_FORCE_GENERIC)
    res["jbf_big_corner"] = bool(torch.equal(a[crop], b[:, 100:, 100:]))
    del a, b, sub_j, sub_s
    g3 = rf.ops.guided_filter_u8(scene, grey, 45, 3.0)
    g1 = rf.ops.guided_filter_u8(scene, grey[..., :1].contiguous(), 45, 3.0)
    res["gf_big_grey_vs_1ch"] = bool(torch.equal(g3, g1.expand(-1, -1, -1, 3)))
    del g3, g1
    r, r8 = rf.get_reflectance_batch(scene)
    tiles = scene.reshape(1, 60, 100, 64, 128, 3).permute(0, 1, 3, 2, 4, 5).reshape(-1, 100, 128, 3).contiguous()
    rt, r8t = rf.get_reflectance_batch(tiles)            # 3840 small images
    back = rt.reshape(1, 60, 64, 100, 128).permute(0, 1, 3, 2, 4).reshape(1, h, w)
    res["cnn_big_vs_3840_small"] = bool(torch.equal(back, r)) and tiles.shape[0] == 3840
    del rt, r8t, back, tiles
    torch.cuda.empty_cache()
    # ---- 70,000 images of 8 x 8 (more than one grid.z / grid.y worth)
    n = 70000
    sc, gr = bench.synth_batch(torch, 1, 8 * 280, 8 * 250, 5, dev)
    small_j = sc.reshape(280, 8, 250, 8, 3).permute(0, 2, 1, 3, 4).reshape(n, 8, 8, 3).contiguous()
    small_s = gr.reshape(280, 8, 250, 8, 3).permute(0, 2, 1, 3, 4).reshape(n, 8, 8, 3).contiguous()
    a = rf.ops.joint_bilateral_u8(small_j, small_s, 5, 20.0, 3.0)
    idx = torch.tensor([0, 1, 65534, 65535, 65536, n - 1], device=dev)
    b = rf.ops.joint_bilateral_u8(small_j[idx].contiguous(), small_s[idx].contiguous(), 5, 20.0, 3.0)
    res["jbf_70000_small"] = bool(torch.equal(a[idx], b))
    a = rf.ops.guided_filter_u8(small_j, small_s, 2, 3.0)
    b = rf.ops.guided_filter_u8(small_j[idx].contiguous(), small_s[idx].contiguous(), 2, 3.0)
    res["gf_70000_small"] = bool(torch.equal(a[idx], b))
    # the fused stage 2 (radius 45) on the same 70,000 tiny images: 210,000 single-wave work items
    # whose every tap is a reflected border pixel; colour sources and, in a second call, grey ones
    a = rf.ops.guided_filter_u8(small_j, small_s, 45, 3.0, iterations=2)
    b = rf.ops.guided_filter_u8(small_j[idx].contiguous(), small_s[idx].contiguous(), 45, 3.0,
                                iterations=2)
    a2 = rf.ops.guided_filter_u8(small_s, small_j, 45, 3.0)
    b2 = rf.ops.guided_filter_u8(small_s[idx].contiguous(), small_j[idx].contiguous(), 45, 3.0)
    res["gf_fused_70000_small"] = bool(torch.equal(a[idx], b) and torch.equal(a2[idx], b2))
    r, _ = rf.get_reflectance_batch(small_j)
    refl, shad = rf.ops.colorize_srgb_u8(small_j[:60000].contiguous(), r[:60000].contiguous())
    refl2, shad2 = rf.ops.colorize_srgb_u8(small_j[idx[:2]].contiguous(), r[idx[:2]].contiguous())
    res["colorize_60000_small"] = bool(torch.equal(refl[idx[:2]], refl2) and torch.equal(shad[idx[:2]], shad2))
    print(json.dumps(res))
    return 0 if all(res.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
