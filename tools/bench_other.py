#!/usr/bin/env python
"""Timing of the other two operators on the path (guided filter, CNN) and of the chained
configurations of BASELINE.json (C3: CNN + BF(CNN,CNN) at IIW size; C5: 3x GF at 4K).

    python tools/bench_other.py [--gf-batch 8] [--cnn-batch 256]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(torch, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gf-batch", type=int, default=32)
    ap.add_argument("--cnn-batch", type=int, default=256)
    args = ap.parse_args()
    import torch
    import bench
    import reflectance_filtering_amd as rf
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {}

    # C5: 3x guided filter c=3 s=45 at 3840x2160, piecewise-constant guidance
    n, h, w = args.gf_batch, 2160, 3840
    scene, grey = bench.synth_batch(torch, n, h, w, 5000, dev)
    flat = bench.flat_guide(scene)  # seeded Voronoi cells of flat colour
    dst = torch.empty_like(grey)
    # grey src (the CNN map the reference filters; three equal channels -> one-channel path)
    # and a colour src (all three channels computed)
    for tag, src in (("", grey), ("_colour_src", scene)):
        for iters in (1, 3):
            ms = timed(torch, lambda: rf.ops.guided_filter_u8(flat, src, 45, 3.0,
                                                              iterations=iters, out=dst))
            out["gf_4k_x%d%s" % (iters, tag)] = {"ms": ms, "batch": n,
                                                 "mp_per_s": n * h * w / 1e6 / (ms * 1e-3)}
    del scene, grey, flat, dst
    torch.cuda.empty_cache()

    # Joint bilateral beside the headline parameter set (c20 s22 -> radius 33, 3,409 taps, the 144-texel
    # row pitch): the reference's other published set c15 s28 with a flat joint and a colour src
    # (/root/reference/README.md:64: radius 42, 5,525 taps, the 176-texel pitch whose colour tiles run
    # as 32/16/8-row passes of 8-byte texels), the same with the grey map as src, the headline radius
    # with the same inputs, and a radius beyond the LDS tile (sigma_s 36 -> radius 54: generic kernel).
    # taps/s next to MP/s, so that loops of different radii can be compared.
    def taps_of(radius):
        return sum(1 for i in range(-radius, radius + 1) for j in range(-radius, radius + 1)
                   if (i * i + j * j) ** 0.5 <= radius)

    nj, hj, wj = 32, 1080, 1920
    scene, grey = bench.synth_batch(torch, nj, hj, wj, 5005, dev)
    flatj = bench.flat_guide(scene)
    colour_src = scene.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
    dstj = torch.empty_like(scene)
    # (radius 53..132 run in tap-row slabs since round 6; the *_generic_* / *_bands_* tags keep their
    #  round-4 / round-5 names - they now time that form - and "*_untiled_*" is the one-thread-per-pixel
    #  kernel those radii fell to before, forced through RF_JBF_FORCE_GENERIC on one image)
    from reflectance_filtering_amd import _ffi
    for tag, joint, src, sc, ss, fl in (("jbf_c15s28_flat_colour", flatj, colour_src, 15.0, 28.0, 0),
                                        ("jbf_c15s28_flat_grey", flatj, grey, 15.0, 28.0, 0),
                                        ("jbf_c20s22_flat_colour", flatj, colour_src, 20.0, 22.0, 0),
                                        ("jbf_c20s22_flat_grey", flatj, grey, 20.0, 22.0, 0),
                                        ("jbf_c20s36_generic_colour", flatj[:8], colour_src[:8], 20.0, 36.0, 0),
                                        ("jbf_c20s36_generic_grey", flatj[:8], grey[:8], 20.0, 36.0, 0),
                                        ("jbf_c20s40_bands_colour", flatj[:8], colour_src[:8], 20.0, 40.0, 0),
                                        ("jbf_c20s40_bands_grey", flatj[:8], grey[:8], 20.0, 40.0, 0),
                                        ("jbf_c20s36_untiled_colour", flatj[:1], colour_src[:1], 20.0, 36.0,
                                         _ffi.JBF_FORCE_GENERIC),
                                        ("jbf_c20s47_bands_grey", flatj[:8], grey[:8], 20.0, 47.0, 0),
                                        # (round 6: radius 73..132 in tap-row slabs; "*_untiled_*" forces the
                                        #  one-thread-per-pixel kernel they ran on before)
                                        ("jbf_c20s51_slabs_grey", flatj[:4], grey[:4], 20.0, 51.0, 0),
                                        ("jbf_c20s51_slabs_colour", flatj[:4], colour_src[:4], 20.0, 51.0, 0),
                                        ("jbf_c20s51_untiled_grey", flatj[:1], grey[:1], 20.0, 51.0,
                                         _ffi.JBF_FORCE_GENERIC),
                                        ("jbf_c20s66_slabs_grey", flatj[:4], grey[:4], 20.0, 66.0, 0),
                                        ("jbf_c20s88_slabs_grey", flatj[:4], grey[:4], 20.0, 88.0, 0),
                                        ("jbf_c20s88_untiled_grey", flatj[:1], grey[:1], 20.0, 88.0,
                                         _ffi.JBF_FORCE_GENERIC),
                                        ("jbf_c20s133_slabs_grey", flatj[:2], grey[:2], 20.0, 133.0, 0),
                                        ("jbf_c20s133_slabs_colour", flatj[:1], colour_src[:1], 20.0, 133.0, 0),
                                        ("jbf_c20s300_slabs_grey", flatj[:1], grey[:1], 20.0, 300.0, 0)):
        radius = int(round(1.5 * ss))
        nb_ = joint.shape[0]
        d_ = dstj[:nb_]
        ms = timed(torch, lambda: rf.ops.joint_bilateral_u8(joint, src, -1, sc, ss, out=d_, flags=fl), reps=2)
        mp = nb_ * hj * wj / 1e6
        out[tag] = {"ms": ms, "batch": nb_, "radius": radius, "taps_per_px": taps_of(radius),
                    "mp_per_s": mp / (ms * 1e-3),
                    "gtaps_per_s": mp * 1e6 * taps_of(radius) / (ms * 1e-3) / 1e9}
    del scene, grey, flatj, colour_src, dstj
    torch.cuda.empty_cache()

    # CV_32F variants (no BASELINE config uses them): 1080p, [0,1] data
    nf, hf, wf = 4, 1080, 1920
    sc_u8, gr_u8 = bench.synth_batch(torch, nf, hf, wf, 5003, dev)
    jf = sc_u8.float().div_(255.0).contiguous()
    sf = gr_u8.float().div_(255.0).contiguous()
    ms = timed(torch, lambda: rf.ops.guided_filter_f32(jf, sf, 45, 3.0 / 255 ** 2), reps=2)
    out["gf_f32_1080p"] = {"ms": ms, "mp_per_s": nf * hf * wf / 1e6 / (ms * 1e-3), "batch": nf}
    ms = timed(torch, lambda: rf.ops.joint_bilateral_f32(jf[:1], sf[:1], -1, 20 / 255.0, 22.0), reps=2)
    out["jbf_f32_1080p"] = {"ms": ms, "mp_per_s": hf * wf / 1e6 / (ms * 1e-3), "batch": 1,
                            "note": "register-tiled kernel, includes the host round trip for the "
                                    "value range and the table upload"}
    del sc_u8, gr_u8, jf, sf
    torch.cuda.empty_cache()

    # IIW-size guided filter with the reference's two parameter sets (the README's "0.08 s per
    # image" case): 256 x 500x333, self-guided c7 s52 and flat-guided c3 s45 on the grey CNN map
    n, h, w = args.cnn_batch, 333, 500
    scene, grey = bench.synth_batch(torch, n, h, w, 5004, dev)
    flat = bench.flat_guide(scene)
    dst = torch.empty_like(grey)
    for tag, guide, src, radius, eps in (("gf_iiw_c7s52_self", scene, scene, 52, 7.0),
                                        ("gf_iiw_c3s45_flat_grey", flat, grey, 45, 3.0)):
        d2 = torch.empty_like(src)
        ms = timed(torch, lambda: rf.ops.guided_filter_u8(guide, src, radius, eps, out=d2))
        out[tag] = {"ms": ms, "batch": n, "mp_per_s": n * h * w / 1e6 / (ms * 1e-3),
                    "ms_per_image": ms / n}
    del scene, grey, flat, dst, d2
    torch.cuda.empty_cache()

    # C1-size guided filter, single image latency
    scene, grey = bench.synth_batch(torch, 1, 256, 256, 5001, dev)
    ms = timed(torch, lambda: rf.ops.guided_filter_u8(scene, grey, 52, 7.0))
    out["gf_256_single"] = {"ms": ms}
    # the same call, and the 3x chain, replayed from a HIP graph (launch overhead of 6 / 16 kernels)
    o1 = torch.empty_like(grey)
    ws = rf.ops.gf_workspace(1, 256, 256, 3, 52, dev, torch)
    for iters in (1, 3):
        cap = rf.ops.CapturedCall(lambda: rf.ops.guided_filter_u8(scene, grey, 52, 7.0,
                                                                   iterations=iters, out=o1,
                                                                   workspace=ws))
        eager = timed(torch, lambda: rf.ops.guided_filter_u8(scene, grey, 52, 7.0, iterations=iters,
                                                             out=o1, workspace=ws))
        out["gf_256_single_x%d_graph" % iters] = {"ms": timed(torch, cap.replay), "eager_ms": eager}

    # C3: CNN + BF(CNN,CNN) at IIW size, landscape 500x333 and portrait 333x500
    for tag, (h, w) in (("iiw", (333, 500)), ("iiw_portrait", (500, 333))):
        n = args.cnn_batch
        scene, _ = bench.synth_batch(torch, n, h, w, 5002, dev)
        mp = n * h * w / 1e6
        if tag == "iiw":
            # a 3 ms launch after an idle gap runs at a clock that has not ramped: 25 launches
            ms_cnn = timed(torch, lambda: rf.get_reflectance_batch(scene), reps=25)
            out["cnn_iiw"] = {"ms": ms_cnn, "mp_per_s": mp / (ms_cnn * 1e-3), "batch": n}
            _, r8 = rf.get_reflectance_batch(scene)
            r3 = r8.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
            r3b, dst = r3.clone(), torch.empty_like(r3)
            ms_bf = timed(torch, lambda: rf.ops.joint_bilateral_u8(r3b, r3, -1, 20.0, 22.0,
                                                                   out=dst))
            out["bf_cnn_cnn_iiw"] = {"ms": ms_bf, "mp_per_s": mp / (ms_bf * 1e-3), "batch": n}
            del r3, r3b, dst
        if tag == "iiw":
            r, _ = rf.get_reflectance_batch(scene)
            ms_col = timed(torch, lambda: rf.ops.colorize_srgb_u8(scene, r))
            out["colorize_iiw"] = {"ms": ms_col, "mp_per_s": mp / (ms_col * 1e-3), "batch": n}
            del r
        # fused chain: u8 BGR -> CNN -> trunc*255 -> BF(CNN,CNN) -> u8, no host hand-off
        ms = timed(torch, lambda: rf.decompose_and_filter_batch(scene))
        out["c3_chain_" + tag] = {"ms": ms, "mp_per_s": mp / (ms * 1e-3), "batch": n,
                                  "ms_per_image": ms / n}
        del scene

    # roofline figures on ALGORITHMIC bytes against 8 TB/s, and the VALU floor
    # of the CNN (4,352 MACs/px as packed FMAs: 2 MACs per lane per 4 cycles, 1024 SIMDs, 2.4 GHz)
    def hbm(entry, bytes_per_px, passes=1):
        gbs = entry["mp_per_s"] * 1e6 * bytes_per_px * passes / 1e9
        entry["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                             "frac": gbs / 8000.0, "algorithmic_bytes_per_px": bytes_per_px * passes}
    for key, entry in out.items():
        if key.startswith("gf_4k_x1") or key.startswith("gf_iiw"):
            hbm(entry, 9)
        elif key.startswith("gf_4k_x3"):
            hbm(entry, 21)                    # guide read once per pass, u8 hand-offs: 3+3+3, +6, +6
        elif key == "cnn_iiw":
            hbm(entry, 8)                     # 3 in, 4 (float r) + 1 (byte r) out
            floor_gpx = 1024 * 2.4e9 / (4352 / 2 * 4 / 64) / 1e9
            entry["valu"] = {"bound": "packed-fma issue", "floor_gp_per_s": floor_gpx,
                             "frac": entry["mp_per_s"] / 1e3 / floor_gpx}
        elif key == "gf_f32_1080p":
            hbm(entry, 36)                    # 12 + 12 in, 12 out (float pixels)
        elif key == "jbf_f32_1080p":
            hbm(entry, 36)
        elif key.startswith("jbf_c"):
            hbm(entry, 9)
        elif key == "colorize_iiw":
            hbm(entry, 11)                    # 3 + 4 in, 3 + 1 out
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
