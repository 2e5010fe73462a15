#!/usr/bin/env python
"""Freeze the oracle: fixtures F5-F7 of SURVEY.md section 8c.

    python tests/golden/make_filter_vectors.py          # writes filter_vectors.npz / .json

The joint-bilateral / guided-filter / CNN arithmetic is third-party (OpenCV-contrib, Caffe) and
absent from /root/reference, so these vectors are NOT reference outputs: they are the outputs of
the C restatement (T1, oracle/rf_oracle.c) at the commit that generated them, cross-checked at
generation time against the independent numpy restatement (T1', oracle/t1_numpy.py, bit for bit)
and the float64 definitions (T0, oracle/t0_numpy.py, distance recorded in the manifest).  Their
purpose is to stop the oracle from drifting together with the kernels: tests/test_golden_filters.py
requires today's oracle AND the HIP path to reproduce the committed bytes and SHA-256 digests.
Regenerating them is a deliberate act that shows up in the diff of filter_vectors.json.

  F5  joint bilateral: 48x64, sigma_s 22 (radius 33, 3,409 taps) and 28 (radius 42, 5,525 taps),
      all joint/src channel combinations
  F6  guided filter: 240x320 piecewise-constant guide, radius 45, eps 3 (grey and colour src,
      1 and 3 chained passes); self-guided radius 52, eps 7
  F7  known answers and images smaller than the radius (multi-bounce borders); CNN forward
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle import c_oracle as co        # noqa: E402
from oracle import t0_numpy as t0        # noqa: E402
from oracle import t1_numpy as t1        # noqa: E402
from tests import synth                  # noqa: E402
from reflectance_filtering_amd import weights as rf_weights  # noqa: E402


KEEP_FLOAT = ("gf_240x320_r45_e3_flat_grey", "cnn_32x32")   # others: SHA-256 only


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def cases():
    """name -> (kind, params, inputs...) ; inputs are stored so the fixture is self-contained."""
    out = {}
    sc = synth.scene_u8(48, 64, seed=501)
    gr = synth.reflectance_like_u8(48, 64, seed=502)
    sc2 = synth.scene_u8(48, 64, seed=503)
    # F5
    out["jbf_48x64_s22_rgbjoint_greysrc"] = ("jbf", dict(d=-1, sc=20.0, ss=22.0), sc, gr)
    out["jbf_48x64_s22_rgbjoint_rgbsrc"] = ("jbf", dict(d=-1, sc=20.0, ss=22.0), sc, sc2)
    out["jbf_48x64_s22_self_grey"] = ("jbf", dict(d=-1, sc=20.0, ss=22.0), gr.copy(), gr)
    out["jbf_48x64_s28_rgbjoint_rgbsrc"] = ("jbf", dict(d=-1, sc=15.0, ss=28.0), sc, sc2)
    out["jbf_48x64_s22_1chjoint_rgbsrc"] = ("jbf", dict(d=-1, sc=20.0, ss=22.0),
                                            np.ascontiguousarray(sc[:, :, 1]), sc2)
    out["jbf_48x64_s22_rgbjoint_1chsrc"] = ("jbf", dict(d=-1, sc=20.0, ss=22.0), sc,
                                            np.ascontiguousarray(gr[:, :, 0]))
    out["jbf_48x64_d9_s3"] = ("jbf", dict(d=9, sc=12.0, ss=3.0), sc, sc2)
    # F6
    flat = synth.flat_guide_u8(240, 320, seed=601)
    g2 = synth.reflectance_like_u8(240, 320, seed=602)
    c2 = synth.scene_u8(240, 320, seed=603)
    out["gf_240x320_r45_e3_flat_grey"] = ("gf", dict(radius=45, eps=3.0, iters=1), flat, g2)
    out["gf_240x320_r45_e3_flat_colour"] = ("gf", dict(radius=45, eps=3.0, iters=1), flat, c2)
    out["gf_240x320_r45_e3_flat_grey_x3"] = ("gf", dict(radius=45, eps=3.0, iters=3), flat, g2)
    out["gf_240x320_r52_e7_self"] = ("gf", dict(radius=52, eps=7.0, iters=1), c2.copy(), c2)
    out["gf_240x320_r45_e3_1chsrc"] = ("gf", dict(radius=45, eps=3.0, iters=1), flat,
                                       np.ascontiguousarray(g2[:, :, 0]))
    # F7: known answers + multi-bounce borders
    rng = np.random.default_rng(701)
    const = np.full((20, 30, 3), 77, np.uint8)
    noise = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    out["jbf_const_src"] = ("jbf", dict(d=-1, sc=20.0, ss=5.0), noise, const)
    out["jbf_const_joint"] = ("jbf", dict(d=-1, sc=20.0, ss=5.0), const, noise)
    out["gf_const_guide"] = ("gf", dict(radius=4, eps=3.0, iters=1), const, noise)
    for (h, w) in ((5, 7), (1, 9), (9, 1), (3, 3), (2, 40)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        b = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out["jbf_tiny_%dx%d_r33" % (h, w)] = ("jbf", dict(d=-1, sc=40.0, ss=22.0), a, b)
        out["gf_tiny_%dx%d_r45" % (h, w)] = ("gf", dict(radius=45, eps=3.0, iters=1), a, b)
    out["cnn_32x32"] = ("cnn", {}, synth.scene_u8(32, 32, seed=702), None)
    return out


def run_oracle(kind, p, a, b):
    if kind == "jbf":
        return {"out": co.joint_bilateral_filter(a, b, p["d"], p["sc"], p["ss"])}
    if kind == "gf":
        cur, qf = b, None
        for _ in range(p["iters"]):
            cur, qf = co.guided_filter(a, cur, p["radius"], p["eps"], return_float=True)
        return {"out": cur, "qf": qf}
    r, r8 = co.cnn_reflectance(a, rf_weights.load_weights())
    return {"out": r8, "r": r}


def main():
    arrays, manifest = {}, {"note": "outputs of oracle/rf_oracle.c (T1); see make_filter_vectors.py",
                            "cases": {}}
    for name, (kind, p, a, b) in cases().items():
        res = run_oracle(kind, p, a, b)
        entry = {"kind": kind, "params": p, "sha256": {k: sha(v) for k, v in res.items()}}
        # generation-time cross-checks against the two independent restatements
        if kind == "jbf" and a.shape[0] * a.shape[1] <= 48 * 64:
            t1o = t1.joint_bilateral_f32seq(a, b, p["sc"], p["ss"], d=p["d"])
            assert np.array_equal(t1o, res["out"]), name
            t0o, _ = t0.joint_bilateral_f64(a, b, p["sc"], p["ss"], d=p["d"])
            entry["max_abs_vs_float64"] = float(np.abs(res["out"].astype(np.float64) - t0o).max())
        if kind == "gf" and p["iters"] == 1 and a.shape[0] * a.shape[1] <= 240 * 320:
            t1u, t1q = t1.guided_filter_f32seq(a, b, p["radius"], p["eps"])
            assert np.array_equal(t1u, res["out"]), name
            assert np.array_equal(t1q, res["qf"]), name
            t0o = t0.guided_filter_f64(a, b, p["radius"], p["eps"])
            entry["max_abs_qf_vs_float64"] = float(np.abs(res["qf"].astype(np.float64) - t0o).max())
        # inputs are stored once (several cases share them); float results only where listed
        for tag, arr in (("a", a), ("b", b)):
            if arr is not None:
                key = "in/" + sha(arr)[:12]
                arrays[key] = arr
                entry[tag] = key
        for k, v in res.items():
            if v.dtype == np.uint8 or name in KEEP_FLOAT:
                arrays[name + "/" + k] = v
        manifest["cases"][name] = entry
        print(name, entry["sha256"]["out"][:16])
    np.savez_compressed(os.path.join(HERE, "filter_vectors.npz"), **arrays)
    with open(os.path.join(HERE, "filter_vectors.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
