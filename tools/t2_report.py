#!/usr/bin/env python
"""Pin the oracle on a real OpenCV in one command (tier T2).

    python tools/t2_report.py [--out profiles/t2_opencv.json] [--no-gpu]

On any machine with `opencv-contrib` (`import cv2; cv2.ximgproc`) this runs the reference's exact
calls (/root/reference/filter_reflectance.py:60-70)

    cv2.ximgproc.jointBilateralFilter(joint, image, -1, sigma_color, sigma_spatial)
    cv2.ximgproc.guidedFilter(guide=joint, src=image, radius=int(sigma_spatial), eps=sigma_color)

on the frozen inputs F5-F7 (tests/golden/filter_vectors.npz: the joint-bilateral and guided-filter
cases; guided cases chained `iters` times like the reference's 3x GF) and writes, per case, the
max-abs difference and the flip rate of the C oracle against OpenCV and - when a HIP device and
librf_hip.so are present - of the HIP path against OpenCV, next to the OpenCV version and a digest
of its build information.  All-zero rows upgrade the oracle from "parity unpinned" to pinned on
that build; anything else names the case and the build that disagreed.

Without OpenCV the tool says so, writes a report whose `opencv` field is null, and exits with
code 3.  Nothing of the reference is needed to run it: inputs are the committed fixtures.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def compare(got, want):
    diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
    return {"max_abs": int(diff.max()), "flip_rate": float((diff != 0).mean()),
            "bytes": int(diff.size)}


def opencv_result(cv2, entry, a, b):
    p = entry["params"]
    if entry["kind"] == "jbf":
        return cv2.ximgproc.jointBilateralFilter(a, b, p["d"], p["sc"], p["ss"])
    cur = b
    for _ in range(p["iters"]):
        cur = cv2.ximgproc.guidedFilter(guide=a, src=cur, radius=p["radius"], eps=p["eps"])
    return cur


def oracle_result(co, entry, a, b):
    p = entry["params"]
    if entry["kind"] == "jbf":
        return co.joint_bilateral_filter(a, b, p["d"], p["sc"], p["ss"])
    cur = b
    for _ in range(p["iters"]):
        cur = co.guided_filter(a, cur, p["radius"], p["eps"])
    return cur


def hip_result(rf, entry, a, b):
    p = entry["params"]
    if entry["kind"] == "jbf":
        return rf.ximgproc.jointBilateralFilter(a, b, p["d"], p["sc"], p["ss"])
    cur = b
    for _ in range(p["iters"]):
        cur = rf.ximgproc.guidedFilter(a, cur, p["radius"], p["eps"])
    return cur


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "t2_opencv.json"))
    ap.add_argument("--no-gpu", action="store_true")
    args = ap.parse_args(argv)
    with open(os.path.join(GOLDEN, "filter_vectors.json")) as fh:
        manifest = json.load(fh)["cases"]
    vectors = np.load(os.path.join(GOLDEN, "filter_vectors.npz"))
    report = {"opencv": None, "opencv_build_sha256": None, "threads": None, "hip_path": False,
              "inputs": "tests/golden/filter_vectors.npz (F5-F7)", "cases": [],
              "verdict": "parity unpinned: cv2.ximgproc is not importable on this machine"}
    try:
        import cv2
        cv2.ximgproc.jointBilateralFilter
        cv2.ximgproc.guidedFilter
    except (ImportError, AttributeError) as exc:
        cv2 = None
        report["import_error"] = repr(exc)
    rf = None
    if cv2 is not None and not args.no_gpu:
        try:
            import torch
            if torch.cuda.is_available():
                import reflectance_filtering_amd as rf_mod
                rf_mod._ffi.load_library()
                rf = rf_mod
        except Exception as exc:                    # noqa: BLE001 - the HIP column is optional
            report["hip_error"] = repr(exc)
    if cv2 is not None:
        from oracle import c_oracle as co
        report["opencv"] = cv2.__version__
        report["opencv_build_sha256"] = hashlib.sha256(cv2.getBuildInformation().encode()).hexdigest()
        report["threads"] = int(cv2.getNumThreads())
        report["hip_path"] = rf is not None
        worst = 0
        for name in sorted(manifest):
            entry = manifest[name]
            if entry["kind"] not in ("jbf", "gf"):
                continue
            a = vectors[entry["a"]]
            b = vectors[entry["b"]]
            row = {"case": name, "kind": entry["kind"], "params": entry["params"]}
            try:
                want = opencv_result(cv2, entry, a, b)
            except cv2.error as exc:               # e.g. a channel combination this build refuses
                row["opencv_error"] = str(exc).strip().splitlines()[-1]
                report["cases"].append(row)
                continue
            want = want.reshape(vectors[name + "/out"].shape)
            row["oracle_vs_opencv"] = compare(oracle_result(co, entry, a, b).reshape(want.shape), want)
            row["frozen_vector_vs_opencv"] = compare(vectors[name + "/out"], want)
            worst = max(worst, row["oracle_vs_opencv"]["max_abs"])
            if rf is not None:
                row["hip_vs_opencv"] = compare(np.asarray(hip_result(rf, entry, a, b)).reshape(want.shape), want)
                worst = max(worst, row["hip_vs_opencv"]["max_abs"])
            report["cases"].append(row)
        compared = [r for r in report["cases"] if "oracle_vs_opencv" in r]
        report["worst_max_abs"] = worst
        report["verdict"] = ("pinned: %d cases byte-identical to OpenCV %s" % (len(compared), cv2.__version__)
                             if compared and worst == 0 else
                             "differs from OpenCV %s by up to %d grey levels - see the cases"
                             % (cv2.__version__, worst))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(report, fh, indent=1, sort_keys=True)
    print(report["verdict"])
    print("wrote", args.out)
    return 0 if cv2 is not None else 3


if __name__ == "__main__":
    sys.exit(main())
