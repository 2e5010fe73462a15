"""T2: the oracle and the HIP path against a REAL cv2.ximgproc, wherever one is
installed.  Skips cleanly without OpenCV-contrib (it is absent from the build container and from
the GPU boxes of this project; the reference hints at OpenCV 3.1.0,
/root/reference/filter_reflectance.py:37-43).

The calls are exactly the reference's (/root/reference/filter_reflectance.py:60-70):
    cv2.ximgproc.jointBilateralFilter(joint, image, -1, sigma_color, sigma_spatial)
    cv2.ximgproc.guidedFilter(guide=joint, src=image, radius=int(sigma_spatial), eps=sigma_color)
What is asserted: the tolerance BASELINE.json states for the filters (1e-4 max-abs on [0,1]-scaled
data is below one uint8 step, so the bytes must agree) -- reported with max-abs, flip rate, the
OpenCV version and a digest of its build information, so that a disagreement says which build
disagreed.  A passing run on any machine is what upgrades the oracle from "unpinned" to "pinned".
"""
import hashlib
import json

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")
if not hasattr(cv2, "ximgproc") or not hasattr(cv2.ximgproc, "jointBilateralFilter"):
    pytest.skip("cv2 has no ximgproc (opencv-contrib not installed)", allow_module_level=True)

from oracle import c_oracle as co   # noqa: E402
from tests import synth             # noqa: E402


def _report(tag, got, want):
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    rec = {"case": tag, "max_abs": int(diff.max()), "flip_rate": float((diff != 0).mean()),
           "opencv": cv2.__version__,
           "build_sha256": hashlib.sha256(cv2.getBuildInformation().encode()).hexdigest()[:16]}
    print("T2 " + json.dumps(rec))
    return rec


CASES_JBF = [(48, 64, 20.0, 22.0), (96, 160, 20.0, 22.0), (64, 80, 15.0, 28.0), (40, 56, 7.5, 3.3)]
CASES_GF = [(256, 256, 45, 3.0), (256, 256, 52, 7.0), (240, 320, 45, 3.0), (130, 517, 9, 3.0)]


@pytest.mark.parametrize("h,w,sc,ss", CASES_JBF)
def test_oracle_jbf_equals_opencv(h, w, sc, ss):
    joint = synth.scene_u8(h, w, seed=h + w)
    for src in (synth.reflectance_like_u8(h, w, seed=h * w), synth.scene_u8(h, w, seed=7)):
        ref = cv2.ximgproc.jointBilateralFilter(joint, src, -1, sc, ss)
        rec = _report("jbf %dx%d c%g s%g" % (h, w, sc, ss), co.joint_bilateral_filter(joint, src, -1, sc, ss), ref)
        assert rec["max_abs"] == 0, rec


@pytest.mark.parametrize("h,w,r,eps", CASES_GF)
def test_oracle_gf_equals_opencv(h, w, r, eps):
    guide = synth.flat_guide_u8(h, w, seed=h) if r == 45 else synth.scene_u8(h, w, seed=h)
    for src in (synth.reflectance_like_u8(h, w, seed=w), synth.scene_u8(h, w, seed=9)):
        ref = cv2.ximgproc.guidedFilter(guide=guide, src=src, radius=r, eps=eps)
        rec = _report("gf %dx%d r%d eps%g" % (h, w, r, eps), co.guided_filter(guide, src, r, eps), ref)
        # OpenCV builds with FMA contraction or IPP box filters may differ in the last float bit
        # of q; a differing byte needs q within float rounding of .5, so flips must be rare and 1 LSB
        assert rec["max_abs"] <= 1 and rec["flip_rate"] < 1e-3, rec


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,sc,ss", CASES_JBF[:2])
def test_hip_jbf_equals_opencv(built, h, w, sc, ss):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import reflectance_filtering_amd as rf
    joint = synth.scene_u8(h, w, seed=h + w)
    src = synth.reflectance_like_u8(h, w, seed=h * w)
    ref = cv2.ximgproc.jointBilateralFilter(joint, src, -1, sc, ss)
    rec = _report("hip jbf %dx%d" % (h, w), rf.ximgproc.jointBilateralFilter(joint, src, -1, sc, ss), ref)
    assert rec["max_abs"] == 0, rec


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,r,eps", CASES_GF[:2])
def test_hip_gf_equals_opencv(built, h, w, r, eps):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import reflectance_filtering_amd as rf
    guide = synth.flat_guide_u8(h, w, seed=h) if r == 45 else synth.scene_u8(h, w, seed=h)
    src = synth.reflectance_like_u8(h, w, seed=w)
    ref = cv2.ximgproc.guidedFilter(guide=guide, src=src, radius=r, eps=eps)
    rec = _report("hip gf %dx%d r%d" % (h, w, r), rf.ximgproc.guidedFilter(guide, src, r, eps), ref)
    assert rec["max_abs"] <= 1 and rec["flip_rate"] < 1e-3, rec
