#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING the reference's own Python helpers.

Run in the build container only (needs /root/reference; nothing here travels to
the GPU box except the .npz/.json outputs):

    python tests/golden/make_golden.py

The reference imports ``cv2`` and ``caffe``, which are not installed anywhere we
can reach, so both are replaced by recording stand-ins *for the purpose of
capturing the reference's own Python-side behaviour* (argument plumbing, colour
maths, dtype handling, file naming).  No filter arithmetic comes out of this
script: cv2.ximgproc / caffe.Net are third-party and absent, which is why the
oracle header says "parity unpinned" for them.

Fixtures (ids follow SURVEY.md section 8c):
  F1 colour_tables.npz      srgb_to_rgb / rgb_to_srgb on all 256 byte levels (+ float cases)
  F2 normalize_colorize.npz normalize (both branches), colorize, imwrite's uint8 conversion
  F3 caffe_blob.npz         imgCV2_to_caffeBlob on seeded uint8 BGR; gray-blob shape check
  F4 cnn_forward.npz        decoded weights, get_reflectance_caffe(FakeNet) on two inputs
  F8 cli_plumbing.json      what apply_filter / read_filter_write / decompose_image hand to
     decompose_outputs.npz  cv2.ximgproc, cv2.imwrite and caffe (recorded calls + arrays)
  F9 colorize_write.npz     colorize + imwrite(sRGB=True) bytes on larger inputs (pins the device
                            colourise path, rf_colorize_srgb_u8)
  F10 whdr.npz              training/layers/whdr_layer.py: pixel coordinates and whdr() values
                            (pins whdr.py and rf_whdr_f32)
"""
import hashlib
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from reflectance_filtering_amd import weights as rf_weights  # noqa: E402

warnings.simplefilter("ignore", DeprecationWarning)

# --------------------------------------------------------------------------- stubs
calls = []          # every call the reference makes into the fake cv2 / caffe
files_in = {}       # filename -> uint8 BGR array served by the fake cv2.imread
files_out = {}      # filename -> array handed to the fake cv2.imwrite

cv2 = types.ModuleType("cv2")


def _imread(filename):
    calls.append(["cv2.imread", filename])
    return files_in.get(filename)


def _imwrite(filename, image):
    calls.append(["cv2.imwrite", filename, str(image.dtype), list(image.shape)])
    if "/nonexistent/" in filename:
        return False
    files_out[filename] = np.array(image, copy=True)
    return True


cv2.imread = _imread
cv2.imwrite = _imwrite
ximgproc = types.ModuleType("cv2.ximgproc")


def _jbf(*args, **kwargs):
    calls.append(["cv2.ximgproc.jointBilateralFilter",
                  [a.tobytes()[:4].hex() if isinstance(a, np.ndarray) else a for a in args],
                  {k: (v.tobytes()[:4].hex() if isinstance(v, np.ndarray) else v)
                   for k, v in kwargs.items()}])
    return np.full_like(args[1], 7)


def _gf(*args, **kwargs):
    calls.append(["cv2.ximgproc.guidedFilter",
                  [a.tobytes()[:4].hex() if isinstance(a, np.ndarray) else a for a in args],
                  {k: (v.tobytes()[:4].hex() if isinstance(v, np.ndarray) else v)
                   for k, v in kwargs.items()}])
    return np.full_like(kwargs["src"], 9)


ximgproc.jointBilateralFilter = _jbf
ximgproc.guidedFilter = _gf
cv2.ximgproc = ximgproc
sys.modules["cv2"] = cv2
sys.modules["cv2.ximgproc"] = ximgproc

WEIGHTS = rf_weights.load_weights(os.path.join(REF, "learned_weights.caffemodel"))


class _Blob(object):
    def __init__(self, shape):
        self.data = np.zeros(shape, np.float32)

    def reshape(self, *shape):
        calls.append(["blob.reshape", list(shape)])
        self.data = np.zeros(shape, np.float32)


def _forward_f64(x_nchw_f32, wts):
    """Float64 forward of the shipped architecture on the float32 input blob."""
    w = wts.astype(np.float64)
    n, c, hh, ww = x_nchw_f32.shape
    x = x_nchw_f32.astype(np.float64).transpose(0, 2, 3, 1).reshape(-1, 3)
    cur = np.maximum(x @ w[:96].reshape(32, 3).T + w[96:128], 0)
    cat = [cur]
    q = 128
    for _ in range(4):
        cur = np.maximum(cur @ w[q:q + 1024].reshape(32, 32).T + w[q + 1024:q + 1056], 0)
        cat.append(cur)
        q += 1056
    z = np.concatenate(cat, 1) @ w[q:q + 160] + w[q + 160]
    return (1.0 / (1.0 + np.exp(-z))).reshape(n, 1, hh, ww).astype(np.float32)


class FakeNet(object):
    """Duck-typed caffe.Net: blobs['images'] / blobs['reflectance_intensity'], forward()."""

    def __init__(self, *args, **kwargs):
        calls.append(["caffe.Net", [os.path.basename(str(a)) if isinstance(a, str) else a
                                    for a in args],
                      {k: os.path.basename(str(v)) for k, v in kwargs.items()}])
        self.blobs = {"images": _Blob((1, 3, 256, 256)),
                      "reflectance_intensity": _Blob((1, 1, 256, 256))}

    def forward(self):
        calls.append(["net.forward"])
        self.blobs["reflectance_intensity"].data = _forward_f64(self.blobs["images"].data,
                                                                WEIGHTS)


caffe = types.ModuleType("caffe")
caffe.TEST = "TEST"
caffe.Net = FakeNet
caffe.Layer = object   # base class of the reference's python layers (training/layers/*.py)
sys.modules["caffe"] = caffe

sys.path.insert(0, REF)
import image_utils as ref_iu  # noqa: E402
import filter_reflectance as ref_fr  # noqa: E402
import decompose_with_trained_CNN as ref_dc  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


# ------------------------------------------------------------------------------- F1
levels = np.arange(256, dtype=np.float64) / 255.0
extra = np.array([0.0, 0.0031308, 0.0031309, 0.04045, 0.04046, 0.5, 1.0, 1.5, 2.0])
save("colour_tables.npz",
     levels=levels,
     srgb_to_rgb_levels=ref_iu.srgb_to_rgb(levels),
     rgb_to_srgb_levels=ref_iu.rgb_to_srgb(levels),
     extra=extra,
     srgb_to_rgb_extra=ref_iu.srgb_to_rgb(extra),
     rgb_to_srgb_extra=ref_iu.rgb_to_srgb(extra),
     srgb_to_rgb_levels_f32=ref_iu.srgb_to_rgb(levels.astype(np.float32)))

# ------------------------------------------------------------------------------- F2
rng = np.random.default_rng(20171201)
img_small = rng.random((16, 16, 3))                      # max <= 1: identity branch
img_big = rng.random((16, 16, 3)) * 300.0                # max > 1: percentile branch
img_big_f32 = img_big.astype(np.float32)
bgr16 = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
inten16 = (0.05 + 0.95 * rng.random((16, 16))).astype(np.float32)
refl16, shad16 = ref_iu.colorize(inten16, bgr16)
out = {}
for tag, arr, srgb in (("small_lin", img_small, False), ("small_srgb", img_small, True),
                       ("big_lin", img_big, False), ("big_srgb", img_big, True),
                       ("big32_lin", img_big_f32, False), ("refl_srgb", refl16, True),
                       ("shad_srgb", shad16, True), ("u8_passthrough", bgr16, False),
                       ("inten_lin", inten16, False)):
    fn = "/golden/%s.png" % tag
    ref_iu.imwrite(fn, arr, sRGB=srgb)
    out["imwrite_" + tag] = files_out[fn]
save("normalize_colorize.npz",
     img_small=img_small, img_big=img_big, img_big_f32=img_big_f32,
     normalize_small=ref_iu.normalize(img_small), normalize_big=ref_iu.normalize(img_big),
     normalize_big_f32=ref_iu.normalize(img_big_f32),
     bgr16=bgr16, inten16=inten16, colorize_reflectance=refl16, colorize_shading=shad16, **out)

# ------------------------------------------------------------------------------- F3
bgr8 = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
ramp = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)  # 1x256x3 grey ramp
blob8 = ref_dc.imgCV2_to_caffeBlob(bgr8)
blob_ramp = ref_dc.imgCV2_to_caffeBlob(ramp)
gray_ok = ref_dc.caffeBlob_to_imgGrayLinear(np.arange(12, dtype=np.float32).reshape(1, 1, 3, 4))
shape_errors = []
for shp in ((2, 1, 3, 4), (1, 3, 3, 4)):
    try:
        ref_dc.caffeBlob_to_imgGrayLinear(np.zeros(shp, np.float32))
        shape_errors.append("")
    except ValueError as exc:
        shape_errors.append(str(exc))
save("caffe_blob.npz", bgr8=bgr8, blob8=blob8, ramp=ramp, blob_ramp=blob_ramp,
     blob_ramp_f32=blob_ramp.astype(np.float32), gray_ok=gray_ok,
     shape_errors=np.array(shape_errors))

# ------------------------------------------------------------------------------- F4
bgr32 = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
net = FakeNet()
r32 = ref_dc.get_reflectance_caffe(net, bgr32)
blob32 = net.blobs["images"].data.copy()
net2 = FakeNet()
r_ramp = ref_dc.get_reflectance_caffe(net2, ramp)
with open(os.path.join(REF, "learned_weights.caffemodel"), "rb") as fh:
    sha = hashlib.sha256(fh.read()).hexdigest()
save("cnn_forward.npz", weights=WEIGHTS, bgr32=bgr32, blob32=blob32, r32=np.array(r32),
     ramp=ramp, r_ramp=np.array(r_ramp), caffemodel_sha256=np.array(sha))
np.save(os.path.join(REPO, "reflectance_filtering_amd", "data", "cnn_weights_f32.npy"), WEIGHTS)

# ------------------------------------------------------------------------------- F8
plumbing = {}
img_a = np.full((4, 5, 3), 0x11, np.uint8)
img_b = np.full((4, 5, 3), 0x22, np.uint8)
del calls[:]
res = ref_fr.apply_filter("bilateral", img_a, img_b, 20, 22)
plumbing["apply_filter_bilateral"] = {"calls": list(calls), "result_fill": int(res[0, 0, 0])}
del calls[:]
res = ref_fr.apply_filter("guided", img_a, img_b, 3.0, 45.9)
plumbing["apply_filter_guided"] = {"calls": list(calls), "result_fill": int(res[0, 0, 0])}
errors = {}
for key, args in (("sigma_color_zero", ("bilateral", img_a, img_b, 0, 22)),
                  ("sigma_spatial_negative", ("guided", img_a, img_b, 3, -1)),
                  ("bad_type", ("median", img_a, img_b, 3, 3))):
    try:
        ref_fr.apply_filter(*args)
        errors[key] = None
    except ValueError as exc:
        errors[key] = ["ValueError", str(exc)]
plumbing["apply_filter_errors"] = errors

files_in["/in/photo.final.png"] = img_a
files_in["/in/guide.png"] = img_b
names = {}
for tag, (ftype, sc, ss) in {"int_like": ("bilateral", 20.0, 22.0),
                             "fractional": ("guided", 3.5, 45.25),
                             "python_int": ("guided", 7, 52)}.items():
    del calls[:]
    files_out.clear()
    ref_fr.read_filter_write(ftype, "/in/photo.final.png", "/in/guide.png", sc, ss, "/out/dir")
    names[tag] = {"written": sorted(files_out), "calls": list(calls)}
plumbing["read_filter_write"] = names
errs = {}
try:
    ref_fr.read_filter_write("guided", "/in/missing.png", "/in/guide.png", 3.0, 45.0, "/out")
except Exception as exc:  # noqa: BLE001 - the reference raises bare Exception
    errs["unreadable"] = [type(exc).__name__, str(exc)]
try:
    ref_fr.read_filter_write("guided", "/in/photo.final.png", "/in/guide.png", 3.0, 45.0,
                             "/nonexistent/dir")
except Exception as exc:  # noqa: BLE001
    errs["unwritable"] = [type(exc).__name__, str(exc)]
plumbing["read_filter_write_errors"] = errs

scene = rng.integers(0, 256, (24, 20, 3), dtype=np.uint8)
files_in["/in/scene.01.jpg"] = scene
del calls[:]
files_out.clear()
r_scene = ref_dc.decompose_image("/in/scene.01.jpg", "/out/dir")
plumbing["decompose_image"] = {"written": sorted(files_out), "calls": list(calls)}
save("decompose_outputs.npz", scene=scene, r=np.array(r_scene),
     r_png=files_out["/out/dir/scene.01-r.png"],
     r_colorized_png=files_out["/out/dir/scene.01-r_colorized.png"],
     s_colorized_png=files_out["/out/dir/scene.01-s_colorized.png"])

# F9: colorize + imwrite(sRGB=True) on larger inputs (the device colourise path is pinned on these):
# a natural-looking image, a dark one (max <= 1: no normalisation), one with black pixels and a
# saturated region, a tiny one.  Inputs and the bytes the reference hands to cv2.imwrite.
rng9 = np.random.default_rng(909)
f9 = {}
for tag, (hh, ww) in {"natural": (48, 64), "dark": (20, 24), "holes": (33, 31), "tiny": (1, 3)}.items():
    yy, xx = np.mgrid[0:hh, 0:ww]
    base = 110 + 70 * np.sin(yy / 7.0) * np.cos(xx / 5.0)
    img = np.clip(base[:, :, None] + rng9.normal(0, 25, (hh, ww, 3)), 0, 255).astype(np.uint8)
    r = (0.1 + 0.85 * rng9.random((hh, ww))).astype(np.float32)
    if tag == "dark":
        img = (img // 128).astype(np.uint8)          # values 0/1
        r = np.full((hh, ww), 0.999, np.float32)     # refl = r * img / mean <= ~3; shading <= ~1
        img[:, :] = img[:, :, :1]                    # grey: reflectance = r <= 1, shading = 1/r > 1
    if tag == "holes":
        img[5:12, 3:20] = 0
        img[20:30, 10:25] = 255
        r[0, 0] = 1e-4
    files_out.clear()
    refl9, shad9 = ref_iu.colorize(r, img)
    ref_iu.imwrite("/out/f9-r_colorized.png", refl9, sRGB=True)
    ref_iu.imwrite("/out/f9-s_colorized.png", shad9, sRGB=True)
    f9[tag + "_image"] = img
    f9[tag + "_r"] = r
    f9[tag + "_refl_png"] = files_out["/out/f9-r_colorized.png"]
    f9[tag + "_shading_png"] = files_out["/out/f9-s_colorized.png"]
save("colorize_write.npz", **f9)

# F10: WHDR (training/layers/whdr_layer.py): pixel-coordinate extraction and the weighted
# disagreement rate on seeded predictions, 1 and 3 channels, incl. an image without comparisons
# and points whose lightness is floored at eps.
sys.path.insert(0, os.path.join(REF, "training", "layers"))
import whdr_layer as ref_whdr  # noqa: E402
rng10 = np.random.default_rng(1010)
f10 = {}
for tag, (cc, hh, ww, ncomp) in {"grey": (1, 40, 56, 300), "colour": (3, 33, 47, 257),
                                 "none": (1, 8, 8, 0), "single": (3, 5, 4, 1)}.items():
    refl = (0.02 + 0.98 * rng10.random((cc, hh, ww))).astype(np.float32)
    refl[:, 0, 0] = 0.0                      # lightness floored at eps
    refl[:, 1, 1] = refl[:, 2, 2] * np.float32(1.1)   # ratios right at 1 + delta
    blob = np.full((ncomp + 1, 6), np.nan)
    blob[:ncomp, 0:4] = rng10.random((ncomp, 4)) * 0.999
    if ncomp > 4:
        blob[0, 0:4] = [0.0, 0.0, 0.5, 0.5]
        blob[1, 0:4] = [1.5 / ww, 1.5 / hh, 2.5 / ww, 2.5 / hh]
        blob[2, 0:4] = [2.5 / ww, 2.5 / hh, 1.5 / ww, 1.5 / hh]
    blob[:ncomp, 4] = rng10.integers(0, 3, ncomp)
    blob[:ncomp, 5] = rng10.random(ncomp) * 2.0
    blob[ncomp, 0] = ncomp
    blob[ncomp, 1] = 12345.0
    blob[ncomp, 2] = 0
    px = ref_whdr._extract_valid_comparisons_with_actual_size(blob, hh, ww)
    f10[tag + "_refl"] = refl
    f10[tag + "_comparisons"] = blob[:ncomp].copy()
    f10[tag + "_pixels"] = px
    f10[tag + "_whdr"] = np.array([ref_whdr.whdr(refl, px, d) for d in (0.1, 0.0, 0.25)])
save("whdr.npz", **f10)

with open(os.path.join(HERE, "cli_plumbing.json"), "w") as fh:
    json.dump(plumbing, fh, indent=1, sort_keys=True, default=str)
print("wrote cli_plumbing.json")
