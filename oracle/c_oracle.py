"""ctypes front-end of oracle/librf_oracle.so (the C restatement, "T1").

TEST INFRASTRUCTURE -- see oracle/rf_oracle.c for what is restated and why parity
with OpenCV/Caffe themselves is unpinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librf_oracle.so")
_lib = None

BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = range(5)
BORDER_DEFAULT = BORDER_REFLECT_101
FLAG_TRUE_DIVISION = 1
CNN_NPARAMS = 4513
# rf_oracle.c RFO_VAR_*: the recalled choices a real OpenCV / Caffe could overturn (0 = the restatement
# the kernels follow); name -> bit, grouped by the operator each one affects
VARIANTS = {"jbf_true_division": 0x01, "jbf_fma": 0x02, "gf_diag_add_eps": 0x04,
            "gf_float_boxsum": 0x08, "gf_fma": 0x10, "cnn_sigmoid_tanh": 0x20,
            "cnn_gemm_no_fma": 0x40}
VARIANTS_OF = {"jbf": ("jbf_true_division", "jbf_fma"),
               "gf": ("gf_diag_add_eps", "gf_float_boxsum", "gf_fma"),
               "cnn": ("cnn_sigmoid_tanh", "cnn_gemm_no_fma")}


def build(force=False):
    """Compile librf_oracle.so with gcc (no-op if it is newer than the source)."""
    src = os.path.join(_HERE, "rf_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "librf_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int)
        L.rfo_set_variants.argtypes = [ctypes.c_uint]
        L.rfo_set_variants.restype = ctypes.c_uint
        L.rfo_get_variants.argtypes = []
        L.rfo_get_variants.restype = ctypes.c_uint
        L.rfo_border_interpolate.argtypes = [ctypes.c_int] * 3
        L.rfo_border_interpolate.restype = ctypes.c_int
        L.rfo_jbf_radius.argtypes = [ctypes.c_int, ctypes.c_double]
        L.rfo_jbf_radius.restype = ctypes.c_int
        L.rfo_jbf_taps.argtypes = [ctypes.c_int, ctypes.c_double, i32p, i32p, f32p]
        L.rfo_jbf_taps.restype = ctypes.c_int
        L.rfo_jbf_color_lut.argtypes = [ctypes.c_double, ctypes.c_int, f32p]
        L.rfo_jbf_color_lut.restype = None
        L.rfo_jbf_u8.argtypes = [u8p, u8p, u8p] + [ctypes.c_int] * 5 + [
            ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.rfo_jbf_u8.restype = ctypes.c_int
        L.rfo_gf_u8.argtypes = [u8p, u8p, u8p, f32p] + [ctypes.c_int] * 5 + [
            ctypes.c_double, ctypes.c_int]
        L.rfo_gf_u8.restype = ctypes.c_int
        L.rfo_jbf_f32.argtypes = [f32p, f32p, f32p] + [ctypes.c_int] * 5 + [
            ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.rfo_jbf_f32.restype = ctypes.c_int
        L.rfo_gf_f32.argtypes = [f32p, f32p, f32p] + [ctypes.c_int] * 5 + [
            ctypes.c_double, ctypes.c_int]
        L.rfo_gf_f32.restype = ctypes.c_int
        L.rfo_box_mean_f32.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.rfo_box_mean_f32.restype = None
        L.rfo_srgb_lut.argtypes = [f32p]
        L.rfo_srgb_lut.restype = None
        L.rfo_cnn_reflectance_u8.argtypes = [u8p, f32p, u8p, ctypes.c_int, ctypes.c_int, f32p,
                                             ctypes.c_int]
        L.rfo_cnn_reflectance_u8.restype = ctypes.c_int
        _lib = L
    return _lib


class variants:
    """``with variants("gf_fma", "gf_diag_add_eps"): ...`` runs the oracle with those recalled
    choices flipped (names of VARIANTS, or a mask); the previous mask is restored on exit."""

    def __init__(self, *names):
        self.mask = 0
        for n in names:
            self.mask |= n if isinstance(n, int) else VARIANTS[n]

    def __enter__(self):
        self.prev = lib().rfo_set_variants(self.mask)
        return self

    def __exit__(self, *exc):
        lib().rfo_set_variants(self.prev)
        return False


def variant_names(mask):
    return [n for n, b in VARIANTS.items() if mask & b]


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _as_hwc_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    return img


def border_interpolate(p, length, border):
    return lib().rfo_border_interpolate(int(p), int(length), int(border))


def jbf_radius(d, sigma_space):
    return lib().rfo_jbf_radius(int(d), float(sigma_space))


def jbf_taps(radius, sigma_space):
    n = (2 * radius + 1) ** 2
    di = np.zeros(n, np.int32)
    dj = np.zeros(n, np.int32)
    sw = np.zeros(n, np.float32)
    ip = ctypes.POINTER(ctypes.c_int)
    k = lib().rfo_jbf_taps(int(radius), float(sigma_space), di.ctypes.data_as(ip),
                           dj.ctypes.data_as(ip), _f32(sw))
    return di[:k].copy(), dj[:k].copy(), sw[:k].copy()


def jbf_color_lut(sigma_color, joint_cn):
    lut = np.zeros(256 * joint_cn, np.float32)
    lib().rfo_jbf_color_lut(float(sigma_color), int(joint_cn), _f32(lut))
    return lut


def joint_bilateral_filter(joint, src, d, sigma_color, sigma_space, border=BORDER_DEFAULT,
                           flags=0, threads=0):
    """cv2.ximgproc.jointBilateralFilter(joint, src, d, sigmaColor, sigmaSpace) on uint8."""
    j = _as_hwc_u8(joint)
    s = _as_hwc_u8(src)
    if j.shape[:2] != s.shape[:2]:
        raise ValueError("joint and src sizes differ")
    h, w = s.shape[:2]
    out = np.empty_like(s)
    rc = lib().rfo_jbf_u8(_u8(j), _u8(s), _u8(out), h, w, j.shape[2], s.shape[2], int(d),
                          float(sigma_color), float(sigma_space), int(border), int(flags),
                          int(threads))
    if rc != 0:
        raise ValueError("rfo_jbf_u8 rejected its arguments")
    return out if np.ndim(src) == 3 else out[:, :, 0]


def _as_hwc_f32(img):
    img = np.ascontiguousarray(img, dtype=np.float32)
    if img.ndim == 2:
        img = img[:, :, None]
    return img


def joint_bilateral_filter_f32(joint, src, d, sigma_color, sigma_space, border=BORDER_DEFAULT,
                               threads=0):
    """cv2.ximgproc.jointBilateralFilter on float32 images (interpolated colour table)."""
    j = _as_hwc_f32(joint)
    s = _as_hwc_f32(src)
    if j.shape[:2] != s.shape[:2]:
        raise ValueError("joint and src sizes differ")
    h, w = s.shape[:2]
    out = np.empty_like(s)
    rc = lib().rfo_jbf_f32(_f32(j), _f32(s), _f32(out), h, w, j.shape[2], s.shape[2], int(d),
                           float(sigma_color), float(sigma_space), int(border), int(threads))
    if rc == -2:
        raise NotImplementedError("constant joint image (OpenCV falls back to a Gaussian blur)")
    if rc != 0:
        raise ValueError("rfo_jbf_f32 rejected its arguments")
    return out if np.ndim(src) == 3 else out[:, :, 0]


def guided_filter_f32(guide, src, radius, eps, threads=0):
    """cv2.ximgproc.guidedFilter(guide, src, radius, eps) on float32 (3-channel guide)."""
    g = _as_hwc_f32(guide)
    s = _as_hwc_f32(src)
    if g.shape[:2] != s.shape[:2]:
        raise ValueError("guide and src sizes differ")
    h, w = s.shape[:2]
    out = np.empty_like(s)
    rc = lib().rfo_gf_f32(_f32(g), _f32(s), _f32(out), h, w, g.shape[2], s.shape[2], int(radius),
                          float(eps), int(threads))
    if rc != 0:
        raise ValueError("rfo_gf_f32 rejected its arguments")
    return out if np.ndim(src) == 3 else out[:, :, 0]


def guided_filter(guide, src, radius, eps, threads=0, return_float=False):
    """cv2.ximgproc.guidedFilter(guide, src, radius, eps) on uint8 (3-channel guide)."""
    g = _as_hwc_u8(guide)
    s = _as_hwc_u8(src)
    if g.shape[:2] != s.shape[:2]:
        raise ValueError("guide and src sizes differ")
    h, w = s.shape[:2]
    out = np.empty_like(s)
    qf = np.empty(s.shape, np.float32)
    rc = lib().rfo_gf_u8(_u8(g), _u8(s), _u8(out), _f32(qf), h, w, g.shape[2], s.shape[2],
                         int(radius), float(eps), int(threads))
    if rc != 0:
        raise ValueError("rfo_gf_u8 rejected its arguments")
    if np.ndim(src) == 2:
        out, qf = out[:, :, 0], qf[:, :, 0]
    return (out, qf) if return_float else out


def box_mean_f32(plane, radius):
    p = np.ascontiguousarray(plane, dtype=np.float32)
    out = np.empty_like(p)
    lib().rfo_box_mean_f32(_f32(p), _f32(out), p.shape[0], p.shape[1], int(radius))
    return out


def srgb_lut():
    lut = np.zeros(256, np.float32)
    lib().rfo_srgb_lut(_f32(lut))
    return lut


def cnn_reflectance(bgr_u8, weights, threads=0):
    """Returns (r float32 [H,W], r_u8 uint8 [H,W]) for one uint8 BGR image."""
    img = np.ascontiguousarray(bgr_u8, dtype=np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3
    wts = np.ascontiguousarray(weights, dtype=np.float32).ravel()
    assert wts.size == CNN_NPARAMS
    h, w = img.shape[:2]
    r = np.empty((h, w), np.float32)
    r8 = np.empty((h, w), np.uint8)
    rc = lib().rfo_cnn_reflectance_u8(_u8(img), _f32(r), _u8(r8), h, w, _f32(wts), int(threads))
    if rc != 0:
        raise ValueError("rfo_cnn_reflectance_u8 rejected its arguments")
    return r, r8
