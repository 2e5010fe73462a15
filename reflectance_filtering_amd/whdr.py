"""WHDR (weighted human disagreement rate) of reflectance predictions against IIW judgements.

Mirrors the evaluation arithmetic of the reference's training layer
(/root/reference/training/layers/whdr_layer.py:180-196 `_lightness`,
:239-250 `_extract_valid_comparisons_with_actual_size`, :253-287 `whdr`) and the JSON reading of
/root/reference/training/createNumpyArrayWithComparisonsForIIW.py:320-347,613-645.

    comparisons[k] = (x1, y1, x2, y2, darker, weight)   x, y normalised to [0,1);
                     darker: 0 = 'E' (about equal), 1 = point 1 darker, 2 = point 2 darker

`whdr` is the per-image host form: one fancy-indexed gather and vectorised decisions over all
judgements of an image, with the dtype promotions the reference's scalar loop would go through
under the installed numpy (pinned on the reference's own values, tests/golden/whdr.npz);
`whdr_batch` evaluates many device-resident predictions in one launch (rf_whdr_f32).
"""
from __future__ import division, print_function

import json

import numpy as np

from . import _ffi

EPS = np.finfo(np.float32).eps  # lightness floor, whdr_layer.py:177
DARKER_CODE = {"1": 1, "2": 2, "E": 0}


def load_judgements(json_path):
    """IIW `<id>.json` -> float64 [n,6] rows (x1, y1, x2, y2, darker, darker_score) with
    normalised coordinates (createNumpyArrayWithComparisonsForIIW.py:320-347, 628-641)."""
    with open(json_path) as fh:
        data = json.load(fh)
    points = {p["id"]: (p["x"], p["y"]) for p in data["intrinsic_points"]}
    rows = []
    for c in data["intrinsic_comparisons"]:
        x1, y1 = points[c["point1"]]
        x2, y2 = points[c["point2"]]
        rows.append([x1, y1, x2, y2, DARKER_CODE[c["darker"]], c["darker_score"]])
    return np.array(rows, dtype=np.float64).reshape(-1, 6)


def to_pixels(comparisons, height, width):
    """Normalised -> pixel coordinates (x * width, y * height, truncated toward zero), stored
    back in the array's own dtype; the judgement and weight columns are untouched
    (contract: whdr_layer.py:239-250)."""
    px = np.array(comparisons, copy=True)
    # the products are formed in the array's own dtype (float32 blobs stay float32: for a
    # coordinate k / width the float32 product rounds to k, the float64 one truncates to k - 1)
    for cols, size in (([0, 2], int(width)), ([1, 3], int(height))):
        px[:, cols] = np.trunc(px[:, cols] * size)
    return px


def _point_lightness(reflectance, ys, xs):
    """Lightness of the pixels (ys[i], xs[i]) of a [c,h,w] image, one gather for all points:
    the channel mean (3 channels, in the image's dtype like np.mean of one pixel) or the value
    itself (1 channel), floored at EPS.  A NaN fails `> EPS` and becomes EPS, as it does through
    Python's max() in the reference (whdr_layer.py:180-196)."""
    channels = reflectance.shape[0]
    if channels not in (1, 3):
        raise Exception("Expecting 1 or 3 channels to compute lightness!")
    picked = reflectance[:, ys, xs]                       # [c, n]
    value = picked[0] if channels == 1 else picked.mean(axis=0)
    return np.where(value > EPS, value, EPS)


def whdr(reflectance, comparisons, delta=0.1):
    """WHDR of one [c,h,w] reflectance image against judgements in pixel coordinates: the
    weight of the judgements the image disagrees with over the weight of all of them
    (arithmetic contract: whdr_layer.py:253-287).  No judgements -> 0.0.

    All judgements are decided at once: the image says "point 1 darker" (1) when l2/l1 exceeds
    1 + delta, else "point 2 darker" (2) when l1/l2 does, else "about equal" (0).  The two
    weight totals are running sums in judgement order (np.cumsum adds sequentially) in the
    dtype `0.0 + weight` has under the installed numpy, and the threshold is compared in the
    dtype `lightness ratio > python float` is compared in under it - the same promotions the
    reference's scalar loop goes through."""
    reflectance = np.asarray(reflectance)
    judgements = np.asarray(comparisons)
    if judgements.shape[0] == 0:
        if reflectance.shape[0] not in (1, 3):
            raise Exception("Expecting 1 or 3 channels to compute lightness!")
        return 0.0
    pts = judgements[:, :5].astype(int)
    first = _point_lightness(reflectance, pts[:, 1], pts[:, 0])
    second = _point_lightness(reflectance, pts[:, 3], pts[:, 2])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        up, down = second / first, first / second
    # scalar-vs-python-float promotion of the installed numpy (float32 under NumPy >= 2)
    cmp_t = (first.dtype.type(1) * 1.5).dtype
    limit = cmp_t.type(1 + delta)
    verdict = np.where(up.astype(cmp_t) > limit, 1, np.where(down.astype(cmp_t) > limit, 2, 0))
    weights = judgements[:, 5]
    acc_t = (0.0 + weights.dtype.type(0)).dtype
    weights = weights.astype(acc_t)
    total = np.cumsum(weights)[-1]
    if not total:
        return 0.0
    wrong = np.cumsum(np.where(verdict != pts[:, 4], weights, acc_t.type(0)))[-1]
    return wrong / total


def whdr_batch(reflectances, comparisons_px, delta=0.1):
    """WHDR of N device-resident predictions in one launch.
    reflectances: CUDA float32 [N,C,H,W] (C = 1 or 3) or [N,H,W];
    comparisons_px: list of N arrays [n_i,6] in pixel coordinates (to_pixels), any n_i >= 0.
    Returns float64 [N] (host).  Lightness, ratios and the decision are float32 like the
    reference's blobs, compared against float32(1 + delta) (NumPy >= 2 scalar promotion; NumPy 1
    compared in float64, a difference confined to ratios within one float32 ulp above
    1 + delta); the weighted sums are float64, accumulated in comparison order."""
    torch = _ffi.require_gpu()
    lib = _ffi.load_library()
    r = reflectances
    if not (torch.is_tensor(r) and r.is_cuda and r.dtype == torch.float32 and r.is_contiguous()
            and r.dim() in (3, 4)):
        raise ValueError("reflectances must be a contiguous CUDA float32 tensor [N,C,H,W] or [N,H,W]")
    if r.dim() == 3:
        r = r.unsqueeze(1)
    n, c, h, w = r.shape
    if c not in (1, 3):
        raise Exception("Expecting 1 or 3 channels to compute lightness!")
    if len(comparisons_px) != n:
        raise ValueError("one comparison array per image")
    offsets = np.zeros(n + 1, dtype=np.int32)
    rows = []
    for i, comp in enumerate(comparisons_px):
        comp = np.asarray(comp, dtype=np.float64).reshape(-1, 6)
        xy = comp[:, :4].astype(np.int64)
        if comp.shape[0] and (xy.min() < 0 or xy[:, [0, 2]].max() >= w or xy[:, [1, 3]].max() >= h):
            raise IndexError("comparison point outside the %dx%d image %d" % (w, h, i))
        rows.append(comp)
        offsets[i + 1] = offsets[i] + comp.shape[0]
    allc = np.concatenate(rows, axis=0) if rows else np.zeros((0, 6))
    pts = np.ascontiguousarray(allc[:, :5].astype(np.int32))
    wts = np.ascontiguousarray(allc[:, 5].astype(np.float64))
    dev = r.device
    d_pts = torch.from_numpy(pts).to(dev) if pts.size else torch.zeros((1, 5), dtype=torch.int32, device=dev)
    d_wts = torch.from_numpy(wts).to(dev) if wts.size else torch.zeros(1, dtype=torch.float64, device=dev)
    d_off = torch.from_numpy(offsets).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    if n:
        rc = lib.rf_whdr_f32(r.data_ptr(), n, c, h, w, d_pts.data_ptr(), d_wts.data_ptr(),
                             d_off.data_ptr(), float(delta), out.data_ptr(),
                             _ffi.current_stream_ptr(torch))
        _ffi.check(rc, "rf_whdr_f32")
    return out.cpu().numpy()
