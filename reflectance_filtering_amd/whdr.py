"""WHDR (weighted human disagreement rate) of reflectance predictions against IIW judgements.

Mirrors the evaluation arithmetic of the reference's training layer
(/root/reference/training/layers/whdr_layer.py:180-196 `_lightness`,
:239-250 `_extract_valid_comparisons_with_actual_size`, :253-287 `whdr`) and the JSON reading of
/root/reference/training/createNumpyArrayWithComparisonsForIIW.py:320-347,613-645.

    comparisons[k] = (x1, y1, x2, y2, darker, weight)   x, y normalised to [0,1);
                     darker: 0 = 'E' (about equal), 1 = point 1 darker, 2 = point 2 darker

`whdr` is the per-image host form (same numpy operations as the reference, so it follows the
installed numpy's scalar promotion rules exactly as the reference would under it);
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
    """Normalised -> pixel coordinates by truncation, in the array's own dtype
    (whdr_layer.py:239-250)."""
    res = np.array(comparisons, copy=True)
    res[:, [0, 2]] = (res[:, [0, 2]] * width).astype(int)
    res[:, [1, 3]] = (res[:, [1, 3]] * height).astype(int)
    return res


def _lightness(r):
    if len(r) == 3:
        return max(EPS, np.mean(r))
    if len(r) == 1:
        return max(EPS, r)
    raise Exception("Expecting 1 or 3 channels to compute lightness!")


def whdr(reflectance, comparisons, delta=0.1):
    """WHDR of one [c,h,w] reflectance image; `comparisons` in pixel coordinates
    (whdr_layer.py:253-287).  No comparisons -> 0.0."""
    error_sum = 0.0
    weight_sum = 0.0
    for c in range(comparisons.shape[0]):
        x1, y1, x2, y2, darker = comparisons[c, :5].astype(int)
        weight = comparisons[c, 5]
        l1 = _lightness(reflectance[:, y1, x1])
        l2 = _lightness(reflectance[:, y2, x2])
        if l2 / l1 > 1 + delta:
            alg_darker = 1
        elif l1 / l2 > 1 + delta:
            alg_darker = 2
        else:
            alg_darker = 0
        if darker != alg_darker:
            error_sum += weight
        weight_sum += weight
    return error_sum / weight_sum if weight_sum else 0.0


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
