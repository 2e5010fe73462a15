"""``cv2.ximgproc``-shaped entry points backed by the HIP kernels.

These two functions have the call signatures of the OpenCV bindings the reference uses
(/root/reference/filter_reflectance.py:60-70), take and return host ``numpy`` images, and run
on the MI355X through librf_hip.so.  A maintainer of the reference can switch to this
framework by replacing ``cv2.ximgproc`` with this module.
"""
import numpy as np

from . import _ffi, ops


def _to_device(img, name, torch):
    arr = np.asarray(img)
    if arr.dtype not in (np.uint8, np.float32):
        raise ValueError("%s: only 8-bit and float32 images are supported (got %s)"
                         % (name, arr.dtype))
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.ndim != 3 or arr.shape[2] not in (1, 3):
        raise ValueError("%s: expected HxW or HxWx{1,3} (got shape %s)" % (name, arr.shape))
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda().unsqueeze(0)


def _to_host(t, like):
    out = t[0].cpu().numpy()
    return out[:, :, 0] if np.ndim(like) == 2 else out


def jointBilateralFilter(joint, src, d, sigmaColor, sigmaSpace, dst=None,
                         borderType=_ffi.BORDER_DEFAULT):
    """cv2.ximgproc.jointBilateralFilter(joint, src, d, sigmaColor, sigmaSpace[, dst[, borderType]])
    for uint8 or float32 images (both of the same depth) with 1 or 3 channels each."""
    # OpenCV's own special case (opencv_contrib joint_bilateral_filter.cpp: `if (joint.empty() ||
    # src.data == joint.data) { bilateralFilter(src, dst, d, sigmaColor, sigmaSpace, borderType); return; }`):
    # one buffer passed as both images (or no joint) is filtered by cv::bilateralFilter.  Its 8-bit
    # path has the same taps, tables and accumulation order; what differs is the last step of a
    # 1-channel image, `cvRound(sum / wsum)` - a true division - where the joint filter (and its own
    # 3-channel path) multiply by `1.f / wsum`.  [recalled, unverified like the rest of the OpenCV
    # arithmetic: DESIGN.md 4.]  The reference never gets here - it reads two files into two buffers
    # (/root/reference/filter_reflectance.py:84-85) - but a caller passing the same array twice does.
    # Detected like OpenCV does, by the data pointer (equal CONTENT in two buffers is the joint filter
    # there as well).  For float32 images that shortcut lands in cv::bilateralFilter's float path - a
    # different operator (its own quantised colour table and scaling), which this package does not
    # implement: refused rather than answered with the joint filter's arithmetic.
    same_buffer = joint is None or joint is src or (
        isinstance(joint, np.ndarray) and isinstance(src, np.ndarray) and joint.shape == src.shape
        and joint.strides == src.strides and joint.ctypes.data == src.ctypes.data)
    if same_buffer and np.asarray(src).dtype == np.float32:
        raise ValueError("jointBilateralFilter: one float32 buffer as both joint and src is "
                         "cv::bilateralFilter's float path in OpenCV, which is not implemented here; "
                         "pass a copy as joint for the joint filter's arithmetic")
    torch = _ffi.require_gpu()
    if joint is None:
        joint = src
    if np.asarray(joint).shape[:2] != np.asarray(src).shape[:2]:
        raise ValueError("joint and src must have the same size")
    if np.asarray(joint).dtype != np.asarray(src).dtype:
        raise ValueError("joint and src must have the same depth")
    # (grey 3-channel images - the CNN's `-r.png` as cv2.imread returns it - need no special case
    #  here: the kernel recognises grey src and grey joint tiles on the device and takes the
    #  one-accumulator / single-channel-joint loops; checking and re-packing on the host cost more
    #  than the 0.1 ms a 1080p upload takes: 6.6 ms against 3.1 ms per call, measured)
    j = _to_device(joint, "joint", torch)
    s = _to_device(src, "src", torch)
    if s.dtype == torch.float32:
        out = ops.joint_bilateral_f32(j, s, d, sigmaColor, sigmaSpace, border=borderType)
    else:
        flags = _ffi.JBF_TRUE_DIVISION if same_buffer and s.shape[3] == 1 else 0
        out = ops.joint_bilateral_u8(j, s, d, sigmaColor, sigmaSpace, border=borderType, flags=flags)
    res = _to_host(out, src)
    if dst is not None:
        np.copyto(dst, res)
        return dst
    return res


CV_8U, CV_32F = 0, 5


def guidedFilter(guide, src, radius, eps, dst=None, dDepth=-1):
    """cv2.ximgproc.guidedFilter(guide, src, radius, eps[, dst[, dDepth]]) for a 3-channel
    guide and a 1- or 3-channel src, each uint8 or float32.  dDepth = -1 (depth of src), CV_8U
    or CV_32F.  uint8 guide + uint8 src + uint8 result is the reference's call
    (/root/reference/filter_reflectance.py:67-70) and runs the 8-bit kernels; every other
    combination runs the float kernels on the values as they are (OpenCV converts without
    scaling) and, for an 8-bit result, rounds like saturate_cast<uchar> - on 8-bit inputs the
    same bytes as the 8-bit kernels, whose float core it is."""
    torch = _ffi.require_gpu()
    g_np, s_np = np.asarray(guide), np.asarray(src)
    if g_np.shape[:2] != s_np.shape[:2]:
        raise ValueError("guide and src must have the same size")
    if dDepth == -1:
        want = s_np.dtype
    elif dDepth in (CV_8U, CV_32F):
        want = np.dtype(np.uint8 if dDepth == CV_8U else np.float32)
    else:
        raise ValueError("guidedFilter: dDepth must be -1, CV_8U (0) or CV_32F (5)")
    if g_np.dtype == np.uint8 and s_np.dtype == np.uint8 and want == np.uint8:
        out = ops.guided_filter_u8(_to_device(guide, "guide", torch), _to_device(src, "src", torch),
                                   radius, eps)
        res = _to_host(out, src)
    else:
        for name, a in (("guide", g_np), ("src", s_np)):
            if a.dtype not in (np.uint8, np.float32):
                raise ValueError("%s: only 8-bit and float32 images are supported (got %s)"
                                 % (name, a.dtype))
        out = ops.guided_filter_f32(_to_device(g_np.astype(np.float32), "guide", torch),
                                    _to_device(s_np.astype(np.float32), "src", torch), radius, eps)
        res = _to_host(out, src)
        if want == np.uint8:   # saturate_cast<uchar>: round half to even, clamp, NaN -> 0
            res = np.clip(np.rint(np.nan_to_num(res, nan=0.0)), 0, 255).astype(np.uint8)
    if dst is not None:
        np.copyto(dst, res)
        return dst
    return res
