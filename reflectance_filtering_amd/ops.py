"""Batched device operators: thin, torch-tensor-in / torch-tensor-out wrappers over the C ABI.

All tensors are CUDA(HIP) uint8, contiguous, shaped [N,H,W,C] (C in {1,3}) like a stack of
``cv2.imread`` results.  Work is enqueued on torch's current stream; nothing synchronises.
"""
import numpy as np

from . import _ffi

_gf_workspaces = {}
_GF_CACHE_PER_DEVICE = 4
_GF_CACHE_BYTES_PER_DEVICE = 64 << 30     # ... and at most this much scratch kept per device
_cnn_consts = {}


def release_workspaces():
    """Drop the cached guided-filter scratch buffers and CNN constants (device memory)."""
    _gf_workspaces.clear()
    _cnn_consts.clear()
    _steps_dev.clear()


def _chk_images(t, name, torch):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8
            and t.dim() == 4 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous CUDA uint8 tensor [N,H,W,C]" % name)


def joint_bilateral_u8(joint, src, d, sigma_color, sigma_space, border=_ffi.BORDER_DEFAULT,
                       flags=0, out=None, grey_as_bgr=False):
    """Batched cv2.ximgproc.jointBilateralFilter(joint, src, d, sigmaColor, sigmaSpace).
    grey_as_bgr: a 1-channel joint is filtered as the 3-equal-channel image cv2.imread would
    have produced from it (no replicated copy is made)."""
    if grey_as_bgr:
        flags |= _ffi.JBF_GREY_AS_BGR
    torch = _ffi.require_gpu()
    lib = _ffi.load_library()
    _chk_images(joint, "joint", torch)
    _chk_images(src, "src", torch)
    if joint.shape[:3] != src.shape[:3]:
        raise ValueError("joint and src must have the same N,H,W")
    if out is None:
        out = torch.empty_like(src)
    n, h, w, scn = src.shape
    rc = lib.rf_jbf_u8(joint.data_ptr(), src.data_ptr(), out.data_ptr(), n, h, w,
                       joint.shape[3], scn, int(d), float(sigma_color), float(sigma_space),
                       int(border), int(flags), _ffi.current_stream_ptr(torch))
    _ffi.check(rc, "rf_jbf_u8")
    return out


def gf_workspace(n, h, w, scn, radius, device, torch):
    """Guided-filter scratch for the CURRENT stream of `device`, cached per (device, stream):
    two streams (or threads with their own streams) never share planes.  The cache keeps one
    buffer per key, sized by rf_gf_workspace_bytes (capped at 1/8 of the device's memory, at most
    32 GiB), at most four buffers and 64 GiB per device; release_workspaces() drops them."""
    lib = _ffi.load_library()
    need = lib.rf_gf_workspace_bytes(n, h, w, 3, scn, radius)
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _gf_workspaces.pop(key, None)       # re-inserted below: dict order = least recently used first
    if ws is None or ws.numel() < need:
        ws = None
        # streams come and go: keep at most _GF_CACHE_PER_DEVICE buffers per device (dropping a
        # buffer is safe: torch's stream-ordered allocator does not recycle the block before the
        # stream it was allocated on has passed the work that used it)
        mine = [k for k in _gf_workspaces if k[0] == dev]
        for k in mine[:max(0, len(mine) - (_GF_CACHE_PER_DEVICE - 1))]:
            del _gf_workspaces[k]
        mine = [k for k in _gf_workspaces if k[0] == dev]       # least recently used first
        held = sum(_gf_workspaces[k].numel() for k in mine)
        while mine and held + need > _GF_CACHE_BYTES_PER_DEVICE:
            held -= _gf_workspaces.pop(mine.pop(0)).numel()
        ws = torch.empty(need, dtype=torch.uint8, device=device)
    _gf_workspaces[key] = ws
    return ws


def guided_filter_u8(guide, src, radius, eps, iterations=1, out=None, workspace=None):
    """Batched cv2.ximgproc.guidedFilter(guide, src, radius, eps), applied `iterations` times
    with the uint8 result fed back as src (the reference's chained CLI runs)."""
    torch = _ffi.require_gpu()
    lib = _ffi.load_library()
    _chk_images(guide, "guide", torch)
    _chk_images(src, "src", torch)
    if guide.shape[:3] != src.shape[:3]:
        raise ValueError("guide and src must have the same N,H,W")
    if out is None:
        out = torch.empty_like(src)
    n, h, w, scn = src.shape
    if workspace is None:
        workspace = gf_workspace(n, h, w, scn, int(radius), src.device, torch)
    rc = lib.rf_gf_u8(guide.data_ptr(), src.data_ptr(), out.data_ptr(), n, h, w, guide.shape[3],
                      scn, int(radius), float(eps), int(iterations), workspace.data_ptr(),
                      workspace.numel(), _ffi.current_stream_ptr(torch))
    _ffi.check(rc, "rf_gf_u8")
    return out


def _cnn_device_consts(torch, device, weights):
    """(packed weights, sRGB table) on `device`.  Packing is the net-load step
    (rf_cnn_pack_weights): done once for the shipped weights (cached per device), once per call
    for caller-supplied ones; the forward pass itself keeps no state in the library."""
    from . import image_utils as iu
    from . import weights as wmod
    lib = _ffi.load_library()

    def pack(w_host):
        raw = torch.from_numpy(w_host).to(device)
        packed = torch.empty(_ffi.CNN_NPACKED, dtype=torch.float32, device=device)
        _ffi.check(lib.rf_cnn_pack_weights(raw.data_ptr(), packed.data_ptr(),
                                           _ffi.current_stream_ptr(torch)), "rf_cnn_pack_weights")
        return packed                       # `raw` may go: the pack is ordered on this stream

    if weights is None:
        key = (str(device), "default")
        if key not in _cnn_consts:
            lut = torch.from_numpy(iu.srgb_byte_lut()).to(device)
            _cnn_consts[key] = (pack(wmod.load_weights()), lut)
            # other streams may use the cached copy: make it visible to all of them once
            torch.cuda.current_stream(device).synchronize()
        return _cnn_consts[key]
    w = np.ascontiguousarray(weights, dtype=np.float32).ravel()
    if w.size != _ffi.CNN_NPARAMS:
        raise ValueError("weights must hold %d floats" % _ffi.CNN_NPARAMS)
    return (pack(w), torch.from_numpy(iu.srgb_byte_lut()).to(device))


def cnn_reflectance_u8(bgr, weights=None, want_float=True, want_u8=True):
    """Batched reflectance prediction: uint8 BGR [N,H,W,3] -> (r float32 [N,H,W], r_u8 [N,H,W]).
    r_u8 = trunc(r*255) is the content of the reference's `<base>-r.png`."""
    torch = _ffi.require_gpu()
    lib = _ffi.load_library()
    _chk_images(bgr, "bgr", torch)
    if bgr.shape[3] != 3:
        raise ValueError("bgr must have 3 channels")
    n, h, w, _ = bgr.shape
    packed, lut = _cnn_device_consts(torch, bgr.device, weights)
    r = torch.empty((n, h, w), dtype=torch.float32, device=bgr.device) if want_float else None
    r8 = torch.empty((n, h, w), dtype=torch.uint8, device=bgr.device) if want_u8 else None
    rc = lib.rf_cnn_reflectance_packed_u8(bgr.data_ptr(), r.data_ptr() if want_float else None,
                                          r8.data_ptr() if want_u8 else None, n, h, w,
                                          packed.data_ptr(), lut.data_ptr(),
                                          _ffi.current_stream_ptr(torch))
    _ffi.check(rc, "rf_cnn_reflectance_packed_u8")
    return r, r8


_steps_dev = {}


def colorize_srgb_u8(images, r, want_reflectance=True, want_shading=True):
    """Device form of iu.colorize(r, image) followed by iu.imwrite(..., sRGB=True) of both
    results (/root/reference/decompose_with_trained_CNN.py:121-128): returns the uint8 bytes of
    `<base>-r_colorized.png` [N,H,W,3] (BGR) and `<base>-s_colorized.png` [N,H,W].
    images: CUDA uint8 [N,H,W,3]; r: CUDA float32 [N,H,W] (the CNN's reflectance intensity)."""
    from . import image_utils as iu
    torch = _ffi.require_gpu()
    lib = _ffi.load_library()
    _chk_images(images, "images", torch)
    if images.shape[3] != 3:
        raise ValueError("images must have 3 channels")
    if (not torch.is_tensor(r) or not r.is_cuda or r.dtype != torch.float32
            or tuple(r.shape) != tuple(images.shape[:3]) or not r.is_contiguous()):
        raise ValueError("r must be a contiguous CUDA float32 tensor [N,H,W] matching images")
    n, h, w, _ = images.shape
    dev = images.device
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    steps = _steps_dev.get(key)
    if steps is None:
        steps = torch.from_numpy(iu.srgb_write_steps()).to(dev)
        _steps_dev[key] = steps
    refl = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev) if want_reflectance else None
    shad = torch.empty((n, h, w), dtype=torch.uint8, device=dev) if want_shading else None
    if n == 0 or (refl is None and shad is None):
        return refl, shad
    ws = torch.empty(lib.rf_colorize_workspace_bytes(n), dtype=torch.uint8, device=dev)
    rc = lib.rf_colorize_srgb_u8(images.data_ptr(), r.data_ptr(),
                                 refl.data_ptr() if refl is not None else None,
                                 shad.data_ptr() if shad is not None else None, n, h, w,
                                 iu.percentile_rank(3 * h * w), iu.percentile_rank(h * w),
                                 steps.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _ffi.current_stream_ptr(torch))
    _ffi.check(rc, "rf_colorize_srgb_u8")
    return refl, shad


def _chk_images_f32(t, name, torch):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32
            and t.dim() == 4 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous CUDA float32 tensor [N,H,W,C]" % name)


def joint_bilateral_f32(joint, src, d, sigma_color, sigma_space, border=_ffi.BORDER_DEFAULT,
                        out=None):
    """Batched cv2.ximgproc.jointBilateralFilter on float32 images (CV_32F path: interpolated
    colour table over each joint image's value range).  Synchronises the current stream."""
    torch = _ffi.require_gpu()
    lib = _ffi.load_library()
    _chk_images_f32(joint, "joint", torch)
    _chk_images_f32(src, "src", torch)
    if joint.shape[:3] != src.shape[:3]:
        raise ValueError("joint and src must have the same N,H,W")
    if out is None:
        out = torch.empty_like(src)
    n, h, w, scn = src.shape
    ws = torch.empty(max(1, lib.rf_jbf_f32_workspace_bytes(n, joint.shape[3])), dtype=torch.uint8,
                     device=src.device)
    rc = lib.rf_jbf_f32(joint.data_ptr(), src.data_ptr(), out.data_ptr(), n, h, w, joint.shape[3],
                        scn, int(d), float(sigma_color), float(sigma_space), int(border),
                        ws.data_ptr(), ws.numel(), _ffi.current_stream_ptr(torch))
    _ffi.check(rc, "rf_jbf_f32")
    return out


def guided_filter_f32(guide, src, radius, eps, iterations=1, out=None):
    """Batched cv2.ximgproc.guidedFilter on float32 guide/src: float32 result, no rounding."""
    torch = _ffi.require_gpu()
    lib = _ffi.load_library()
    _chk_images_f32(guide, "guide", torch)
    _chk_images_f32(src, "src", torch)
    if guide.shape[:3] != src.shape[:3]:
        raise ValueError("guide and src must have the same N,H,W")
    if out is None:
        out = torch.empty_like(src)
    n, h, w, scn = src.shape
    ws = torch.empty(max(1, lib.rf_gf_f32_workspace_bytes(n, h, w, 3, scn, int(radius))),
                     dtype=torch.uint8, device=src.device)
    rc = lib.rf_gf_f32(guide.data_ptr(), src.data_ptr(), out.data_ptr(), n, h, w, guide.shape[3],
                       scn, int(radius), float(eps), int(iterations), ws.data_ptr(), ws.numel(),
                       _ffi.current_stream_ptr(torch))
    _ffi.check(rc, "rf_gf_f32")
    return out


class CapturedCall(object):
    """A fixed sequence of operator calls captured once into a HIP graph and replayed.

    The launch-bound cases (single small images: the 3x guided-filter chain is 16 launches, the
    colourise path 18) pay one graph launch instead.  `fn` must only call operators of this
    module on pre-allocated tensors (out= arguments, an explicit guided-filter workspace): it is
    run once eagerly first, so that the parameter tables a call uploads on first use exist
    before the capture, then captured on a side stream.  rf_jbf_f32 cannot be captured (it
    synchronises).  replay() re-runs the sequence on the same buffers."""

    def __init__(self, fn):
        torch = _ffi.require_gpu()
        self._torch = torch
        fn()                                   # warm-up: table uploads, kernel attribute calls
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = fn()

    def replay(self):
        self.graph.replay()
        return self.result
