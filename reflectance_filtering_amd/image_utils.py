"""Image helpers with the reference's names and behaviour, without OpenCV.

Mirrors /root/reference/image_utils.py (functions ``srgb_to_rgb``, ``rgb_to_srgb``, ``imread``,
``imwrite``, ``colorize``, ``normalize``) so the two command-line tools stay drop-in.  File I/O
goes through Pillow when it is importable and through a small built-in PNG codec otherwise;
array layout is OpenCV's: ``uint8``, H x W x 3, **BGR**.

Behaviour kept on purpose (pinned by tests/golden/*.npz, captured from the reference):
  * ``rgb_to_srgb`` uses ``(1.055*x)**(1/2.4) - 0.055`` above the knee
    (/root/reference/image_utils.py:48), not the textbook ``1.055*x**(1/2.4) - 0.055``.
  * ``imwrite`` of a non-uint8 image truncates ``image*255`` toward zero (``astype(uint8)``)
    after ``normalize`` (/root/reference/image_utils.py:63-68).
  * ``normalize`` divides by the 99.9th percentile taken with the 'lower' rule, only when
    ``max > 1`` (/root/reference/image_utils.py:84-92).
"""
from __future__ import division, print_function

import os
import struct
import zlib

import numpy as np

try:  # Pillow is optional; the PNG fallback below covers the formats the tools produce
    from PIL import Image as _PILImage
    from PIL import ImageOps as _PILImageOps
except Exception:  # pragma: no cover - depends on the environment
    _PILImage = _PILImageOps = None

_SRGB_KNEE = 0.04045
_LINEAR_KNEE = 0.0031308


def srgb_to_rgb(srgb):
    """sRGB -> linear RGB, element-wise (Bell et al. 2014 convention).
    Mirrors /root/reference/image_utils.py:32-39."""
    srgb = np.asarray(srgb)
    linear = np.zeros_like(srgb)
    low = srgb <= _SRGB_KNEE
    high = ~low & (srgb > _SRGB_KNEE)  # NaNs stay 0 exactly like the masked assignment does
    linear[low] = srgb[low] / 12.92
    linear[high] = np.power((srgb[high] + 0.055) / 1.055, 2.4)
    return linear


def rgb_to_srgb(rgb):
    """linear RGB -> sRGB with the reference's formula (see module docstring).
    Mirrors /root/reference/image_utils.py:42-49."""
    rgb = np.asarray(rgb)
    encoded = np.zeros_like(rgb)
    low = rgb <= _LINEAR_KNEE
    high = ~low & (rgb > _LINEAR_KNEE)
    encoded[low] = rgb[low] * 12.92
    encoded[high] = np.power(1.055 * rgb[high], 1.0 / 2.4) - 0.055
    return encoded


def srgb_byte_lut():
    """float32[256]: what the Caffe input blob holds for each sRGB byte value, i.e.
    float32(srgb_to_rgb(v / 255.0)) evaluated in float64 like
    /root/reference/decompose_with_trained_CNN.py:60-68 and cast at the blob assignment (:88)."""
    return srgb_to_rgb(np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)


_steps_cache = None


def srgb_write_steps():
    """float64[255] for the device colourise path: entry k-1 is the smallest x in (0.0031308, 1]
    for which imwrite(..., sRGB=True) stores a byte >= k, i.e.
    ((rgb_to_srgb(x)) * 255).astype(uint8) >= k, evaluated with THIS host's numpy (np.power is
    libm's pow, which a GPU cannot reproduce bit for bit; a table of the <= 255 steps of the
    monotone byte curve can).  +inf where no x <= 1 reaches the byte (rgb_to_srgb(1) = 0.9676)."""
    global _steps_cache
    if _steps_cache is not None:
        return _steps_cache

    def byte_of(x):
        return (rgb_to_srgb(x) * 255).astype(np.uint8).astype(np.int64)

    ks = np.arange(1, 256, dtype=np.int64)
    first = np.nextafter(np.float64(0.0031308), np.float64(1.0))
    lo = np.full(255, first.view(np.int64), dtype=np.int64)       # candidates as bit patterns
    hi = np.full(255, np.float64(1.0).view(np.int64), dtype=np.int64)
    reach = byte_of(np.full(255, 1.0)) >= ks                       # byte(1.0) >= k ?
    # invariant: byte(hi) >= k (where reachable); find the smallest such bit pattern in [lo, hi]
    while np.any(lo < hi):
        mid = lo + (hi - lo) // 2
        ok = byte_of(mid.view(np.float64)) >= ks
        hi = np.where(ok, mid, hi)
        lo = np.where(ok, lo, np.minimum(mid + 1, hi))
    steps = hi.view(np.float64).copy()
    steps[~reach] = np.inf
    _steps_cache = steps
    return steps


_rank_cache = {}


def percentile_rank(count):
    """0-based index, into the sorted values, of np.percentile(x, 99.9, 'lower') for x.size ==
    count - obtained from numpy itself so that the installed version's index arithmetic is the
    one that counts (/root/reference/image_utils.py:90)."""
    count = int(count)
    if count not in _rank_cache:
        _rank_cache[count] = int(np.percentile(np.arange(count, dtype=np.int64), 99.9,
                                               method="lower"))
    return _rank_cache[count]


def normalize(img):
    """Scale to 0..1: identity if max <= 1, else divide by the 99.9th percentile ('lower') and
    clip.  Mirrors /root/reference/image_utils.py:84-92 (works on a copy)."""
    img = img.copy()
    if np.max(img) > 1:
        img /= np.percentile(img, 99.9, method="lower")
        img = np.clip(img, 0, 1)
    return img


def colorize(intensity, image, eps=1e-3):
    """Colour reflectance and grey shading from a reflectance-intensity map and the input
    image (used as it comes from imread).  Mirrors /root/reference/image_utils.py:76-81."""
    mean_rgb = np.mean(image, axis=2)
    shading = mean_rgb / intensity
    reflectance = image / np.maximum(shading, eps)[:, :, np.newaxis]
    return reflectance, shading


# --------------------------------------------------------------------------------------
# file I/O (cv2.imread / cv2.imwrite semantics)
# --------------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _png_decode(data):
    """Minimal PNG reader: 8/16-bit gray, gray+alpha, RGB, RGBA, 8-bit palette; no interlace."""
    if data[:8] != _PNG_SIG:
        return None
    pos = 8
    idat = []
    palette = None
    width = height = depth = ctype = interlace = None
    while pos + 8 <= len(data):
        (length,) = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if tag == b"IHDR":
            width, height, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    if width is None or interlace or depth not in (8, 16):
        return None
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if chans is None or (ctype == 3 and depth != 8):
        return None
    bpp = chans * depth // 8
    stride = width * bpp
    raw = zlib.decompress(b"".join(idat))
    out = np.zeros((height, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(height):
        ftype = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ftype == 1:
                    pred = a
                elif ftype == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    if depth == 16:  # keep the high byte, like libpng's strip_16 under IMREAD_COLOR
        out = out.reshape(height, width * chans, 2)[:, :, 0]
    px = out.reshape(height, width, chans)
    if ctype == 3:
        if palette is None:
            return None
        px = palette[px[:, :, 0]]
    elif chans == 1:
        px = np.repeat(px, 3, axis=2)
    elif chans == 2:
        px = np.repeat(px[:, :, :1], 3, axis=2)
    elif chans == 4:
        px = px[:, :, :3]
    return np.ascontiguousarray(px[:, :, ::-1])  # RGB -> BGR


def _png_encode(image, level=1):
    """8-bit gray or BGR image -> PNG bytes.  Every row uses the "Up" filter (difference to the
    row above, computed for the whole image at once with numpy) and zlib runs at its fastest
    level, which is also OpenCV's default for cv2.imwrite (IMWRITE_PNG_COMPRESSION = 1): encoding
    a 1080p BGR image takes a quarter of the time of unfiltered rows at level 6.  PNG is
    lossless, so the pixels a reader gets back do not depend on these choices."""
    if image.ndim == 2:
        ctype, rows = 0, image
    else:
        ctype, rows = 2, image[:, :, ::-1].reshape(image.shape[0], -1)
    height, width = image.shape[:2]
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    raw = np.empty((height, 1 + rows.shape[1]), np.uint8)
    raw[:, 0] = 2                                    # filter type 2: Up
    raw[0, 1:] = rows[0]                             # the row above the first one counts as zeros
    np.subtract(rows[1:], rows[:-1], out=raw[1:, 1:])   # uint8 arithmetic wraps modulo 256

    # level 1 + Z_RLE: OpenCV's defaults for cv2.imwrite (IMWRITE_PNG_COMPRESSION 1, strategy RLE)
    packer = zlib.compressobj(level, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    deflated = packer.compress(raw.tobytes()) + packer.flush()

    def chunk(tag, body):
        return (struct.pack(">I", len(body)) + tag + body
                + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF))

    return (_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, ctype, 0, 0, 0))
            + chunk(b"IDAT", deflated) + chunk(b"IEND", b""))


def _read_any(filename):
    """File -> uint8 HxWx3 BGR, or None if it cannot be decoded (cv2.imread contract)."""
    try:
        if _PILImage is not None:
            with _PILImage.open(filename) as im:
                if im.mode in ("I;16", "I;16B", "I;16L", "I"):
                    # cv2.imread(IMREAD_COLOR) always reduces 16-bit samples to their high byte:
                    # decided by the file's bit depth (Pillow's mode), never by the value range
                    arr = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
                    return np.ascontiguousarray(np.repeat(arr[:, :, None], 3, axis=2))
                if _PILImageOps is not None and getattr(im, "format", None) in ("JPEG", "MPO", "TIFF"):
                    im = _PILImageOps.exif_transpose(im)   # cv2.imread applies the EXIF orientation
                rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
            return np.ascontiguousarray(rgb[:, :, ::-1])
        with open(filename, "rb") as fh:
            return _png_decode(fh.read())
    except Exception:
        return None


def imread(filename):
    """Read an image the way ``cv2.imread(filename)`` does (uint8, HxWx3, BGR) and fail loudly.
    Mirrors /root/reference/image_utils.py:52-57."""
    img = _read_any(filename)
    if img is None:
        raise Exception("Input image not readable: {}".format(filename))
    return img


def _write_u8(filename, image):
    """uint8 HxW or HxWx3(BGR) -> file; returns success like cv2.imwrite."""
    if image.ndim == 3 and image.shape[2] == 1:
        image = image[:, :, 0]
    if image.dtype != np.uint8 or image.ndim not in (2, 3) or (image.ndim == 3
                                                               and image.shape[2] != 3):
        return False
    try:
        ext = os.path.splitext(filename)[1].lower()
        if ext == ".png" or _PILImage is None:
            if ext != ".png":
                return False
            with open(filename, "wb") as fh:
                fh.write(_png_encode(image))
            return True
        pil = _PILImage.fromarray(image if image.ndim == 2
                                  else np.ascontiguousarray(image[:, :, ::-1]))
        pil.save(filename)
        return True
    except (OSError, ValueError, KeyError):
        return False


def imwrite(filename, image, sRGB=False):
    """Write an image; non-uint8 input is normalised to 0..1, optionally gamma-encoded, scaled
    by 255 and truncated.  Mirrors /root/reference/image_utils.py:60-73."""
    if image.dtype != np.uint8:
        image = normalize(image)
        if sRGB:
            image = rgb_to_srgb(image)
        image = (image * 255).astype(np.uint8)
    if not _write_u8(filename, image):
        raise Exception("Not able to write {}, does the folder exist?".format(filename))
