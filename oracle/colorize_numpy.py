"""TEST INFRASTRUCTURE ONLY - CPU restatement of the colourised outputs of decompose_image.

Restates, with numpy float64 exactly as the reference evaluates it,
  /root/reference/image_utils.py:76-81   colorize(intensity, image, eps=1e-3)
  /root/reference/image_utils.py:84-92   normalize (99.9th percentile 'lower', clip)
  /root/reference/image_utils.py:42-49   rgb_to_srgb ((1.055*x)**(1/2.4) - 0.055 above 0.0031308)
  /root/reference/image_utils.py:60-68   imwrite's (image * 255).astype(uint8)
as called by /root/reference/decompose_with_trained_CNN.py:121-128.  Pinned: this file is
checked against bytes captured from the reference's own functions
(tests/golden/colorize_write.npz, decompose_outputs.npz; generator tests/golden/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import numpy as np


def _write_bytes(x):
    """What imwrite(..., sRGB=True) hands to the PNG encoder for a float64 array."""
    x = np.array(x, dtype=np.float64, copy=True)
    if x.max() > 1:
        flat = np.sort(x, axis=None)
        # 'lower' percentile = an order statistic; the index comes from numpy's own rule
        k = int(np.percentile(np.arange(flat.size), 99.9, method="lower"))
        x = x / flat[k]
        x = np.minimum(np.maximum(x, 0.0), 1.0)
    out = np.zeros_like(x)
    low = x <= 0.0031308
    high = x > 0.0031308
    out[low] = x[low] * 12.92
    out[high] = np.power(1.055 * x[high], 1.0 / 2.4) - 0.055
    return (out * 255).astype(np.uint8)


def colorize_srgb_u8(image_bgr_u8, r_f32):
    """(reflectance bytes [H,W,3], shading bytes [H,W]) for one image."""
    img = np.asarray(image_bgr_u8)
    r = np.asarray(r_f32, dtype=np.float32)
    mean = img.astype(np.float64).sum(axis=2) / 3.0
    shading = mean / r.astype(np.float64)
    refl = img.astype(np.float64) / np.maximum(shading, 1e-3)[:, :, None]
    return _write_bytes(refl), _write_bytes(shading)
