"""Seeded synthetic inputs shared by tests, smoke() and bench.py (numpy on the host,
torch on the device).  Natural-image-like on purpose: white noise would zero almost every
colour weight of the bilateral filter and make every variance of the guided filter huge."""
import numpy as np


def _smooth_field(h, w, rng, octaves=5):
    """Sum of bilinearly upsampled noise octaves (1/f-like), zero mean, unit-ish variance."""
    acc = np.zeros((h, w))
    amp, total = 1.0, 0.0
    for o in range(octaves):
        gh, gw = 2 + (h >> (octaves - o)), 2 + (w >> (octaves - o))
        g = rng.standard_normal((gh, gw))
        ys = np.linspace(0, gh - 1.001, h)
        xs = np.linspace(0, gw - 1.001, w)
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
        b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
        acc += amp * (a * (1 - fy) + b * fy)
        total += amp * amp
        amp *= 0.55
    return acc / np.sqrt(total)


def scene_u8(h, w, seed):
    """Correlated-channel RGB 'photo' (uint8 HxWx3, BGR), mean 128 / std ~48, mild noise."""
    rng = np.random.default_rng(seed)
    base = _smooth_field(h, w, rng)
    img = np.empty((h, w, 3))
    for c in range(3):
        img[:, :, c] = 0.9 * base + 0.44 * _smooth_field(h, w, rng)
    img = 128 + 48 * img + rng.normal(0, 1.5, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def reflectance_like_u8(h, w, seed):
    """Grey 'CNN output' replicated to 3 channels: trunc(r*255), r smooth in [0.15, 1)."""
    rng = np.random.default_rng(seed)
    f = _smooth_field(h, w, rng, octaves=4)
    r = 0.15 + 0.85 / (1 + np.exp(-1.5 * f))
    g = np.floor(np.clip(r, 0, 0.9999) * 255).astype(np.uint8)
    return np.repeat(g[:, :, None], 3, axis=2)


def flat_guide_u8(h, w, seed, cells=40):
    """Piecewise-constant 'L1-flattened' guidance: Voronoi cells of flat colour, +-1 dither."""
    rng = np.random.default_rng(seed)
    pts = rng.random((cells, 2)) * [h, w]
    cols = rng.integers(20, 236, (cells, 3))
    yy, xx = np.mgrid[0:h, 0:w]
    d = (yy[:, :, None] - pts[:, 0]) ** 2 + (xx[:, :, None] - pts[:, 1]) ** 2
    lab = np.argmin(d, axis=2)
    img = cols[lab] + rng.integers(-1, 2, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)
