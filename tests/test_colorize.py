"""CPU suite: the colourised-output path of decompose_image.

The numpy oracle is pinned on bytes captured from the reference's own colorize/imwrite
(tests/golden/colorize_write.npz, decompose_outputs.npz); the two host-side tables the device
path relies on (sRGB step positions, percentile rank) are checked against numpy itself."""
import os

import numpy as np
import pytest

from oracle import colorize_numpy as oc
from reflectance_filtering_amd import image_utils as iu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = ("natural", "dark", "holes", "tiny")


def test_oracle_reproduces_reference_bytes():
    d = np.load(os.path.join(G, "colorize_write.npz"))
    for tag in TAGS:
        refl, shad = oc.colorize_srgb_u8(d[tag + "_image"], d[tag + "_r"])
        assert np.array_equal(refl, d[tag + "_refl_png"]), tag
        assert np.array_equal(shad, d[tag + "_shading_png"]), tag
    g = np.load(os.path.join(G, "decompose_outputs.npz"))
    refl, shad = oc.colorize_srgb_u8(g["scene"], g["r"])
    assert np.array_equal(refl, g["r_colorized_png"])
    assert np.array_equal(shad, g["s_colorized_png"])


def _byte_of(x):
    return (iu.rgb_to_srgb(np.asarray(x, dtype=np.float64)) * 255).astype(np.uint8).astype(int)


def test_srgb_write_steps_are_the_steps_of_the_byte_curve():
    steps = iu.srgb_write_steps()
    assert steps.shape == (255,) and steps.dtype == np.float64
    fin = np.isfinite(steps)
    k = np.arange(1, 256)
    assert np.all(np.diff(steps[fin]) >= 0) and not fin[246:].any() and fin[:246].all()
    first = np.nextafter(0.0031308, 1.0)
    assert np.all(_byte_of(steps[fin]) >= k[fin])
    below = np.nextafter(steps[fin], 0.0)
    inside = below >= first
    assert np.all(_byte_of(below[inside]) < k[fin][inside])
    assert np.all(steps[fin][~inside] == first)
    # counting steps <= x is the byte, on random values and right around every step
    rng = np.random.default_rng(3)
    xs = [rng.uniform(first, 1.0, 200000), 10.0 ** rng.uniform(-2.5, 0.0, 100000), np.array([first, 1.0])]
    for d in range(-3, 4):
        x = steps[fin].copy()
        for _ in range(abs(d)):
            x = np.nextafter(x, 2.0 if d > 0 else 0.0)
        xs.append(x[(x >= first) & (x <= 1.0)])
    x = np.concatenate(xs)
    x = x[(x >= first) & (x <= 1.0)]
    assert np.array_equal(np.searchsorted(steps, x, side="right"), _byte_of(x))


@pytest.mark.parametrize("count", [1, 2, 3, 999, 1000, 1001, 1002, 2001, 12345, 166500, 499500])
def test_percentile_rank_is_numpys_lower_percentile(count):
    rng = np.random.default_rng(count)
    x = rng.random(count) * 300
    k = iu.percentile_rank(count)
    assert np.sort(x)[k] == np.percentile(x, 99.9, method="lower")
