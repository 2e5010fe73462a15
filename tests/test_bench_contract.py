"""CPU suite, part 5: the pieces of bench.py that do not need a GPU (JSON sub-objects)."""
import numpy as np

import bench
from tests import synth


def test_cpu_baseline_object():
    joint = synth.scene_u8(120, 96, seed=1)
    src = synth.reflectance_like_u8(120, 96, seed=2)
    obj = bench.cpu_baseline(joint, src, 20.0, 4.0, target_s=0.2)
    assert set(obj) == {"value", "unit", "cores", "kind", "sample"}
    assert obj["unit"] == "MP/s" and obj["kind"] == "port" and obj["value"] > 0
    assert 1 <= obj["cores"] == bench.usable_cores()


def test_valu_roofline_object():
    taps = 3409.0 * 256 * 1080 * 1920
    obj = bench.valu_roofline(256, 1080, 1920, 33, 265.0, taps)
    assert 0.3 < obj["frac"] < 1.0 and obj["bound"] == "valu-issue"
    # 67 tap rows, groups of 4 columns covering the disk: between the 3,409 taps/4 and the
    # 4,489-tap square/4, plus the zero-weight padding columns
    per_wave = obj["column_steps_per_launch"] / (256 * 30 * 17 * 16)
    assert 3409 / 4 < per_wave < 67 * (67 + 11) / 1 and per_wave % 4 == 0
    assert np.isclose(obj["taps_per_s"], taps / 0.265)
