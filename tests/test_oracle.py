"""CPU suite, part 1: the oracle itself.

The C restatement (T1, oracle/rf_oracle.c) is checked against (a) the golden vectors captured
from the reference's own Python helpers, (b) the float64 definitions (T0) and (c) known-answer
cases.  Nothing here touches the GPU or the product library.
"""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import t0_numpy as t0
from tests import synth

G = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------ shared primitives
def test_border_interpolate_matches_numpy_pad():
    for length in (1, 2, 3, 7, 40):
        base = np.arange(length)
        for pad in (1, 5, 3 * length + 2):
            for border, mode in ((co.BORDER_REFLECT_101, "reflect"), (co.BORDER_REFLECT, "symmetric"),
                                 (co.BORDER_REPLICATE, "edge"), (co.BORDER_WRAP, "wrap")):
                want = np.pad(base, pad, mode=mode)
                got = [co.border_interpolate(p, length, border) for p in range(-pad, length + pad)]
                assert list(want) == got, (length, pad, mode)
    assert co.border_interpolate(-1, 5, co.BORDER_CONSTANT) == -1


def test_jbf_radius_and_tap_table():
    # d=-1, sigma_space=22 -> cvRound(33.0)=33, 3409 taps; 28 -> 42, 5525 taps
    assert co.jbf_radius(-1, 22.0) == 33
    assert co.jbf_radius(-1, 28.0) == 42
    assert co.jbf_radius(-1, 1.0) == 2      # cvRound(1.5) = 2 (half to even)
    assert co.jbf_radius(-1, 0.3) == 1      # max(radius, 1)
    assert co.jbf_radius(9, 22.0) == 4
    di, dj, sw = co.jbf_taps(33, 22.0)
    assert len(di) == 3409
    assert (di[0], dj[0]) == (-33, 0) and (di[-1], dj[-1]) == (33, 0)
    order = di.astype(np.int64) * 1000 + dj
    assert np.all(np.diff(order) > 0), "taps must be row-major"
    k0 = np.flatnonzero((di == 0) & (dj == 0))[0]
    assert sw[k0] == 1.0
    assert len(co.jbf_taps(42, 28.0)[0]) == 5525


def test_jbf_color_lut_underflows_to_zero():
    lut = co.jbf_color_lut(20.0, 3)
    assert lut.shape == (768,) and lut[0] == 1.0
    first_zero = int(np.flatnonzero(lut == 0)[0])
    assert 280 <= first_zero <= 300  # exp(-a^2/800) leaves float32 near a = 288
    assert np.all(np.diff(lut) <= 0)


# ------------------------------------------------------------------ joint bilateral
@pytest.mark.parametrize("jcn,scn", [(3, 3), (1, 3), (3, 1), (1, 1)])
def test_jbf_t1_matches_float64_definition(jcn, scn):
    joint = synth.scene_u8(40, 56, seed=11)
    src = synth.scene_u8(40, 56, seed=12)
    joint = joint if jcn == 3 else joint[:, :, 0]
    src = src if scn == 3 else src[:, :, 1]
    got = co.joint_bilateral_filter(joint, src, -1, 20, 22)
    want, ntaps = t0.joint_bilateral_f64(joint, src, 20, 22)
    assert ntaps == 3409
    assert got.shape == src.shape and got.dtype == np.uint8
    # float32 sequential accumulation vs float64: < 1e-3 before rounding, so the bytes agree
    # except where the exact value sits within 1e-3 of a .5 boundary
    diff = np.abs(got.astype(np.float64) - want)
    assert diff.max() <= 0.5 + 1e-3
    assert np.mean(got != np.clip(np.rint(want), 0, 255)) < 5e-3


def test_jbf_known_answers():
    rng = np.random.default_rng(3)
    const = np.full((20, 30, 3), 77, np.uint8)
    src = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    # constant src -> identity whatever the joint is
    assert np.array_equal(co.joint_bilateral_filter(src, const, -1, 20, 5), const)
    # constant joint -> colour weight 1 everywhere -> disk-masked Gaussian blur of src
    got = co.joint_bilateral_filter(const, src, -1, 20, 3)
    want, _ = t0.joint_bilateral_f64(const, src, 1e9, 3)
    assert np.abs(got - want).max() <= 0.5 + 1e-3
    # tiny sigma_color with a noisy joint -> only the centre tap survives -> identity
    joint = (np.arange(20 * 30 * 3).reshape(20, 30, 3) * 37 % 251).astype(np.uint8)
    assert np.array_equal(co.joint_bilateral_filter(joint, src, -1, 0.05, 2), src)


def test_jbf_image_smaller_than_radius_and_flags():
    joint = synth.scene_u8(9, 7, seed=5)
    src = synth.scene_u8(9, 7, seed=6)
    got = co.joint_bilateral_filter(joint, src, -1, 25, 22)  # radius 33 > image: multi-bounce
    want, _ = t0.joint_bilateral_f64(joint, src, 25, 22)
    assert np.abs(got - want).max() <= 0.5 + 1e-3
    a = co.joint_bilateral_filter(joint, src, -1, 25, 4)
    b = co.joint_bilateral_filter(joint, src, -1, 25, 4, flags=co.FLAG_TRUE_DIVISION)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    with pytest.raises(ValueError):
        co.joint_bilateral_filter(joint[:, :, :2], src, -1, 25, 4)


def test_jbf_c_oracle_matches_second_restatement_bitwise():
    """rf_oracle.c against the numpy spelling of the same operation order (oracle/t1_numpy.py)."""
    from oracle import t1_numpy as t1
    for (h, w, jcn, scn, sc, ss) in ((19, 23, 3, 3, 20, 4), (12, 30, 1, 3, 35, 3), (9, 7, 3, 1, 20, 22)):
        joint = synth.scene_u8(h, w, seed=h)
        src = synth.scene_u8(h, w, seed=w)
        joint = joint if jcn == 3 else joint[:, :, 0]
        src = src if scn == 3 else src[:, :, 2]
        assert np.array_equal(co.joint_bilateral_filter(joint, src, -1, sc, ss),
                              t1.joint_bilateral_f32seq(joint, src, sc, ss))
    assert np.array_equal(
        co.joint_bilateral_filter(joint, src, 7, 20, 4, flags=co.FLAG_TRUE_DIVISION),
        t1.joint_bilateral_f32seq(joint, src, 20, 4, d=7, true_division=True))


def test_gf_c_oracle_matches_second_restatement_bitwise():
    from oracle import t1_numpy as t1
    rng = np.random.default_rng(8)
    plane = (rng.standard_normal((23, 31)) * 40).astype(np.float32)
    for r in (1, 5, 40):
        assert np.array_equal(co.box_mean_f32(plane, r), t1.box_mean_seq(plane, r))
    guide = synth.flat_guide_u8(40, 52, seed=3, cells=9)
    src = synth.reflectance_like_u8(40, 52, seed=4)
    for r, eps in ((6, 3.0), (45, 7.0), (2, 1e-3)):
        want_u8, want_f = t1.guided_filter_f32seq(guide, src, r, eps)
        got_u8, got_f = co.guided_filter(guide, src, r, eps, return_float=True)
        assert np.array_equal(got_f, want_f) and np.array_equal(got_u8, want_u8)


# ------------------------------------------------------------------ guided filter
def test_box_mean_matches_float64_and_is_exact_on_integers():
    rng = np.random.default_rng(4)
    ints = rng.integers(0, 65026, (37, 53)).astype(np.float32)
    for r in (1, 4, 20, 45):  # 45 > both image dimensions: multi-bounce BORDER_REFLECT
        got = co.box_mean_f32(ints, r)
        want = t0.box_mean_f64(ints, r)
        # integer inputs: sums are exact, the only roundings are *scale and ->float32
        k2 = (2 * r + 1) ** 2
        exact = np.rint(want * k2)
        assert np.array_equal(got, (exact * (1.0 / k2)).astype(np.float32))
    fl = rng.standard_normal((37, 53)).astype(np.float32)
    assert np.abs(co.box_mean_f32(fl, 6) - t0.box_mean_f64(fl, 6)).max() < 1e-6


@pytest.mark.parametrize("scn", [3, 1])
def test_gf_t1_close_to_float64_definition(scn):
    guide = synth.flat_guide_u8(64, 80, seed=21, cells=12)
    src = synth.reflectance_like_u8(64, 80, seed=22)
    src = src if scn == 3 else src[:, :, 0]
    got, qf = co.guided_filter(guide, src, 9, 3.0, return_float=True)
    want = t0.guided_filter_f64(guide, src, 9, 3.0)
    assert got.shape == src.shape
    # cancellation in cov = E[Ip]-E[I]E[p] at float32 limits the agreement
    assert np.abs(qf - want).max() < 0.5
    assert np.mean(np.abs(qf - want)) < 0.05
    assert np.array_equal(got, np.clip(np.rint(qf), 0, 255).astype(np.uint8))


def test_gf_known_answers():
    guide = synth.scene_u8(48, 48, seed=31)
    const = np.full((48, 48, 3), 140, np.uint8)
    # constant src: cov(I,p)=0 -> alpha=0, beta=mean(p)=p
    assert np.array_equal(co.guided_filter(guide, const, 7, 3.0), const)
    # huge eps: alpha -> 0, output = box(box(p))
    src = synth.scene_u8(48, 48, seed=32)
    got, qf = co.guided_filter(guide, src, 3, 1e12, return_float=True)
    bb = np.stack([t0.box_mean_f64(t0.box_mean_f64(src[:, :, c], 3), 3) for c in range(3)], 2)
    assert np.abs(qf - bb).max() < 1e-2
    with pytest.raises(ValueError):
        co.guided_filter(guide[:, :, 0], src, 3, 1.0)  # 1-channel guide not restated


# ------------------------------------------------------------------ CNN + colour pipeline
def test_srgb_lut_equals_reference_table():
    g = np.load(os.path.join(G, "colour_tables.npz"))
    assert np.array_equal(co.srgb_lut(), g["srgb_to_rgb_levels"].astype(np.float32))
    c = np.load(os.path.join(G, "caffe_blob.npz"))
    assert np.array_equal(co.srgb_lut(), c["blob_ramp_f32"][0, 0, 0])


def test_cnn_oracle_against_reference_plumbing_and_float64():
    g = np.load(os.path.join(G, "cnn_forward.npz"))
    w = g["weights"]
    assert w.shape == (4513,) and abs(float(w[-1]) - 0.24792) < 1e-4
    r, r8 = co.cnn_reflectance(g["bgr32"], w)
    # r32 = reference get_reflectance_caffe() around a float64 forward of the same weights
    assert np.abs(r - g["r32"]).max() < 2e-6
    r64 = t0.cnn_reflectance_f64(g["bgr32"], w)
    assert np.abs(r - r64).max() < 2e-6
    assert np.array_equal(r8, (r * 255).astype(np.uint8))
    ramp_r, _ = co.cnn_reflectance(g["ramp"], w)
    assert np.abs(ramp_r - g["r_ramp"]).max() < 2e-6
    probes = {0: 0.4797, 32: 0.5760, 64: 0.6776, 96: 0.7511, 128: 0.8179, 160: 0.8685,
              192: 0.9106, 224: 0.9459}  # SURVEY App. B grey-ramp response
    for v, want in probes.items():
        assert abs(float(ramp_r[0, v]) - want) < 1e-4


def test_r_png_bytes_match_reference_imwrite():
    d = np.load(os.path.join(G, "decompose_outputs.npz"))
    g = np.load(os.path.join(G, "cnn_forward.npz"))
    r, r8 = co.cnn_reflectance(d["scene"], g["weights"])
    assert np.abs(r - d["r"]).max() < 2e-6
    # the reference truncates r*255; 1-ulp differences in r may move a byte by one, rarely
    delta = r8.astype(int) - d["r_png"].astype(int)
    assert np.abs(delta).max() <= 1 and np.mean(delta != 0) < 0.01


def test_box_mean_against_scipy_uniform_filter():
    """Third-party cross-check of the box filter restatement: scipy.ndimage.uniform_filter with
    mode='reflect' (d c b a | a b c d | d c b a) is cv::BORDER_REFLECT, the border cv::boxFilter
    gets from guidedFilter."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(21)
    for h, w, r in ((40, 57, 3), (25, 31, 12), (9, 200, 45), (64, 64, 0)):
        plane = (rng.random((h, w)) * 255).astype(np.float32)
        want = ndi.uniform_filter(plane.astype(np.float64), size=2 * r + 1, mode="reflect")
        got = co.box_mean_f32(plane, r)
        assert np.abs(got - want).max() < 1e-3 * max(1.0, np.abs(want).max()) * 1e-1, (h, w, r)


def test_box_census_counts_rounded_operations():
    """The exactness census (rfo_census, tools/gf_exactness.py) leaves the filter's bytes alone, finds no
    rounded operation on smooth data, and does find them when one alpha/beta-like plane mixes
    magnitudes 2^60 apart (a double cannot hold both in one sum)."""
    import ctypes
    from tests import synth
    L = co.lib()
    L.rfo_census.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
    L.rfo_census.restype = None
    h, w, r = 60, 83, 7
    guide = synth.flat_guide_u8(h, w, seed=3, cells=9)
    src = synth.reflectance_like_u8(h, w, seed=4)[:, :, :1].copy()
    want = co.guided_filter(guide, src, r, 3.0)
    buf = (ctypes.c_ulonglong * 16)()
    L.rfo_census(1, None)
    try:
        got = co.guided_filter(guide, src, r, 3.0)
    finally:
        L.rfo_census(0, buf)
    assert np.array_equal(got, want)
    rows, rows_bad, row_ops, row_ops_bad, cols, cols_bad, col_ops, col_ops_bad, ok_rows, planes = buf[:10]
    assert planes == 4 and rows == 4 * h and cols == 4 * w
    assert row_ops == 4 * h * (2 * r + 1 + 2 * (w - 1)) and col_ops == 4 * w * (2 * r + 2 * h)
    assert rows_bad == 0 and row_ops_bad == 0 and ok_rows <= rows
    # the instrument itself: TwoSum flags a sum whose addends do not fit one double
    L.rfo_box_census_f32 = getattr(L, "rfo_box_census_f32")
    L.rfo_box_census_f32.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int]
    plane = np.full((12, 40), 1.0, np.float32)
    plane[5, 20] = np.float32(2.0 ** -60)
    out = np.empty_like(plane)
    L.rfo_census(1, None)
    L.rfo_box_census_f32(plane.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 12, 40, 3)
    L.rfo_census(0, buf)
    assert np.array_equal(out, co.box_mean_f32(plane, 3))
    assert buf[1] == 1 and buf[3] > 0 and buf[8] == 11        # one row rounds, and fails the test


def test_exact_rows_sufficient_test_implies_exact_chain():
    """The proof obligation of the GPU's exact-row stage 2 (debug option gf_exact; DESIGN.md 3.2,
    rf_gf_fused.hpp) as a property test, no GPU: over 10^5 random rows - smooth, alpha-like with zero
    crossings, huge dynamic range, denormals, zeros, signs, rows built to sit exactly on the
    limit - WHENEVER a row passes the sufficient test as the GPU evaluates it, not one operation of
    RowSum<float,double>'s chain rounds, and the row sum rebuilt from the sums of aligned 16-column
    blocks (any order of exact additions) is the chain's double at every 16th column.  Rows that
    fail the test may or may not round: the test is sufficient, not necessary - but some of the
    failing rows must round, or the instrument could not tell."""
    import ctypes
    L = co.lib()
    fn = L.rfo_exact_rows_check
    i32p, f32p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
    fn.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, i32p, i32p]
    fn.restype = None
    rng = np.random.default_rng(2024)
    total = passed = failed_and_rounded = 0
    for w, r, rows in ((256, 45, 30000), (96, 45, 20000), (320, 52, 20000), (64, 7, 15000),
                       (128, 96, 10000), (48, 16, 5000)):
        kind = rng.integers(0, 8, rows)
        x = rng.standard_normal((rows, w)).astype(np.float32)
        scale = np.exp2(rng.integers(-30, 30, (rows, 1))).astype(np.float32)
        wide = np.exp2(rng.integers(-60, 20, (rows, w)).astype(np.float64)).astype(np.float32)
        plane = x * scale                                            # kind 0, 1: one magnitude per row
        plane = np.where(kind[:, None] == 2, x * wide, plane)        # huge dynamic range inside a row
        plane = np.where(kind[:, None] == 3, np.cumsum(x, 1).astype(np.float32) * np.float32(1e-3), plane)
        den = (rng.integers(1, 1 << 20, (rows, w)).astype(np.float64) * 2.0 ** -149).astype(np.float32)
        plane = np.where(kind[:, None] == 4, den * np.sign(x), plane)            # denormals
        plane = np.where((kind[:, None] == 5) & (rng.random((rows, w)) < 0.7), np.float32(0), plane)
        # kind 6: integers times one power of two plus ONE tiny value: right at / just past the limit
        ints = rng.integers(-(1 << 12), 1 << 12, (rows, w)).astype(np.float32)
        edge = ints * scale
        tiny = (scale[:, 0] * np.exp2(rng.integers(-40, -8, rows)).astype(np.float32))
        edge[np.arange(rows), rng.integers(0, w, rows)] = tiny
        plane = np.where(kind[:, None] == 6, edge, plane)
        plane = np.where(kind[:, None] == 7, ints, plane)                        # exact integers
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        ok = np.zeros(rows, np.int32)
        bad = np.zeros(rows, np.int32)
        eq = np.zeros(rows, np.int32)
        fn(plane.ctypes.data_as(f32p), rows, w, r, ok.ctypes.data_as(i32p), bad.ctypes.data_as(i32p),
           eq.ctypes.data_as(i32p))
        sel = ok != 0
        assert not bad[sel].any(), (w, r, "a row that passes the test rounds")
        assert eq[sel].all(), (w, r, "block sums differ from the chain on a row that passes")
        total += rows
        passed += int(sel.sum())
        failed_and_rounded += int((bad[~sel] != 0).sum())
    assert total >= 100000 and passed > total // 4 and failed_and_rounded > 100
    # a row with an infinity or a NaN never passes (its chain would carry the NaN to the row's end,
    # a window sum only while the value is inside the window)
    bad_rows = np.ones((3, 64), np.float32)
    bad_rows[0, 5], bad_rows[1, 9], bad_rows[2, :] = np.inf, np.nan, np.inf
    ok = np.ones(3, np.int32)
    fn(bad_rows.ctypes.data_as(f32p), 3, 64, 7, ok.ctypes.data_as(i32p),
       np.zeros(3, np.int32).ctypes.data_as(i32p), np.zeros(3, np.int32).ctypes.data_as(i32p))
    assert not ok.any()


def test_cnn_oracle_against_torch_cpu_conv():
    """Third-party arithmetic for the CNN forward (a7): torch's own CPU float32 convolution, ReLU,
    concatenation and sigmoid on the reference's weights and preprocessing
    (/root/reference/network_definition.prototxt:9-165 spelled with torch.nn.functional) - not
    Caffe, but a float32 implementation nobody in this repository wrote.  Summation orders differ
    (BLAS blocking against the oracle's k-ascending FMA chain), so the bound is a few float32 ulps
    of a value below 1, and the uint8 map may move by one in a few pixels."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    g = np.load(os.path.join(G, "cnn_forward.npz"))
    wts = torch.from_numpy(g["weights"].astype(np.float32))
    lut = torch.from_numpy(co.srgb_lut())
    for key in ("bgr32", "ramp"):
        bgr = g[key]
        r, r8 = co.cnn_reflectance(bgr, g["weights"])
        rgb = torch.from_numpy(np.ascontiguousarray(bgr[:, :, ::-1])).long()
        x = lut[rgb].permute(2, 0, 1).unsqueeze(0)                      # 1 x 3 x H x W, linear
        cur = F.relu(F.conv2d(x, wts[:96].reshape(32, 3, 1, 1), wts[96:128]))
        feats, q = [cur], 128
        for _ in range(4):
            cur = F.relu(F.conv2d(cur, wts[q:q + 1024].reshape(32, 32, 1, 1), wts[q + 1024:q + 1056]))
            q += 1056
            feats.append(cur)
        z = F.conv2d(torch.cat(feats, 1), wts[q:q + 160].reshape(1, 160, 1, 1), wts[q + 160:q + 161])
        rt = torch.sigmoid(z)[0, 0].numpy()
        assert rt.dtype == np.float32
        assert np.abs(rt - r).max() < 1e-5, key
        delta = (rt * 255).astype(np.uint8).astype(int) - r8.astype(int)
        assert np.abs(delta).max() <= 1 and np.mean(delta != 0) < 0.01
