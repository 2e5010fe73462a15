"""CV_32F variants of the two filters.

CPU: the C oracle's float paths against the float64 definitions and against the 8-bit path on
integer-valued data.  GPU: rf_jbf_f32 / rf_gf_f32 bit for bit against the oracle."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests import synth


def _f(img):
    return img.astype(np.float32) / np.float32(255)


def _jbf_f64(joint, src, radius, sc, ss, border):
    """Definition with the exact colour weight (no table), float64."""
    h, w = src.shape[:2]
    pj = np.pad(joint.astype(np.float64), ((radius, radius), (radius, radius), (0, 0)), mode=border)
    ps = np.pad(src.astype(np.float64), ((radius, radius), (radius, radius), (0, 0)), mode=border)
    num = np.zeros(src.shape, np.float64)
    den = np.zeros((h, w, 1), np.float64)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            if i * i + j * j > radius * radius:
                continue
            tj = pj[radius + i:radius + i + h, radius + j:radius + j + w]
            ts = ps[radius + i:radius + i + h, radius + j:radius + j + w]
            alpha = np.abs(tj - joint).sum(axis=2, keepdims=True)
            wgt = np.exp(-(i * i + j * j) / (2 * ss * ss)) * np.exp(-alpha * alpha / (2 * sc * sc))
            num += wgt * ts
            den += wgt
    return num / den


def test_oracle_jbf_f32_close_to_definition():
    joint = _f(synth.scene_u8(30, 41, seed=3))
    src = _f(synth.scene_u8(30, 41, seed=4))
    got = co.joint_bilateral_filter_f32(joint, src, 9, 0.08, 3.0)
    want = _jbf_f64(joint, src, 4, 0.08, 3.0, "reflect")
    # 4096-bin linear interpolation of exp + float32 accumulation of 49 taps
    assert np.abs(got - want).max() < 2e-5
    g1 = co.joint_bilateral_filter_f32(joint[:, :, 1], src[:, :, 0], 9, 0.08, 3.0)
    w1 = _jbf_f64(joint[:, :, 1:2], src[:, :, 0:1], 4, 0.08, 3.0, "reflect")[:, :, 0]
    assert g1.shape == (30, 41) and np.abs(g1 - w1).max() < 2e-5
    with pytest.raises(NotImplementedError):
        co.joint_bilateral_filter_f32(np.ones((6, 6, 3), np.float32), src[:6, :6], 5, 1.0, 1.0)


def test_oracle_gf_f32_is_the_float_core_of_the_8bit_path():
    guide = synth.scene_u8(50, 70, seed=5)
    src = synth.scene_u8(50, 70, seed=6)
    _, qf = co.guided_filter(guide, src, 6, 3.0, return_float=True)
    assert np.array_equal(co.guided_filter_f32(guide.astype(np.float32), src.astype(np.float32),
                                               6, 3.0), qf)
    # scale covariance: on [0,1] data with eps/255^2 the result is the 0..255 one / 255 up to
    # rounding - as long as eps/255^2 stays >= 1e-2 (below that OpenCV replaces near-zero
    # determinants by 1, which float images in [0,1] hit all the time)
    _, qbig = co.guided_filter(guide, src, 6, 1000.0, return_float=True)
    q01 = co.guided_filter_f32(_f(guide), _f(src), 6, 1000.0 / 255.0 ** 2)
    assert np.abs(q01 * 255 - qbig).max() < 2e-2


# ------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def gpu(built):
    import torch
    import reflectance_filtering_amd as rf
    if not torch.cuda.is_available():
        pytest.skip("no HIP device visible")
    return rf, torch


@pytest.mark.gpu
@pytest.mark.parametrize("jcn,scn", [(3, 3), (3, 1), (1, 3), (1, 1)])
def test_gpu_jbf_f32_bitwise(gpu, jcn, scn):
    rf, torch = gpu
    h, w = 83, 120
    joint = _f(synth.scene_u8(h, w, seed=jcn * 10 + scn))[:, :, :jcn]
    src = _f(synth.scene_u8(h, w, seed=7))[:, :, :scn]
    for d, sc, ss, border in ((-1, 0.1, 4.0, 4), (7, 0.03, 2.0, 1), (-1, 0.5, 7.5, 2), (5, 0.2, 1.0, 3)):
        got = rf.ops.joint_bilateral_f32(torch.from_numpy(np.ascontiguousarray(joint[None])).cuda(),
                                         torch.from_numpy(np.ascontiguousarray(src[None])).cuda(),
                                         d, sc, ss, border=border).cpu().numpy()[0]
        want = co.joint_bilateral_filter_f32(joint, src, d, sc, ss, border=border)
        assert np.array_equal(got, want.reshape(got.shape)), (d, sc, ss, border)
        # the register-tiled kernel (default) against the one-thread-per-pixel kernel
        with rf._ffi.debug_options(jbf_f32_untiled=1):
            ref = rf.ops.joint_bilateral_f32(
                torch.from_numpy(np.ascontiguousarray(joint[None])).cuda(),
                torch.from_numpy(np.ascontiguousarray(src[None])).cuda(), d, sc, ss,
                border=border).cpu().numpy()[0]
        assert np.array_equal(got, ref), (d, sc, ss, border)


@pytest.mark.gpu
def test_gpu_jbf_f32_batch_ranges_and_ximgproc(gpu):
    """Every image of a batch gets its own table (value range); negative values; the cv2-shaped
    entry point dispatches on dtype; a constant joint is refused like the oracle does."""
    rf, torch = gpu
    h, w = 40, 52
    joints = np.stack([_f(synth.scene_u8(h, w, seed=1)), _f(synth.scene_u8(h, w, seed=2)) * 3 - 1,
                       _f(synth.scene_u8(h, w, seed=3)) * np.float32(1e-3)])
    srcs = np.stack([_f(synth.scene_u8(h, w, seed=4 + i)) for i in range(3)])
    got = rf.ops.joint_bilateral_f32(torch.from_numpy(joints).cuda(), torch.from_numpy(srcs).cuda(),
                                     -1, 0.2, 3.0).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], co.joint_bilateral_filter_f32(joints[i], srcs[i], -1, 0.2, 3.0)), i
    one = rf.ximgproc.jointBilateralFilter(joints[1], srcs[1], -1, 0.2, 3.0)
    assert one.dtype == np.float32 and np.array_equal(one, got[1])
    with pytest.raises(ValueError):
        rf.ximgproc.jointBilateralFilter(joints[0], (srcs[0] * 255).astype(np.uint8), -1, 0.2, 3.0)
    with pytest.raises(ValueError, match="border"):     # BORDER_CONSTANT: undefined in OpenCV's 32F path
        rf.ops.joint_bilateral_f32(torch.from_numpy(joints).cuda(), torch.from_numpy(srcs).cuda(),
                                   -1, 0.2, 3.0, border=0)
    with pytest.raises(ValueError, match="constant joint"):
        rf.ops.joint_bilateral_f32(torch.ones((1, 8, 8, 3), device="cuda"),
                                   torch.from_numpy(srcs[:1, :8, :8]).cuda().contiguous(), -1, 1.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,r,eps,scn", [(60, 90, 5, 1e-3, 3), (33, 200, 45, 4e-5, 1),
                                           (129, 70, 12, 0.0, 3), (5, 7, 3, 1e-2, 1),
                                           (240, 320, 52, 1e-4, 3)])
def test_gpu_gf_f32_bitwise(gpu, h, w, r, eps, scn):
    rf, torch = gpu
    guide = _f(synth.flat_guide_u8(h, w, seed=r, cells=12) if r % 2 else synth.scene_u8(h, w, seed=r))
    src = _f(synth.scene_u8(h, w, seed=h))[:, :, :scn]
    got = rf.ops.guided_filter_f32(torch.from_numpy(guide[None]).cuda(),
                                   torch.from_numpy(np.ascontiguousarray(src[None])).cuda(), r, eps)
    want = co.guided_filter_f32(guide, src, r, eps)
    assert np.array_equal(got.cpu().numpy()[0], want.reshape(h, w, scn), equal_nan=True)


@pytest.mark.gpu
def test_gpu_gf_f32_iterations_batch_and_ximgproc(gpu):
    rf, torch = gpu
    h, w = 70, 95
    guides = np.stack([_f(synth.scene_u8(h, w, seed=i)) for i in range(3)])
    srcs = np.stack([_f(synth.scene_u8(h, w, seed=10 + i)) for i in range(3)])
    got = rf.ops.guided_filter_f32(torch.from_numpy(guides).cuda(), torch.from_numpy(srcs).cuda(),
                                   8, 1e-3, iterations=2).cpu().numpy()
    for i in range(3):
        want = co.guided_filter_f32(guides[i], co.guided_filter_f32(guides[i], srcs[i], 8, 1e-3),
                                    8, 1e-3)
        assert np.array_equal(got[i], want), i
    one = rf.ximgproc.guidedFilter(guides[0], srcs[0][:, :, 0], 8, 1e-3)
    assert one.dtype == np.float32 and one.shape == (h, w)
    assert np.array_equal(one, co.guided_filter_f32(guides[0], srcs[0][:, :, 0], 8, 1e-3))
    # integer-valued float data reproduces the 8-bit path before its rounding
    g8, s8 = synth.scene_u8(h, w, seed=20), synth.scene_u8(h, w, seed=21)
    q = rf.ximgproc.guidedFilter(g8.astype(np.float32), s8.astype(np.float32), 8, 3.0)
    assert np.array_equal(np.clip(np.rint(q), 0, 255).astype(np.uint8),
                          rf.ximgproc.guidedFilter(g8, s8, 8, 3.0))


@pytest.mark.gpu
def test_guided_filter_ddepth(gpu):
    """cv2's dDepth argument: CV_32F on 8-bit inputs returns the float q the 8-bit path rounds
    (same values as the oracle's float result on the same bytes); CV_8U on float inputs rounds
    like saturate_cast; mixed depths run on the values as they are."""
    from tests import synth
    rf, torch = gpu
    guide = synth.flat_guide_u8(96, 130, seed=3, cells=12)
    src = synth.scene_u8(96, 130, seed=4)
    u8 = rf.ximgproc.guidedFilter(guide, src, 9, 3.0)
    want_u8, want_q = co.guided_filter(guide, src, 9, 3.0, return_float=True)
    assert np.array_equal(u8, want_u8)
    q = rf.ximgproc.guidedFilter(guide, src, 9, 3.0, dDepth=rf.ximgproc.CV_32F)
    assert q.dtype == np.float32 and np.array_equal(q, want_q)
    back = rf.ximgproc.guidedFilter(guide.astype(np.float32), src.astype(np.float32), 9, 3.0,
                                    dDepth=rf.ximgproc.CV_8U)
    assert back.dtype == np.uint8 and np.array_equal(back, want_u8)
    mixed = rf.ximgproc.guidedFilter(guide, src.astype(np.float32), 9, 3.0)
    assert mixed.dtype == np.float32 and np.array_equal(mixed, want_q)
    with pytest.raises(ValueError):
        rf.ximgproc.guidedFilter(guide, src, 9, 3.0, dDepth=2)


@pytest.mark.gpu
def test_gpu_jbf_f32_reference_radius_interior_tiles(gpu):
    """sigma_s = 22 (radius 33) on an image large enough for interior tiles (no border handling)
    next to border tiles, three-channel and single-channel: tiled == untiled == oracle on a crop."""
    rf, torch = gpu
    h, w = 150, 330
    joint = _f(synth.scene_u8(h, w, seed=21))
    src = _f(synth.scene_u8(h, w, seed=22))
    for jc, sc_ in ((3, 3), (1, 1)):
        j = np.ascontiguousarray(joint[:, :, :jc])
        s_ = np.ascontiguousarray(src[:, :, :sc_])
        jd, sd = torch.from_numpy(j[None]).cuda(), torch.from_numpy(s_[None]).cuda()
        got = rf.ops.joint_bilateral_f32(jd, sd, -1, 20 / 255.0, 22.0).cpu().numpy()[0]
        with rf._ffi.debug_options(jbf_f32_untiled=1):
            ref = rf.ops.joint_bilateral_f32(jd, sd, -1, 20 / 255.0, 22.0).cpu().numpy()[0]
        assert np.array_equal(got, ref)
        want = co.joint_bilateral_filter_f32(j, s_, -1, 20 / 255.0, 22.0)
        assert np.array_equal(got, want.reshape(got.shape))


@pytest.mark.gpu
def test_gpu_jbf_f32_non_finite_src_stays_inside_its_disk(gpu):
    """A NaN / Inf src texel reaches exactly the outputs whose disk holds it (OpenCV never reads a
    texel off the disk): the register-tiled kernel visits the whole bounding span of its four
    outputs and must not let 0 * Inf leak into the neighbours.  Tiled == untiled, NaNs included."""
    rf, torch = gpu
    h, w = 70, 96
    joint = np.ascontiguousarray(_f(synth.scene_u8(h, w, seed=31)))
    src = np.ascontiguousarray(_f(synth.scene_u8(h, w, seed=32))[:, :, :1])
    bad = [(20, 30, np.nan), (50, 61, np.inf), (5, 2, -np.inf)]
    for y, x, v in bad:
        src[y, x, 0] = v
    jd, sd = torch.from_numpy(joint[None]).cuda(), torch.from_numpy(src[None]).cuda()
    for d, ss in ((-1, 4.0), (9, 2.0)):
        radius = int(round(ss * 1.5)) if d <= 0 else d // 2
        got = rf.ops.joint_bilateral_f32(jd, sd, d, 0.2, ss).cpu().numpy()[0]
        with rf._ffi.debug_options(jbf_f32_untiled=1):
            ref = rf.ops.joint_bilateral_f32(jd, sd, d, 0.2, ss).cpu().numpy()[0]
        assert np.array_equal(got, ref, equal_nan=True), (d, ss)
        yy, xx = np.mgrid[0:h, 0:w]
        reach = np.zeros((h, w), bool)
        for y, x, _ in bad:
            reach |= (yy - y) ** 2 + (xx - x) ** 2 <= radius * radius
        assert np.isfinite(got[~reach]).all(), (d, ss)
