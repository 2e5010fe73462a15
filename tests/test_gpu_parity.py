"""GPU suite: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Sizes are chosen so the oracle finishes in seconds; full-size behaviour is covered through
locality (the bilateral filter of a crop with its halo equals the crop of the filtered image),
batch independence and idempotence-style properties.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def env(built):
    import torch
    import reflectance_filtering_amd as rf
    from oracle import c_oracle as co
    if not torch.cuda.is_available():
        pytest.skip("no HIP device visible (run with -m 'not gpu' on CPU-only machines)")
    rf._ffi.load_library()
    return rf, co, torch


def _dev(torch, *imgs):
    return [torch.from_numpy(np.ascontiguousarray(i if i.ndim == 4 else i[None])).cuda()
            for i in imgs]


# ------------------------------------------------------------------------------ JBF
@pytest.mark.parametrize("h,w,sc,ss", [(96, 160, 20, 22), (70, 45, 20, 22), (33, 129, 15, 28),
                                       (64, 64, 7.5, 3.3), (40, 200, 60, 10)])
def test_jbf_matches_oracle_bitwise(env, h, w, sc, ss):
    from tests import synth
    rf, co, torch = env
    joint = synth.scene_u8(h, w, seed=h + w)
    src = synth.reflectance_like_u8(h, w, seed=h * w)
    got = rf.ximgproc.jointBilateralFilter(joint, src, -1, sc, ss)
    want = co.joint_bilateral_filter(joint, src, -1, sc, ss)
    assert got.dtype == np.uint8 and got.shape == src.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("jcn,scn", [(1, 3), (3, 1), (1, 1)])
def test_jbf_channel_combinations(env, jcn, scn):
    from tests import synth
    rf, co, torch = env
    joint = synth.scene_u8(80, 100, seed=1)
    src = synth.scene_u8(80, 100, seed=2)
    joint = joint if jcn == 3 else joint[:, :, 2]
    src = src if scn == 3 else src[:, :, 0]
    got = rf.ximgproc.jointBilateralFilter(joint, src, -1, 20, 22)
    assert np.array_equal(got, co.joint_bilateral_filter(joint, src, -1, 20, 22))


def test_jbf_rgb_guidance_and_explicit_diameter(env):
    from tests import synth
    rf, co, torch = env
    joint = synth.scene_u8(120, 90, seed=3)
    src = synth.scene_u8(120, 90, seed=4)
    for d in (-1, 9, 31):
        got = rf.ximgproc.jointBilateralFilter(joint, src, d, 25, 6)
        assert np.array_equal(got, co.joint_bilateral_filter(joint, src, d, 25, 6)), d


def test_jbf_image_smaller_than_radius(env):
    from tests import synth
    rf, co, torch = env
    for h, w in ((9, 7), (1, 50), (50, 1), (2, 2)):
        joint = synth.scene_u8(h, w, seed=5)
        src = synth.scene_u8(h, w, seed=6)
        got = rf.ximgproc.jointBilateralFilter(joint, src, -1, 25, 22)
        assert np.array_equal(got, co.joint_bilateral_filter(joint, src, -1, 25, 22)), (h, w)


def test_jbf_border_types_flags_and_generic_kernel(env):
    from tests import synth
    rf, co, torch = env
    ffi = rf._ffi
    joint = synth.scene_u8(50, 70, seed=7)
    src = synth.scene_u8(50, 70, seed=8)
    j, s = _dev(torch, joint, src)
    for border in (ffi.BORDER_CONSTANT, ffi.BORDER_REPLICATE, ffi.BORDER_REFLECT, ffi.BORDER_WRAP,
                   ffi.BORDER_REFLECT_101):
        want = co.joint_bilateral_filter(joint, src, -1, 20, 8, border=border)
        for flags in (0, ffi.JBF_FORCE_GENERIC):
            got = rf.ops.joint_bilateral_u8(j, s, -1, 20, 8, border=border, flags=flags)
            assert np.array_equal(got[0].cpu().numpy(), want), (border, flags)
    want = co.joint_bilateral_filter(joint, src, -1, 20, 8, flags=co.FLAG_TRUE_DIVISION)
    got = rf.ops.joint_bilateral_u8(j, s, -1, 20, 8, flags=ffi.JBF_TRUE_DIVISION)
    assert np.array_equal(got[0].cpu().numpy(), want)
    # very wide colour kernel: the LUT has no zero tail (exercises the untruncated table)
    want = co.joint_bilateral_filter(joint, src, -1, 400, 5)
    assert np.array_equal(rf.ops.joint_bilateral_u8(j, s, -1, 400, 5)[0].cpu().numpy(), want)
    # radius 60 and 75: the slab kernel (round 6; radius 75 ran untiled before)
    want = co.joint_bilateral_filter(joint, src, -1, 20, 40)
    assert np.array_equal(rf.ops.joint_bilateral_u8(j, s, -1, 20, 40)[0].cpu().numpy(), want)
    want = co.joint_bilateral_filter(joint, src, -1, 20, 50)
    assert np.array_equal(rf.ops.joint_bilateral_u8(j, s, -1, 20, 50)[0].cpu().numpy(), want)


@pytest.mark.parametrize("ss,sc", [(28, 15), (34.5, 20), (24.3, 9), (36, 20), (40, 20), (42.9, 12),
                                   (45.4, 20), (47, 20), (48.2, 9), (49, 20)])
def test_jbf_wide_radius_tiles(env, ss, sc):
    """radius 42 / 52 / 36: the 176-texel row pitch (grey and colour tiles) and the pitch
    boundary; README.md:63 of the reference uses c15 s28.  Radius 54 / 60 / 64 / 68 (--sigma_spatial
    is a free float, /root/reference/filter_reflectance.py:117-119): the slab kernel (round 6; tap rows
    in slabs; the grey loop, or one pass of the colour loop on 6-byte texels) at pitch 208; radius 70 / 72 / 74: the same at
    pitch 240 (radius 74 - sigma 49 - ran one thread per pixel until round 6).  Grey, colour, 1-channel
    and mixed grey / colour tiles."""
    from tests import synth
    rf, co, torch = env
    joint = synth.flat_guide_u8(100, 150, seed=int(ss), cells=30)
    grey = synth.reflectance_like_u8(100, 150, seed=3)
    colour = synth.scene_u8(100, 150, seed=4)
    mixed = grey.copy()
    mixed[30:70, 60:110] = colour[30:70, 60:110]           # colour and grey 64x64 tiles in one image
    for src in (grey, colour, mixed, np.ascontiguousarray(grey[:, :, 0])):
        got = rf.ximgproc.jointBilateralFilter(joint, src, -1, sc, ss)
        assert np.array_equal(got, co.joint_bilateral_filter(joint, src, -1, sc, ss))
    if ss >= 36:
        j1 = np.ascontiguousarray(joint[:, :, 1])            # 1-channel joint, REFLECT border, batch of 3
        import torch as _t
        jb = _t.from_numpy(np.stack([j1, j1[::-1].copy(), j1])[..., None].copy()).cuda()
        sb = _t.from_numpy(np.stack([colour, grey, mixed])).cuda()
        got = rf.ops.joint_bilateral_u8(jb, sb, -1, sc, ss, border=2).cpu().numpy()
        for i in range(3):
            want = co.joint_bilateral_filter(jb[i, :, :, 0].cpu().numpy(), sb[i].cpu().numpy(), -1, sc, ss,
                                             border=2)
            assert np.array_equal(got[i], want.reshape(got[i].shape)), i


def test_jbf_every_radius_of_the_tiled_kernels(env):
    """Radius 1 .. 73 one by one (explicit diameter 2 r + 1): since round 5 a tap row starts at the even
    column below its half-width and runs whole groups of four, so every radius walks its own pattern of
    row phases and group counts - through the 64x64 tiles (radius <= 52, three row pitches) and the slab
    kernel (53 .. 73: pitch 208 and 240).  Grey src and 1-channel buffers (the loop of the
    CNN -> BF(CNN,CNN) chain) at every radius, a colour src at every third."""
    from tests import synth
    rf, co, torch = env
    joint = synth.scene_u8(72, 88, seed=11)
    grey = synth.reflectance_like_u8(72, 88, seed=12)
    colour = synth.scene_u8(72, 88, seed=13)
    g1 = np.ascontiguousarray(grey[:, :, 0])
    g1_joint = g1.copy()          # (the same values in another buffer: passing src itself is OpenCV's
    for radius in range(1, 74):   #  bilateralFilter route, test_jbf_same_buffer_takes_...)
        d = 2 * radius + 1
        srcs = [grey, g1] + ([colour] if radius % 3 == 0 else [])
        for k, src in enumerate(srcs):
            jnt = g1_joint if k == 1 else joint
            got = rf.ximgproc.jointBilateralFilter(jnt, src, d, 20.0, 0.31 * radius + 0.7)
            want = co.joint_bilateral_filter(jnt, src, d, 20.0, 0.31 * radius + 0.7)
            assert np.array_equal(got, want), (radius, k)


@pytest.mark.parametrize("d,ss,border", [(109, 3.0, 4), (121, 50.0, 2), (129, 7.5, 0), (137, 22.0, 1),
                                         (139, 22.0, 3), (145, 22.0, 4), (147, 9.0, 2)])
def test_jbf_wide_diameter_with_any_sigma_and_border(env, d, ss, border):
    """The radius can also come from `d` (radius = d / 2 whatever sigma_spatial is): 54 / 60 / 64 / 68 / 69 /
    72 / 73 on the slab kernel, under every border mode, on an
    image smaller than the radius in one direction (multi-bounce reflection) and a ragged one."""
    from tests import synth
    rf, co, torch = env
    for h, w in ((37, 150), (131, 67)):
        joint = synth.scene_u8(h, w, seed=d)
        src = synth.scene_u8(h, w, seed=d + 1)
        j, s = _dev(torch, joint, src)
        got = rf.ops.joint_bilateral_u8(j, s, d, 25.0, ss, border=border)[0].cpu().numpy()
        want = co.joint_bilateral_filter(joint, src, d, 25.0, ss, border=border)
        assert np.array_equal(got, want), (h, w)


@pytest.mark.parametrize("sigma_space", [36.0, 40.0, 47.0])
def test_jbf_wide_radius_against_the_untiled_kernel(env, sigma_space):
    """Radius 54 / 60 / 70 (the slab kernel, which has a single tap loop and no test switches of
    its own) against the one-thread-per-pixel kernel (RF_JBF_FORCE_GENERIC) - an independent code path
    on the same device - with a colour src (the colour loop on 6-byte texels), a grey src and a
    single-channel joint, and against the oracle for the colour case."""
    from tests import synth
    rf, co, torch = env
    h, w = 150, 210
    joint = synth.scene_u8(h, w, seed=int(sigma_space))
    colour = synth.scene_u8(h, w, seed=int(sigma_space) + 1)
    grey = synth.reflectance_like_u8(h, w, seed=int(sigma_space) + 2)
    j, c, g = _dev(torch, joint, colour, grey)
    for src in (c, g):
        tiled = rf.ops.joint_bilateral_u8(j, src, -1, 15.0, sigma_space)
        plain = rf.ops.joint_bilateral_u8(j, src, -1, 15.0, sigma_space, flags=rf._ffi.JBF_FORCE_GENERIC)
        assert torch.equal(tiled, plain)
    j1 = j[..., :1].contiguous()
    assert torch.equal(rf.ops.joint_bilateral_u8(j1, g, -1, 15.0, sigma_space),
                       rf.ops.joint_bilateral_u8(j1, g, -1, 15.0, sigma_space,
                                                 flags=rf._ffi.JBF_FORCE_GENERIC))
    want = co.joint_bilateral_filter(joint, colour, -1, 15.0, sigma_space)
    assert np.array_equal(rf.ops.joint_bilateral_u8(j, c, -1, 15.0, sigma_space)[0].cpu().numpy(), want)


@pytest.mark.parametrize("radius", [73, 76, 77, 90, 100, 101, 117, 132])
def test_jbf_slab_radius_against_the_untiled_kernel_and_the_oracle(env, radius):
    """Radius 73..132 (round 6: the tile kernel in tap-row slabs, accumulators carried in registers
    from slab to slab; every row pitch 272 / 304 / 336 and both ends of each): a colour src, a grey
    src, a single-channel joint, every border mode the filter takes, an image smaller than the
    radius - against the one-thread-per-pixel kernel (RF_JBF_FORCE_GENERIC) and, for one case per
    radius, against the oracle."""
    from tests import synth
    rf, co, torch = env
    h, w = 100, 140
    joint = synth.scene_u8(h, w, seed=radius)
    colour = synth.scene_u8(h, w, seed=radius + 1)
    grey = synth.reflectance_like_u8(h, w, seed=radius + 2)
    j, c, g = _dev(torch, joint, colour, grey)
    d = 2 * radius + 1
    ss = radius / 2.5
    for src in (c, g):
        tiled = rf.ops.joint_bilateral_u8(j, src, d, 25.0, ss)
        plain = rf.ops.joint_bilateral_u8(j, src, d, 25.0, ss, flags=rf._ffi.JBF_FORCE_GENERIC)
        assert torch.equal(tiled, plain)
    j1 = j[..., :1].contiguous()
    assert torch.equal(rf.ops.joint_bilateral_u8(j1, g, d, 25.0, ss),
                       rf.ops.joint_bilateral_u8(j1, g, d, 25.0, ss, flags=rf._ffi.JBF_FORCE_GENERIC))
    for border in (rf._ffi.BORDER_REPLICATE, rf._ffi.BORDER_REFLECT, rf._ffi.BORDER_WRAP):
        assert torch.equal(rf.ops.joint_bilateral_u8(j, g, d, 25.0, ss, border=border),
                           rf.ops.joint_bilateral_u8(j, g, d, 25.0, ss, border=border,
                                                     flags=rf._ffi.JBF_FORCE_GENERIC)), border
    want = co.joint_bilateral_filter(joint, colour, d, 25.0, ss)
    assert np.array_equal(rf.ops.joint_bilateral_u8(j, c, d, 25.0, ss)[0].cpu().numpy(), want)
    small = j[:, :40, :50].contiguous()
    assert torch.equal(rf.ops.joint_bilateral_u8(small, small.clone(), d, 25.0, ss),
                       rf.ops.joint_bilateral_u8(small, small.clone(), d, 25.0, ss,
                                                 flags=rf._ffi.JBF_FORCE_GENERIC))


@pytest.mark.parametrize("radius", [133, 164, 165, 212, 213, 276, 277, 372, 373, 468, 469])
def test_jbf_slab_radius_beyond_132(env, radius):
    """Radius 133..468: the slab kernel at its coarser row pitches (400 / 496 / 624 / 816 / 1008), both
    ends of each, and 469 (the untiled kernel): a colour src, a grey src and a 1-channel joint against
    the one-thread-per-pixel kernel; the colour case against the oracle where the CPU gets there in
    seconds."""
    from tests import synth
    rf, co, torch = env
    h, w = 70, 90
    joint = synth.scene_u8(h, w, seed=radius)
    colour = synth.scene_u8(h, w, seed=radius + 1)
    grey = synth.reflectance_like_u8(h, w, seed=radius + 2)
    j, c, g = _dev(torch, joint, colour, grey)
    d = 2 * radius + 1
    ss = radius / 2.0
    for src in (c, g):
        tiled = rf.ops.joint_bilateral_u8(j, src, d, 30.0, ss)
        plain = rf.ops.joint_bilateral_u8(j, src, d, 30.0, ss, flags=rf._ffi.JBF_FORCE_GENERIC)
        assert torch.equal(tiled, plain)
    j1 = j[..., :1].contiguous()
    assert torch.equal(rf.ops.joint_bilateral_u8(j1, g, d, 30.0, ss, border=rf._ffi.BORDER_REFLECT),
                       rf.ops.joint_bilateral_u8(j1, g, d, 30.0, ss, border=rf._ffi.BORDER_REFLECT,
                                                 flags=rf._ffi.JBF_FORCE_GENERIC))
    if radius <= 213:
        want = co.joint_bilateral_filter(joint, colour, d, 30.0, ss)
        assert np.array_equal(rf.ops.joint_bilateral_u8(j, c, d, 30.0, ss)[0].cpu().numpy(), want)


def test_jbf_same_buffer_takes_opencvs_bilateral_filter_route(env):
    """cv2.ximgproc.jointBilateralFilter(a, a, ...) - one buffer as joint and src, or no joint - is
    routed by OpenCV to cv::bilateralFilter: for a 1-channel image the last step is a true division
    (the oracle's FLAG_TRUE_DIVISION), for 3 channels the bytes of the joint filter; two equal but
    separate buffers (the reference's two imreads) stay on the joint filter."""
    from tests import synth
    rf, co, torch = env
    # (the rounding hides the last-ulp difference almost everywhere: this seed has one pixel of 19,200
    #  where `sum / wsum` and `sum * (1.f / wsum)` round to different bytes)
    grey = np.ascontiguousarray(synth.scene_u8(120, 160, seed=3)[:, :, 1])
    colour = synth.scene_u8(61, 83, seed=13)
    want_div = co.joint_bilateral_filter(grey, grey, -1, 40, 4, flags=co.FLAG_TRUE_DIVISION)
    want_mul = co.joint_bilateral_filter(grey, grey, -1, 40, 4)
    assert not np.array_equal(want_div, want_mul)          # the case distinguishes the two routes
    assert np.array_equal(rf.ximgproc.jointBilateralFilter(grey, grey, -1, 40, 4), want_div)
    assert np.array_equal(rf.ximgproc.jointBilateralFilter(None, grey, -1, 40, 4), want_div)
    assert np.array_equal(rf.ximgproc.jointBilateralFilter(grey.copy(), grey, -1, 40, 4), want_mul)
    want3 = co.joint_bilateral_filter(colour, colour, -1, 30, 5)
    assert np.array_equal(rf.ximgproc.jointBilateralFilter(colour, colour, -1, 30, 5), want3)


def test_jbf_known_answers_on_device(env):
    rf, co, torch = env
    rng = np.random.default_rng(0)
    const = np.full((70, 130, 3), 99, np.uint8)
    noise = rng.integers(0, 256, (70, 130, 3), dtype=np.uint8)
    assert np.array_equal(rf.ximgproc.jointBilateralFilter(noise, const, -1, 20, 22), const)
    assert np.array_equal(rf.ximgproc.jointBilateralFilter(noise, noise.copy(), -1, 0.05, 22), noise)


def test_jbf_full_1080p_locality_and_batch(env):
    """BASELINE config C2 size: compare crops of the full-size result with the oracle run on
    the crop plus its halo (the filter is local), and check batch independence."""
    from tests import synth
    rf, co, torch = env
    h, w, r = 1080, 1920, 33
    joint = synth.scene_u8(h, w, seed=100)
    src = synth.reflectance_like_u8(h, w, seed=101)
    j, s = _dev(torch, joint, src)
    out = rf.ops.joint_bilateral_u8(j, s, -1, 20, 22)[0].cpu().numpy()
    for (y0, x0, ch, cw) in ((500, 900, 24, 40), (r, w - r - 30, 16, 30), (h - r - 20, r, 20, 25)):
        jc = joint[y0 - r:y0 + ch + r, x0 - r:x0 + cw + r]
        sc_ = src[y0 - r:y0 + ch + r, x0 - r:x0 + cw + r]
        want = co.joint_bilateral_filter(jc, sc_, -1, 20, 22)[r:r + ch, r:r + cw]
        assert np.array_equal(out[y0:y0 + ch, x0:x0 + cw], want), (y0, x0)
    # image borders: top-left and bottom-right corners including the reflected halo
    want = co.joint_bilateral_filter(joint[:60 + r, :50 + r], src[:60 + r, :50 + r], -1, 20, 22)
    assert np.array_equal(out[:60, :50], want[:60, :50])
    want = co.joint_bilateral_filter(joint[-(40 + r):, -(40 + r):], src[-(40 + r):, -(40 + r):],
                                     -1, 20, 22)
    assert np.array_equal(out[-40:, -40:], want[-40:, -40:])
    # a batch is filtered image by image
    j2 = torch.cat([j, torch.flip(j, dims=[1])])
    s2 = torch.cat([s, torch.flip(s, dims=[1])])
    o2 = rf.ops.joint_bilateral_u8(j2, s2, -1, 20, 22)
    assert torch.equal(o2[0], torch.from_numpy(out).cuda())
    alone = rf.ops.joint_bilateral_u8(j2[1:].contiguous(), s2[1:].contiguous(), -1, 20, 22)
    assert torch.equal(o2[1], alone[0])
    # (a flipped input does NOT give the flipped output bit for bit: the tap order is part of
    #  the contract and it is not flip-symmetric)


def test_jbf_randomised_sweep(env):
    """40 seeded random cases: sizes 1..170, both channel counts, every border type, radii on
    both sides of the tile/pitch limits, batches of 1..3."""
    from tests import synth
    rf, co, torch = env
    rng = np.random.default_rng(2024)
    for case in range(40):
        h, w = int(rng.integers(1, 171)), int(rng.integers(1, 171))
        n = int(rng.integers(1, 4))
        jcn, scn = int(rng.choice([1, 3])), int(rng.choice([1, 3]))
        border = int(rng.integers(0, 5))
        sc = float(rng.choice([3.0, 12.5, 20.0, 47.0, 130.0, 600.0]))
        ss = float(rng.choice([0.4, 1.0, 3.7, 9.0, 15.0, 22.0, 24.5, 30.0, 36.0]))
        d = int(rng.choice([-1, -1, -1, 3, 8, 21]))
        joints = np.stack([synth.scene_u8(h, w, seed=1000 * case + i) for i in range(n)])
        srcs = np.stack([synth.scene_u8(h, w, seed=1000 * case + 50 + i) if case % 3 else
                         synth.reflectance_like_u8(h, w, seed=1000 * case + 50 + i)
                         for i in range(n)])
        joints = joints if jcn == 3 else joints[..., :1]
        srcs = srcs if scn == 3 else srcs[..., 1:2]
        got = rf.ops.joint_bilateral_u8(torch.from_numpy(np.ascontiguousarray(joints)).cuda(),
                                        torch.from_numpy(np.ascontiguousarray(srcs)).cuda(), d, sc,
                                        ss, border=border).cpu().numpy()
        for i in range(n):
            want = co.joint_bilateral_filter(joints[i], srcs[i], d, sc, ss, border=border)
            assert np.array_equal(got[i], want.reshape(got[i].shape)), \
                (case, h, w, n, jcn, scn, border, sc, ss, d)


@pytest.mark.parametrize("sc,ss", [(20.0, 22.0), (15.0, 28.0), (9.0, 5.0), (30.0, 23.7)])
def test_jbf_row_pipeline(env, sc, ss):
    """The grey tap loop carries its software pipeline from one tap row into the next (the last
    group of a row prefetches the next row); against the compiler-scheduled loop (debug option), the
    64x64-only launch and the oracle, for several radii (different row shapes)."""
    from tests import synth
    rf, co, torch = env
    h, w = 150, 200
    joint = synth.scene_u8(h, w, seed=int(ss * 10))
    grey = synth.reflectance_like_u8(h, w, seed=int(sc))[:, :, :1].copy()
    jd, sd = _dev(torch, joint, grey)
    got = rf.ops.joint_bilateral_u8(jd, sd, -1, sc, ss)
    assert np.array_equal(got.cpu().numpy()[0], co.joint_bilateral_filter(joint, grey, -1, sc, ss))
    for opt in ("jbf_compiler_loop", "jbf_tile64_only"):
        with rf._ffi.debug_options(**{opt: 1}):
            assert torch.equal(rf.ops.joint_bilateral_u8(jd, sd, -1, sc, ss), got), opt
    # colour src: the hand-scheduled 6-byte-texel loop against the oracle and the compiler's loop
    colour = synth.scene_u8(h, w, seed=int(ss * 10) + 1)
    (cd,) = _dev(torch, colour)
    got3 = rf.ops.joint_bilateral_u8(jd, cd, -1, sc, ss)
    assert np.array_equal(got3.cpu().numpy()[0], co.joint_bilateral_filter(joint, colour, -1, sc, ss))
    with rf._ffi.debug_options(jbf_compiler_loop=1):
        assert torch.equal(rf.ops.joint_bilateral_u8(jd, cd, -1, sc, ss), got3)


def test_jbf_grey_joint_detected_at_run_time(env):
    """3-channel buffers whose joint (and src) have three equal channels - BF(CNN,CNN) through
    the 3-channel API - are recognised per tile and take the single-channel-joint loop; a tile
    with one coloured joint pixel must not.  All channel layouts against the oracle."""
    from tests import synth
    rf, co, torch = env
    h, w = 140, 200
    g = synth.reflectance_like_u8(h, w, seed=5)               # grey, 3 equal channels
    g2 = synth.reflectance_like_u8(h, w, seed=6)
    almost = g.copy()
    almost[70, 100, 1] ^= 0x08                                # one coloured pixel in one tile
    scene = synth.scene_u8(h, w, seed=7)
    cases = [(g, g2), (g, g), (almost, g2), (g, scene), (g[:, :, :1], g2), (g, g2[:, :, :1])]
    for joint, src in cases:
        jd, sd = _dev(torch, np.ascontiguousarray(joint), np.ascontiguousarray(src))
        if jd.data_ptr() == sd.data_ptr():
            sd = sd.clone()
        got = rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0).cpu().numpy()[0]
        want = co.joint_bilateral_filter(joint, src, -1, 20.0, 22.0)
        assert np.array_equal(got, want.reshape(got.shape)), (joint.shape, src.shape)
        with rf._ffi.debug_options(jbf_compiler_loop=1):
            assert torch.equal(rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0).cpu(),
                               torch.from_numpy(got[None]))


def test_jbf_strip_tiles(env):
    """Single-channel sources finish the last h % 64 rows with 32x128 / 16x256 tiles; every
    remainder class must match the oracle and the 64x64-only launch (debug option jbf_tile64_only)."""
    from tests import synth
    rf, co, torch = env
    w = 300
    for h in (7, 16, 17, 32, 33, 48, 49, 64, 64 + 13, 128 + 32, 64 + 40, 333):
        joint = synth.scene_u8(h, w, seed=h)
        grey = synth.reflectance_like_u8(h, w, seed=h + 1)[:, :, :1].copy()
        for jt, as_bgr in ((joint, False), (grey, True)):   # colour joint; grey joint as BGR
            jd, sd = _dev(torch, jt, grey)
            got = rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0, grey_as_bgr=as_bgr)
            j3 = jt if jt.shape[2] == 3 else np.repeat(jt, 3, axis=2)
            want = co.joint_bilateral_filter(j3, grey, -1, 20.0, 22.0)
            assert np.array_equal(got.cpu().numpy()[0], want), (h, as_bgr)
            with rf._ffi.debug_options(jbf_tile64_only=1):
                only64 = rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0, grey_as_bgr=as_bgr)
            assert torch.equal(only64, got), (h, as_bgr)
        if h > 64:
            continue  # (the 3-channel cases below on the small heights only, for time)
        # 3-channel sources (32-row strips only): a colour one and a grey one
        for src3 in (synth.scene_u8(h, w, seed=h + 2), np.repeat(grey, 3, axis=2)):
            jd, sd = _dev(torch, joint, src3)
            got = rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0)
            assert np.array_equal(got.cpu().numpy()[0],
                                  co.joint_bilateral_filter(joint, src3, -1, 20.0, 22.0)), h
            with rf._ffi.debug_options(jbf_tile64_only=1):
                assert torch.equal(rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0), got)


@pytest.mark.parametrize("h,w", [(500, 333), (130, 96), (200, 20), (129, 65), (333, 500),
                                 (77, 77), (140, 33)])
def test_jbf_right_and_bottom_strips(env, h, w):
    """Portrait / odd sizes: the last w % 64 columns of single-channel sources run as 128x32
    tiles, possibly together with bottom strips; bit-equal to the oracle and to 64x64 tiles."""
    from tests import synth
    rf, co, torch = env
    joint = synth.scene_u8(h, w, seed=h * 7 + w)
    grey = synth.reflectance_like_u8(h, w, seed=h + w)[:, :, :1].copy()
    for jt, as_bgr in ((joint, False), (grey, True)):
        jd, sd = _dev(torch, jt, grey)
        got = rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0, grey_as_bgr=as_bgr)
        j3 = jt if jt.shape[2] == 3 else np.repeat(jt, 3, axis=2)
        assert np.array_equal(got.cpu().numpy()[0],
                              co.joint_bilateral_filter(j3, grey, -1, 20.0, 22.0)), as_bgr
        with rf._ffi.debug_options(jbf_tile64_only=1):
            assert torch.equal(rf.ops.joint_bilateral_u8(jd, sd, -1, 20.0, 22.0,
                                                         grey_as_bgr=as_bgr), got)


def test_gf_randomised_sweep(env):
    from tests import synth
    rf, co, torch = env
    rng = np.random.default_rng(77)
    for case in range(24):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 700))
        n = int(rng.integers(1, 3))
        scn = int(rng.choice([1, 3]))
        r = int(rng.choice([0, 1, 2, 7, 19, 45, 52, 80, 120]))
        eps = float(rng.choice([1e-4, 0.5, 3.0, 7.0, 400.0]))
        iters = int(rng.choice([1, 1, 2]))
        guides = np.stack([synth.flat_guide_u8(h, w, seed=case * 10 + i, cells=15) if case % 2
                           else synth.scene_u8(h, w, seed=case * 10 + i) for i in range(n)])
        # grey (three equal channels: the one-channel fast path) and colour sources, mixed
        srcs = np.stack([synth.reflectance_like_u8(h, w, seed=case * 10 + 5 + i)
                         if (case + i) % 3 else synth.scene_u8(h, w, seed=case * 10 + 7 + i)
                         for i in range(n)])
        srcs = srcs if scn == 3 else srcs[..., :1]
        got = rf.ops.guided_filter_u8(torch.from_numpy(guides).cuda(),
                                      torch.from_numpy(np.ascontiguousarray(srcs)).cuda(), r, eps,
                                      iterations=iters).cpu().numpy()
        for i in range(n):
            want = srcs[i] if scn == 3 else srcs[i][:, :, 0]
            for _ in range(iters):
                want = co.guided_filter(guides[i], want, r, eps)
            assert np.array_equal(got[i], want.reshape(got[i].shape)), (case, h, w, n, scn, r, eps)


def test_empty_batch_aliasing_and_shutdown(env):
    from tests import synth
    rf, co, torch = env
    lib = rf._ffi.load_library()
    empty = torch.empty((0, 32, 32, 3), dtype=torch.uint8, device="cuda")
    assert rf.ops.joint_bilateral_u8(empty, empty.clone(), -1, 20, 22).shape == (0, 32, 32, 3)
    assert rf.ops.guided_filter_u8(empty, empty.clone(), 5, 1.0).shape == (0, 32, 32, 3)
    img = torch.from_numpy(synth.scene_u8(40, 40, seed=1)[None]).cuda()
    with pytest.raises(ValueError):
        rf.ops.joint_bilateral_u8(img, img.clone(), -1, 20, 22, out=img)   # dst aliases joint
    with pytest.raises(ValueError):
        rf.ops.joint_bilateral_u8(img[..., :2].contiguous(), img, -1, 20, 22)  # 2 channels
    with pytest.raises(ValueError):
        rf.ops.guided_filter_u8(img[..., :1].contiguous(), img, 5, 1.0)    # 1-channel guide
    a = rf.ops.joint_bilateral_u8(img, img.clone(), -1, 20, 6)
    assert lib.rf_shutdown() == 0          # parameter tables are rebuilt on demand
    b = rf.ops.joint_bilateral_u8(img, img.clone(), -1, 20, 6)
    assert torch.equal(a, b)
    # sigma <= 0 is OpenCV's "becomes 1" at the C boundary (the Python API rejects it earlier)
    c = rf.ops.joint_bilateral_u8(img, img.clone(), 5, 0.0, -3.0)
    want = co.joint_bilateral_filter(img[0].cpu().numpy(), img[0].cpu().numpy(), 5, 1.0, 1.0)
    assert np.array_equal(c[0].cpu().numpy(), want)


# ------------------------------------------------------------------------------ GF
@pytest.mark.parametrize("h,w,r,eps", [(256, 256, 52, 7.0), (256, 256, 45, 3.0), (130, 517, 9, 3.0),
                                       (64, 700, 20, 0.5), (37, 41, 45, 3.0), (90, 64, 1, 1e-3)])
def test_gf_matches_oracle_bitwise(env, h, w, r, eps):
    from tests import synth
    rf, co, torch = env
    guide = synth.flat_guide_u8(h, w, seed=r, cells=25) if r != 9 else synth.scene_u8(h, w, seed=9)
    src = synth.reflectance_like_u8(h, w, seed=h)
    got = rf.ximgproc.guidedFilter(guide, src, r, eps)
    want = co.guided_filter(guide, src, r, eps)
    assert got.dtype == np.uint8 and got.shape == src.shape
    assert np.array_equal(got, want)


def test_gf_degenerate_eps(env):
    """eps = 0 on a flat guide: zero determinants, infinities and NaNs must round the way
    OpenCV's cvRound/saturate_cast does on x86 (to 0), and tiny eps takes the det = 1 branch."""
    from tests import synth
    rf, co, torch = env
    guide = np.full((40, 60, 3), 90, np.uint8)
    guide[:, 30:] = (10, 200, 77)
    src = synth.scene_u8(40, 60, seed=1)
    for eps in (0.0, 1e-7, 5e-3):
        got = rf.ximgproc.guidedFilter(guide, src, 4, eps)
        assert np.array_equal(got, co.guided_filter(guide, src, 4, eps)), eps


def test_gf_grey_and_colour_sources_in_one_batch(env):
    """A 3-channel src with identical channels takes the one-channel path (result written three
    times); one differing byte anywhere - first pixel, middle, the very last pixel of an image
    whose pixel count is not a multiple of 4 - must send the image down the colour path."""
    from tests import synth
    rf, co, torch = env
    h, w = 37, 41
    guide = synth.scene_u8(h, w, seed=3)
    grey = synth.reflectance_like_u8(h, w, seed=4)
    srcs = [grey.copy() for _ in range(6)]
    srcs[1][0, 0, 1] ^= 0x40
    srcs[2][h // 2, w // 3, 2] ^= 0x01
    srcs[3][h - 1, w - 1, 0] ^= 0x80
    srcs[4][h - 1, w - 2, 2] ^= 0x10
    srcs[5] = synth.scene_u8(h, w, seed=5)
    batch = np.stack(srcs)
    guides = np.stack([guide] * len(srcs))
    for iters in (1, 3):
        got = rf.ops.guided_filter_u8(torch.from_numpy(guides).cuda(),
                                      torch.from_numpy(batch).cuda(), 9, 3.0,
                                      iterations=iters).cpu().numpy()
        for i, s in enumerate(srcs):
            want = s
            for _ in range(iters):
                want = co.guided_filter(guide, want, 9, 3.0)
            assert np.array_equal(got[i], want), (iters, i)


def test_gf_single_channel_src_and_iterations(env):
    from tests import synth
    rf, co, torch = env
    guide = synth.flat_guide_u8(150, 210, seed=1)
    src = synth.reflectance_like_u8(150, 210, seed=2)
    got = rf.ximgproc.guidedFilter(guide, src[:, :, 0], 12, 3.0)
    assert got.shape == (150, 210)
    assert np.array_equal(got, co.guided_filter(guide, src[:, :, 0], 12, 3.0))
    # 3x GF with a uint8 hand-off between passes == three CLI runs of the reference
    want = src
    for _ in range(3):
        want = co.guided_filter(guide, want, 12, 3.0)
    g, s = _dev(torch, guide, src)
    out = rf.apply_filter_batch("guided", s, g, 3.0, 12.7, iterations=3)
    assert np.array_equal(out[0].cpu().numpy(), want)
    # in place (dst aliases src) and tiny workspace (one image in flight) give the same bytes
    batch_g = torch.cat([g, g, g])
    batch_s = torch.cat([s, torch.flip(s, dims=[2]), s])
    one = rf._ffi.load_library().rf_gf_workspace_bytes(1, 150, 210, 3, 3, 12)
    ws = torch.empty(one, dtype=torch.uint8, device="cuda")
    a = rf.ops.guided_filter_u8(batch_g, batch_s, 12, 3.0, iterations=2)
    b = batch_s.clone()
    rf.ops.guided_filter_u8(batch_g, b, 12, 3.0, iterations=2, out=b, workspace=ws)
    assert torch.equal(a, b) and torch.equal(a[0], a[2])


def test_gf_1080p_against_oracle(env):
    from tests import synth
    rf, co, torch = env
    guide = synth.flat_guide_u8(1080, 1920, seed=50, cells=60)
    for src in (synth.reflectance_like_u8(1080, 1920, seed=51), synth.scene_u8(1080, 1920, seed=52)):
        got = rf.ximgproc.guidedFilter(guide, src, 45, 3.0)
        assert np.array_equal(got, co.guided_filter(guide, src, 45, 3.0))


# ------------------------------------------------------------------------------ CNN
def test_cnn_matches_oracle_and_golden(env):
    from tests import synth
    rf, co, torch = env
    g = np.load(os.path.join(G, "cnn_forward.npz"))
    w = rf.weights.load_weights()
    imgs = [g["bgr32"], synth.scene_u8(333, 500, seed=77)]
    for img in imgs:
        r, r8 = rf.get_reflectance_batch(_dev(torch, img)[0])
        want_r, want_r8 = co.cnn_reflectance(img, w)
        r, r8 = r[0].cpu().numpy(), r8[0].cpu().numpy()
        # contract: within 2e-7 of the oracle (1-2 ulp of float32 in [0.5,1)); observed: equal
        assert np.abs(r - want_r).max() <= 2e-7
        d8 = np.abs(r8.astype(int) - want_r8.astype(int))
        assert d8.max() <= 1 and np.mean(d8 != 0) < 1e-4
    r, _ = rf.get_reflectance_batch(_dev(torch, g["bgr32"])[0])
    assert np.abs(r[0].cpu().numpy() - g["r32"]).max() < 2e-6  # reference plumbing + f64 forward
    net = rf.decompose_with_trained_CNN.ReflectanceNet()
    out = rf.get_reflectance_caffe(net, g["bgr32"])
    assert out.shape == (32, 32) and out.dtype == np.float32
    assert np.array_equal(out, r[0].cpu().numpy())


def test_grey_as_bgr_and_fused_chain(env):
    """1-channel joint/src with RF_JBF_GREY_AS_BGR == the 3-equal-channel images cv2.imread
    would hand to the filter; decompose_and_filter_batch == the two-CLI chain."""
    from tests import synth
    rf, co, torch = env
    grey = synth.reflectance_like_u8(150, 203, seed=5)[:, :, 0]
    joint = synth.scene_u8(150, 203, seed=6)[:, :, 1]
    g1, j1 = _dev(torch, grey[:, :, None], joint[:, :, None])
    want = co.joint_bilateral_filter(np.repeat(joint[:, :, None], 3, 2),
                                     np.repeat(grey[:, :, None], 3, 2), -1, 20, 22)
    for flags, tune in ((0, 0), (rf._ffi.JBF_FORCE_GENERIC, 0), (0, 1)):
        with rf._ffi.debug_options(jbf_tune=tune):       # tune 1: the 64xTH kernel
            got = rf.ops.joint_bilateral_u8(j1, g1, -1, 20, 22, flags=flags, grey_as_bgr=True)
        assert np.array_equal(got[0, :, :, 0].cpu().numpy(), want[:, :, 0]), flags
    # without the flag a 1-channel joint keeps OpenCV's 1-channel semantics (distance |d|)
    plain = rf.ops.joint_bilateral_u8(j1, g1, -1, 20, 22)
    assert np.array_equal(plain[0, :, :, 0].cpu().numpy(),
                          co.joint_bilateral_filter(joint, grey, -1, 20, 22))
    scenes = np.stack([synth.scene_u8(120, 171, seed=s) for s in (1, 2, 3)])
    r8, filt = rf.decompose_and_filter_batch(torch.from_numpy(scenes).cuda())
    w = rf.weights.load_weights()
    for i in range(3):
        _, want_r8 = co.cnn_reflectance(scenes[i], w)
        assert np.array_equal(r8[i].cpu().numpy(), want_r8)
        r3 = np.repeat(want_r8[:, :, None], 3, 2)
        assert np.array_equal(filt[i].cpu().numpy(),
                              co.joint_bilateral_filter(r3, r3.copy(), -1, 20, 22)[:, :, 0])


def test_batch_front_end_matches_single_image_tools(env, tmp_path):
    """The sharded multi-file front-end writes the same files as per-image CLI runs; two
    'ranks' together cover every file exactly once."""
    from tests import synth
    rf, co, torch = env
    from reflectance_filtering_amd import batch
    iu = rf.image_utils
    photos, preds, single, multi = (tmp_path / d for d in ("photos", "preds", "single", "multi"))
    for d in (photos, preds, single, multi):
        d.mkdir()
    sizes = [(60, 81), (60, 81), (45, 70), (60, 81), (45, 70)]
    for i, (h, w) in enumerate(sizes):
        iu.imwrite(str(photos / ("im%d.png" % i)), synth.scene_u8(h, w, seed=i))
    files = batch.expand_inputs([str(photos / "*.png")])
    for rank in range(2):
        batch.decompose_files(files, str(preds), rank=rank, world=2)
    for f in files:
        assert rf.decompose_with_trained_CNN.main(["--filename_in=" + f,
                                                   "--path_out=" + str(single)]) == 0
    for name in sorted(os.listdir(str(single))):
        assert np.array_equal(iu.imread(str(preds / name)), iu.imread(str(single / name))), name
    rfiles = batch.expand_inputs([str(preds / "*-r.png")])
    assert len(rfiles) == 5
    written = []
    for rank in range(2):
        written += batch.filter_files("guided", rfiles, str(photos / "{base}.png"), 3.0, 9.0,
                                      str(multi), iterations=2, rank=rank, world=2)
    assert len(written) == 5
    for f in rfiles:
        gui = batch.guidance_for(f, str(photos / "{base}.png"))
        cur = f
        for _ in range(2):
            rf.read_filter_write("guided", cur, gui, 3.0, 9.0, str(single))
            cur = rf.filter_reflectance.output_filename(cur, str(single), "guided", 3.0, 9.0)
        twin = os.path.join(str(multi), os.path.basename(cur))
        assert np.array_equal(iu.imread(twin), iu.imread(cur)), cur
    # bilateral: BF(CNN, CNN) (grey src and grey guidance: one-channel path with the guidance
    # counted as three equal channels) and BF(CNN, photo) (grey src, colour guidance)
    for pattern in (None, str(photos / "{base}.png")):
        out_dir = tmp_path / ("bf_self" if pattern is None else "bf_photo")
        out_dir.mkdir()
        for rank in range(2):
            batch.filter_files("bilateral", rfiles, pattern, 20.0, 5.0, str(out_dir), rank=rank,
                               world=2)
        for f in rfiles:
            rf.read_filter_write("bilateral", f, batch.guidance_for(f, pattern), 20.0, 5.0,
                                 str(single))
            name = os.path.basename(rf.filter_reflectance.output_filename(f, str(single),
                                                                          "bilateral", 20.0, 5.0))
            assert np.array_equal(iu.imread(str(out_dir / name)), iu.imread(str(single / name))), name


# ------------------------------------------------------------------------------ CLI chain
def test_cli_chain_bf_cnn_cnn(env, tmp_path):
    """BASELINE config C3 in miniature: decompose CLI -> `-r.png` -> filter CLI with the
    prediction as its own guidance, compared with the oracle run on the same files."""
    from tests import synth
    rf, co, torch = env
    iu = rf.image_utils
    scene = synth.scene_u8(111, 167, seed=123)
    src = str(tmp_path / "img.png")
    iu.imwrite(src, scene)
    assert rf.decompose_with_trained_CNN.main(["--filename_in=" + src,
                                               "--path_out=" + str(tmp_path)]) == 0
    r_png = iu.imread(str(tmp_path / "img-r.png"))
    _, want_r8 = co.cnn_reflectance(scene, rf.weights.load_weights())
    assert np.array_equal(r_png[:, :, 0], want_r8) and np.array_equal(r_png[:, :, 0], r_png[:, :, 2])
    for name in ("img-r_colorized.png", "img-s_colorized.png"):
        assert os.path.exists(str(tmp_path / name))
    rpath = str(tmp_path / "img-r.png")
    assert rf.filter_reflectance.main(["--filter_type=bilateral", "--sigma_color=20",
                                       "--sigma_spatial=22", "--filename_in=" + rpath,
                                       "--guidance_in=" + rpath, "--path_out=" + str(tmp_path)]) == 0
    out = iu.imread(str(tmp_path / "img-r_bilateral_c20.0s22.0.png"))
    assert np.array_equal(out, co.joint_bilateral_filter(r_png, r_png.copy(), -1, 20, 22))
    assert rf.filter_reflectance.main(["--filter_type=guided", "--sigma_color=3",
                                       "--sigma_spatial=45", "--filename_in=" + rpath,
                                       "--guidance_in=" + src, "--path_out=" + str(tmp_path)]) == 0
    out = iu.imread(str(tmp_path / "img-r_guided_c3.0s45.0.png"))
    assert np.array_equal(out, co.guided_filter(scene, r_png, 45, 3.0))


# ------------------------------------------------------------------------------ colourise
def test_colorize_matches_reference_bytes(env):
    """Device colorize + sRGB write path against bytes captured from the reference's own
    colorize/imwrite (tests/golden/colorize_write.npz, decompose_outputs.npz)."""
    rf, co, torch = env
    d = np.load(os.path.join(G, "colorize_write.npz"))
    cases = [(d[t + "_image"], d[t + "_r"], d[t + "_refl_png"], d[t + "_shading_png"])
             for t in ("natural", "dark", "holes", "tiny")]
    g = np.load(os.path.join(G, "decompose_outputs.npz"))
    cases.append((g["scene"], g["r"], g["r_colorized_png"], g["s_colorized_png"]))
    for img, r, want_refl, want_shad in cases:
        refl, shad = rf.ops.colorize_srgb_u8(torch.from_numpy(img[None]).cuda(),
                                             torch.from_numpy(r[None]).cuda())
        assert np.array_equal(refl.cpu().numpy()[0], want_refl)
        assert np.array_equal(shad.cpu().numpy()[0], want_shad)


def test_colorize_batch_against_oracle(env):
    """Batches (per-image percentiles), odd sizes, real CNN output as r, only one output."""
    from oracle import colorize_numpy as oc
    from tests import synth
    rf, co, torch = env
    rng = np.random.default_rng(12)
    for h, w, n in ((333, 500, 3), (7, 5, 4), (64, 257, 2), (1, 1, 2)):
        imgs = np.stack([synth.scene_u8(h, w, seed=h + i) for i in range(n)])
        imgs[0, : h // 2] //= 4                      # a dark half: different percentile per image
        if n > 2:
            imgs[2] = 0                              # black image: nothing to normalise
        dev = torch.from_numpy(imgs).cuda()
        r, _ = rf.get_reflectance_batch(dev)
        refl, shad = rf.ops.colorize_srgb_u8(dev, r)
        r_host = r.cpu().numpy()
        for i in range(n):
            want_refl, want_shad = oc.colorize_srgb_u8(imgs[i], r_host[i])
            assert np.array_equal(refl[i].cpu().numpy(), want_refl), (h, w, i)
            assert np.array_equal(shad[i].cpu().numpy(), want_shad), (h, w, i)
        only_r, none = rf.ops.colorize_srgb_u8(dev, r, want_shading=False)
        assert none is None and torch.equal(only_r, refl)
    # random r (not the CNN's) incl. tiny values
    imgs = rng.integers(0, 256, (2, 40, 30, 3), dtype=np.uint8)
    r = (10.0 ** rng.uniform(-4, 0, (2, 40, 30))).astype(np.float32)
    refl, shad = rf.ops.colorize_srgb_u8(torch.from_numpy(imgs).cuda(), torch.from_numpy(r).cuda())
    for i in range(2):
        want_refl, want_shad = oc.colorize_srgb_u8(imgs[i], r[i])
        assert np.array_equal(refl[i].cpu().numpy(), want_refl)
        assert np.array_equal(shad[i].cpu().numpy(), want_shad)


# ------------------------------------------------------------------------------ HIP graph
def test_captured_call_replays_the_chain(env):
    """A 3x guided-filter chain and a bilateral call captured into a HIP graph give the eager
    bytes, also after the input buffers are refilled."""
    from tests import synth
    rf, co, torch = env
    h, w = 96, 128
    g, s = _dev(torch, synth.flat_guide_u8(h, w, seed=1), synth.reflectance_like_u8(h, w, seed=2))
    out_gf, out_bf = torch.empty_like(s), torch.empty_like(s)
    ws = rf.ops.gf_workspace(1, h, w, 3, 9, s.device, torch)

    def chain():
        rf.ops.guided_filter_u8(g, s, 9, 3.0, iterations=3, out=out_gf, workspace=ws)
        rf.ops.joint_bilateral_u8(g, s, -1, 20.0, 5.0, out=out_bf)
        return out_gf, out_bf

    cap = rf.ops.CapturedCall(chain)
    for seed in (2, 7):
        s.copy_(torch.from_numpy(synth.reflectance_like_u8(h, w, seed=seed)[None]))
        a, b = cap.replay()
        torch.cuda.synchronize()
        want_gf = rf.ops.guided_filter_u8(g, s, 9, 3.0, iterations=3)
        want_bf = rf.ops.joint_bilateral_u8(g, s, -1, 20.0, 5.0)
        assert torch.equal(a, want_gf) and torch.equal(b, want_bf), seed


def test_captured_call_with_forked_guided_filter(env):
    """The guided filter forks a side stream inside the call for the second half of a batch and
    joins it before returning: the call can still be captured into a HIP graph, and the replay
    gives the eager one-stream bytes (radius 45: fused stage 2, two images = two halves)."""
    from tests import synth
    rf, co, torch = env
    h, w = 120, 150
    g = torch.from_numpy(np.stack([synth.flat_guide_u8(h, w, seed=k, cells=12) for k in (1, 2)])).cuda()
    s = torch.from_numpy(np.stack([synth.scene_u8(h, w, seed=k) for k in (3, 4)])).cuda()
    out = torch.empty_like(s)
    ws = rf.ops.gf_workspace(2, h, w, 3, 45, s.device, torch)
    with rf._ffi.debug_options(gf_one_stream=1):
        want = rf.ops.guided_filter_u8(g, s, 45, 3.0, iterations=2)
    # (the library forks from eight images on by itself; the switch makes two images fork)
    with rf._ffi.debug_options(gf_force_two_streams=1):
        cap = rf.ops.CapturedCall(lambda: rf.ops.guided_filter_u8(g, s, 45, 3.0, iterations=2,
                                                                   out=out, workspace=ws))
    got = cap.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # and a batch large enough to fork without the switch
    g8, s8 = g.repeat(4, 1, 1, 1), s.repeat(4, 1, 1, 1)
    out8 = torch.empty_like(s8)
    ws8 = rf.ops.gf_workspace(8, h, w, 3, 45, s.device, torch)
    cap8 = rf.ops.CapturedCall(lambda: rf.ops.guided_filter_u8(g8, s8, 45, 3.0, iterations=2,
                                                                out=out8, workspace=ws8))
    out8.zero_()
    got8 = cap8.replay()
    torch.cuda.synchronize()
    assert torch.equal(got8, want.repeat(4, 1, 1, 1))


def test_captured_call_with_cnn_and_raw_entry_point_rules(env):
    """The CNN forward on pre-packed weights (what ops.cnn_reflectance_u8 calls) keeps no state in
    the library, so the decompose -> filter chain can be captured; the one-call raw-weights entry
    point refuses its FIRST call on a capturing stream (it would allocate) and is capturable once
    that stream has a slot; its slot table stays bounded over many streams."""
    import ctypes
    from tests import synth
    rf, co, torch = env
    lib = rf._ffi.load_library()
    h, w = 64, 96
    img = torch.from_numpy(np.stack([synth.scene_u8(h, w, seed=k) for k in (1, 2)])).cuda()
    want_r, want_r8 = rf.ops.cnn_reflectance_u8(img)
    want_bf = rf.ops.joint_bilateral_u8(want_r8[..., None].contiguous(), want_r8[..., None].contiguous(),
                                        -1, 20.0, 5.0, grey_as_bgr=True)
    packed, lut = rf.ops._cnn_device_consts(torch, img.device, None)
    r = torch.empty((2, h, w), dtype=torch.float32, device="cuda")
    r8 = torch.empty((2, h, w, 1), dtype=torch.uint8, device="cuda")
    bf = torch.empty_like(r8)

    def chain():
        rc = lib.rf_cnn_reflectance_packed_u8(img.data_ptr(), r.data_ptr(), r8.data_ptr(), 2, h, w,
                                              packed.data_ptr(), lut.data_ptr(),
                                              rf._ffi.current_stream_ptr(torch))
        rf._ffi.check(rc, "rf_cnn_reflectance_packed_u8")
        rf.ops.joint_bilateral_u8(r8, r8, -1, 20.0, 5.0, out=bf, grey_as_bgr=True)
        return r, r8, bf

    cap = rf.ops.CapturedCall(chain)
    r.zero_(), r8.zero_(), bf.zero_()
    a, b, c = cap.replay()
    torch.cuda.synchronize()
    assert torch.equal(a, want_r) and torch.equal(b[..., 0], want_r8) and torch.equal(c, want_bf)

    raw = torch.from_numpy(rf.weights.load_weights()).cuda()

    def raw_call(stream_ptr):
        return lib.rf_cnn_reflectance_u8(img.data_ptr(), r.data_ptr(), None, 2, h, w,
                                         raw.data_ptr(), lut.data_ptr(), ctypes.c_void_p(stream_ptr))

    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc_first = raw_call(side.cuda_stream)       # a capturing stream is refused, nothing captured
    assert rc_first == rf._ffi.RF_E_UNSUPPORTED and b"captured" in lib.rf_last_error()
    assert raw_call(side.cuda_stream) == rf._ffi.RF_OK     # eager call makes the slot
    torch.cuda.synchronize()
    assert torch.equal(r, want_r)
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=side):
        # ... and still refused with a slot: a graph would bake in a pointer the library recycles
        assert raw_call(side.cuda_stream) == rf._ffi.RF_E_UNSUPPORTED
    streams = [torch.cuda.Stream() for _ in range(40)]      # more streams than slots
    for st in streams:
        assert raw_call(st.cuda_stream) == rf._ffi.RF_OK
    torch.cuda.synchronize()
    assert torch.equal(r, want_r)


def test_jbf_first_use_of_a_parameter_set_inside_a_capture(env):
    """asynchronous on the passed stream, no hidden synchronisation.  A (sigma_color,
    sigma_space) pair the process has never seen is used for the first time INSIDE a graph capture
    (default, global capture mode): its tables are allocated and uploaded under a relaxed capture mode
    on a stream of the library's own (nothing but kernels enters the graph); the replay gives the
    oracle's bytes, a second replay after the inputs changed the new bytes, an eager call on another
    stream right after the capture is right as well - and the entry the graph points into is PINNED:
    seventy more parameter sets (the cache holds 64) do not evict it, the graph still replays right."""
    from tests import synth
    rf, co, torch = env
    h, w = 90, 140
    joint = synth.scene_u8(h, w, seed=31)
    src = synth.reflectance_like_u8(h, w, seed=32)
    sc, ss = 17.0625, 9.8125              # used nowhere else in the suite: a cache miss
    j, s = _dev(torch, joint, src)
    out = torch.zeros_like(s)
    # fill the library's cache of parameter sets (64 per process) so that the captured first use also
    # EVICTS one: the evicted arrays must not be freed inside the capture (hipFree invalidates it)
    tiny = torch.zeros((1, 4, 4, 3), dtype=torch.uint8, device="cuda")
    for k in range(70):
        rf.ops.joint_bilateral_u8(tiny, tiny.clone(), -1, 31.0 + k / 64.0, 2.0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rf.ops.joint_bilateral_u8(j, s, -1, sc, ss, out=out)
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):        # eager, another stream, before the graph ever ran
        eager = rf.ops.joint_bilateral_u8(j, s, -1, sc, ss)
    other.synchronize()
    want = co.joint_bilateral_filter(joint, src, -1, sc, ss)
    assert np.array_equal(eager[0].cpu().numpy(), want)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), want)
    src2 = synth.reflectance_like_u8(h, w, seed=33)
    s.copy_(torch.from_numpy(src2[None]).cuda())
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), co.joint_bilateral_filter(joint, src2, -1, sc, ss))
    # (round 6) 70 other parameter sets go through the 64-entry cache - eager calls, which also free
    # whatever was retired: the captured entry must have stayed
    for k in range(70):
        rf.ops.joint_bilateral_u8(tiny, tiny.clone(), -1, 41.0 + k / 64.0, 2.0)
    torch.cuda.synchronize()
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), co.joint_bilateral_filter(joint, src2, -1, sc, ss))
    # a wide radius for the first time, captured too (row-band kernel, LDS probe already done)
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=side):
        rf.ops.joint_bilateral_u8(j, s, -1, 11.03125, 36.03125, out=out)
    graph2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out[0].cpu().numpy(),
                          co.joint_bilateral_filter(joint, src2, -1, 11.03125, 36.03125))


def test_gf_captured_in_the_default_capture_mode(env):
    """A guided-filter call that forks its side stream (8 images) and runs the staggered schedule's
    event chain is captured with torch's default (global) capture mode - the call creates events and
    may create its side stream while the caller's stream is capturing - and replays to the eager
    bytes, twice."""
    from tests import synth
    rf, co, torch = env
    h, w = 120, 200
    g = torch.from_numpy(np.stack([synth.flat_guide_u8(h, w, seed=k, cells=10) for k in range(8)])).cuda()
    s = torch.from_numpy(np.stack([synth.reflectance_like_u8(h, w, seed=20 + k) for k in range(8)])).cuda()
    want = rf.ops.guided_filter_u8(g, s, 9, 3.0, iterations=2)
    assert np.array_equal(want[3].cpu().numpy(), co.guided_filter(
        g[3].cpu().numpy(), co.guided_filter(g[3].cpu().numpy(), s[3].cpu().numpy(), 9, 3.0), 9, 3.0))
    out = torch.zeros_like(s)
    ws = rf.ops.gf_workspace(8, h, w, 3, 9, s.device, torch)
    for opts in ({}, {"gf_stagger": 1, "gf_parts": 4}):
        side = torch.cuda.Stream()      # a fresh caller stream: its side stream is made inside the capture
        graph = torch.cuda.CUDAGraph()
        with rf._ffi.debug_options(**opts):
            with torch.cuda.graph(graph, stream=side):
                rf.ops.guided_filter_u8(g, s, 9, 3.0, iterations=2, out=out, workspace=ws)
        for _ in range(2):
            out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want), opts


def test_gf_capture_on_one_thread_eager_on_another(env):
    """Side streams are per caller stream: while one thread captures a two-image guided-filter
    call into a graph (its side stream joins that capture), another thread calls the filter
    eagerly on its own stream - the eager results are right, the capture stays valid and its
    replay gives the eager bytes."""
    import threading
    from tests import synth
    rf, co, torch = env
    h, w = 200, 260
    mk = lambda f, seeds: torch.from_numpy(np.stack([f(h, w, seed=k) for k in seeds])).cuda()
    g1, s1 = mk(synth.flat_guide_u8, (1, 2)), mk(synth.scene_u8, (3, 4))
    g2, s2 = mk(synth.scene_u8, (5, 6)), mk(synth.reflectance_like_u8, (7, 8))
    with rf._ffi.debug_options(gf_one_stream=1):
        want1 = rf.ops.guided_filter_u8(g1, s1, 45, 3.0, iterations=2)
        want2 = rf.ops.guided_filter_u8(g2, s2, 45, 3.0, iterations=2)
    out1 = torch.empty_like(s1)
    ws1 = rf.ops.gf_workspace(2, h, w, 3, 45, s1.device, torch)
    ws2 = torch.empty_like(ws1)
    torch.cuda.synchronize()
    stop = threading.Event()
    errors, eager_runs = [], [0]

    def eager():
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                while not stop.is_set() or eager_runs[0] < 3:
                    got = rf.ops.guided_filter_u8(g2, s2, 45, 3.0, iterations=2, workspace=ws2)
                    st.synchronize()
                    if not torch.equal(got, want2):
                        errors.append("eager result differs")
                    eager_runs[0] += 1
                    if eager_runs[0] > 200:
                        break
        except Exception as exc:               # noqa: BLE001 - reported by the main thread
            errors.append(repr(exc))

    th = threading.Thread(target=eager)
    opts = rf._ffi.debug_options(gf_force_two_streams=1)   # two images fork the side stream
    opts.__enter__()
    th.start()
    try:
        side = torch.cuda.Stream()
        graphs = []
        for _ in range(3):
            graph = torch.cuda.CUDAGraph()
            # thread_local capture mode: the other thread's allocations and launches are its own business
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                rf.ops.guided_filter_u8(g1, s1, 45, 3.0, iterations=2, out=out1, workspace=ws1)
            graphs.append(graph)
    finally:
        stop.set()
        th.join()
        opts.__exit__(None, None, None)
    assert not errors, errors
    assert eager_runs[0] >= 3
    for graph in graphs:
        out1.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out1, want1)


def test_two_streams_do_not_share_scratch(env):
    """Guided-filter workspaces are cached per (device, stream) and every CNN call with its own
    weights packs them into a buffer of its own: calls in flight on two streams give the bytes of
    the one-stream calls (they used to share planes / one weight buffer per device)."""
    from tests import synth
    rf, co, torch = env
    h, w = 700, 900
    # two images per call: with the switch each call forks a side stream of the library for its
    # second image (one side stream per caller stream)
    ga = torch.from_numpy(np.stack([synth.flat_guide_u8(h, w, seed=1, cells=30),
                                    synth.flat_guide_u8(h, w, seed=11, cells=20)])).cuda()
    gb = torch.from_numpy(np.stack([synth.scene_u8(h, w, seed=2), synth.scene_u8(h, w, seed=12)])).cuda()
    sa = torch.from_numpy(np.stack([synth.scene_u8(h, w, seed=3), synth.scene_u8(h, w, seed=13)])).cuda()
    sb = torch.from_numpy(np.stack([synth.reflectance_like_u8(h, w, seed=4),
                                    synth.reflectance_like_u8(h, w, seed=14)])).cuda()
    want_a = rf.ops.guided_filter_u8(ga, sa, 45, 3.0, iterations=2)
    want_b = rf.ops.guided_filter_u8(gb, sb, 45, 3.0, iterations=2)
    wts = rf.weights.load_weights()
    wts2 = (wts * np.float32(0.97)).astype(np.float32)
    want_r1, _ = rf.ops.cnn_reflectance_u8(gb, weights=wts)
    want_r2, _ = rf.ops.cnn_reflectance_u8(gb, weights=wts2)
    assert not torch.equal(want_r1, want_r2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(4):
        with rf._ffi.debug_options(gf_force_two_streams=rnd % 2):   # with and without the inner fork
            with torch.cuda.stream(s1):
                a = rf.ops.guided_filter_u8(ga, sa, 45, 3.0, iterations=2)
                r1, _ = rf.ops.cnn_reflectance_u8(gb, weights=wts)
            with torch.cuda.stream(s2):
                b = rf.ops.guided_filter_u8(gb, sb, 45, 3.0, iterations=2)
                r2, _ = rf.ops.cnn_reflectance_u8(gb, weights=wts2)
            torch.cuda.synchronize()
            assert torch.equal(a, want_a) and torch.equal(b, want_b)
            assert torch.equal(r1, want_r1) and torch.equal(r2, want_r2)


def test_gf_guide_cache_switch(env):
    """Experiment switch gf_guide_cache (the first pass of an iterated call leaves the guide half of
    the per-pixel algebra in the workspace, later passes box-sum only the src quantities): the
    bytes of the default path, for grey / colour / mixed batches, fused and two-kernel stage 2, and
    with a workspace too small for the record (falls back to recomputing)."""
    from tests import synth
    rf, co, torch = env
    h, w = 150, 333
    guide = synth.flat_guide_u8(h, w, seed=5, cells=20)
    srcs = [synth.scene_u8(h, w, seed=6), synth.reflectance_like_u8(h, w, seed=7)]
    g = torch.from_numpy(np.stack([guide, guide])).cuda()
    s = torch.from_numpy(np.stack(srcs)).cuda()
    one = torch.from_numpy(srcs[1][None, :, :, :1].copy()).cuda()
    for radius in (9, 45, 97):
        want = rf.ops.guided_filter_u8(g, s, radius, 3.0, iterations=3)
        want1 = rf.ops.guided_filter_u8(g[:1], one, radius, 3.0, iterations=2)
        with rf._ffi.debug_options(gf_guide_cache=1):
            assert torch.equal(rf.ops.guided_filter_u8(g, s, radius, 3.0, iterations=3), want)
            assert torch.equal(rf.ops.guided_filter_u8(g[:1], one, radius, 3.0, iterations=2), want1)
            small = rf._ffi.load_library().rf_gf_workspace_bytes(1, h, w, 3, 3, radius)
            ws = torch.empty(small, dtype=torch.uint8, device="cuda")   # one image, 3 channels: room
            assert torch.equal(rf.ops.guided_filter_u8(g, s, radius, 3.0, iterations=3, workspace=ws), want)
    cur = srcs[0]
    for _ in range(3):
        cur = co.guided_filter(guide, cur, 45, 3.0)
    with rf._ffi.debug_options(gf_guide_cache=1):
        got = rf.ops.guided_filter_u8(g[:1], s[:1], 45, 3.0, iterations=3)
    assert np.array_equal(got[0].cpu().numpy(), cur)


@pytest.mark.parametrize("radius", [1, 2, 5, 7, 8, 13, 16, 17, 20, 30, 33, 47, 60, 64, 65, 77, 96, 97, 104,
                                    111, 112, 113, 120, 121, 127, 128])
def test_gf_fused_stage2_any_radius(env, radius):
    """The fused stage 2 is instantiated for every radius 1..128 (round 6: 97..120 ran the row-sum /
    column-sum pair, 121..128 the float kernels before; 128 is the last radius whose window sums fit
    32 bits; the pair stays behind the switch gf_two_kernel and is compared below):
    sub-tiles of unequal height (47 = 16 + 16 + 15), one sub-tile per half period (radius < 16),
    the step-by-step row walk of radii below 8, one wave per SIMD above 64 - against the oracle
    (grey, colour and 1-channel src; two chained passes) and against the two-kernel form on a
    shape with partial column blocks and fewer rows than the radius."""
    from tests import synth
    rf, co, torch = env
    eps = 3.0 if radius % 2 else 7.0
    h, w = 70, 83
    guide = synth.flat_guide_u8(h, w, seed=radius, cells=9)
    colour = synth.scene_u8(h, w, seed=radius + 1)
    grey = synth.reflectance_like_u8(h, w, seed=radius + 2)
    g = torch.from_numpy(np.stack([guide, guide])).cuda()
    s = torch.from_numpy(np.stack([colour, grey])).cuda()      # one colour, one grey image
    got = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=2).cpu().numpy()
    for i, src in enumerate((colour, grey)):
        want = co.guided_filter(guide, co.guided_filter(guide, src, radius, eps), radius, eps)
        assert np.array_equal(got[i], want), (radius, i)
    one = torch.from_numpy(grey[None, :, :, :1].copy()).cuda()
    got1 = rf.ops.guided_filter_u8(g[:1], one, radius, eps).cpu().numpy()[0]
    assert np.array_equal(got1, co.guided_filter(guide, grey[:, :, :1].copy(), radius, eps).reshape(got1.shape))
    for hh, ww in ((radius // 2 + 1, 130), (150, 2 * radius + 19)):
        gg = torch.from_numpy(synth.scene_u8(hh, ww, seed=3)[None]).cuda()
        ss = torch.from_numpy(synth.scene_u8(hh, ww, seed=4)[None]).cuda()
        a = rf.ops.guided_filter_u8(gg, ss, radius, eps, iterations=3)
        with rf._ffi.debug_options(gf_two_kernel=1):
            b = rf.ops.guided_filter_u8(gg, ss, radius, eps, iterations=3)
        assert torch.equal(a, b), (radius, hh, ww)


def test_gf_tiny_images_through_every_path(env):
    """Images far smaller than the window (multi-bounce borders, one column block, fewer rows than a
    sub-tile) through the fused path with the one-byte hand-off (grey 3-channel src, three passes),
    a large radius of the fused path (100; the two-kernel pair until round 6) and the float kernels (radius 150) - against the oracle."""
    from tests import synth
    rf, co, torch = env
    for (h, w) in ((1, 1), (1, 7), (5, 1), (3, 2), (17, 33)):
        guide = synth.scene_u8(h, w, seed=h * 100 + w)
        grey = synth.reflectance_like_u8(h, w, seed=h * 100 + w + 1)
        g = torch.from_numpy(guide[None]).cuda()
        s = torch.from_numpy(grey[None]).cuda()
        for radius, eps, iters in ((45, 3.0, 3), (100, 7.0, 2), (150, 3.0, 2)):
            got = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=iters).cpu().numpy()[0]
            cur = grey
            for _ in range(iters):
                cur = co.guided_filter(guide, cur, radius, eps)
            assert np.array_equal(got, cur), (h, w, radius)


@pytest.mark.parametrize("radius,eps", [(45, 3.0), (52, 7.0)])
def test_gf_switches_keep_the_bytes(env, radius, eps):
    """The alternative forms behind the round-4 switches give the default's bytes (which the other
    tests hold to the oracle): the chained column walk (no row-walk kernel; blocks hand their row
    sums to the right through tagged slots), three-channel hand-offs between passes instead of the
    one-byte image, and one stream - on a batch that mixes grey and colour images, three passes,
    widths that are not multiples of 16, more images than ticket queues."""
    from tests import synth
    rf, co, torch = env
    h, w = 150, 203
    n = 11
    guides = np.stack([synth.flat_guide_u8(h, w, seed=radius + i, cells=14) for i in range(n)])
    srcs = np.stack([synth.reflectance_like_u8(h, w, seed=100 + i) if i % 3 else
                     synth.scene_u8(h, w, seed=100 + i) for i in range(n)])
    g, s = torch.from_numpy(guides).cuda(), torch.from_numpy(srcs).cuda()
    want = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3)
    assert np.array_equal(want[1].cpu().numpy(),
                          co.guided_filter(guides[1], co.guided_filter(guides[1], co.guided_filter(
                              guides[1], srcs[1], radius, eps), radius, eps), radius, eps))
    # (round 5: the staggered two-stream schedule - stage 1 of one part chained behind stage 1 of the
    #  other by events - in 2 and 4 parts, stage 1 capped at 2 / 3 workgroups per CU, other segment counts)
    for opts in ({"gf_chained": 1}, {"gf_no_compact": 1}, {"gf_one_stream": 1},
                 {"gf_chained": 1, "gf_no_compact": 1, "gf_force_two_streams": 1},
                 {"gf_stagger": 1}, {"gf_stagger": 1, "gf_parts": 4, "gf_s1_cap": 2},
                 {"gf_stagger": 1, "gf_force_two_streams": 1, "gf_parts": 6, "gf_s1_min_wgs": 64},
                 {"gf_s1_cap": 3, "gf_s1_min_wgs": 4096}, {"gf_s1_cap": 1}, {"gf_s1_cap": 2},
                 # round 6: the rounds-1-5 strip geometry; the exact-row stage 2 (this width is not a
                 # multiple of 16: it must fall back to the row walk by itself)
                 {"gf_s1_legacy_strips": 1}, {"gf_exact": 1}, {"gf_cw_chan_run": 1},
                 {"gf_cw_chan_run": 4}, {"gf_cw_chan_run": 100000, "gf_one_stream": 1}):
        with rf._ffi.debug_options(**opts):
            got = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3)
        assert torch.equal(got, want), opts
    one = s[:, :, :, :1].contiguous()                      # 1-channel src through the chained walk
    want1 = rf.ops.guided_filter_u8(g, one, radius, eps)
    with rf._ffi.debug_options(gf_chained=1):
        assert torch.equal(rf.ops.guided_filter_u8(g, one, radius, eps), want1)


@pytest.mark.parametrize("radius,eps", [(45, 3.0), (52, 7.0)])
def test_gf_exact_rows_against_oracle(env, radius, eps):
    """The exact-row stage 2 (debug option gf_exact, rf_gf_fused.hpp): rows whose alpha/beta pass the
    exactness test take no row walk - stage 1 leaves block sums, the column walk starts its chains
    from them - and must give the ORACLE's bytes: widths that are multiples of 16 (narrower than one
    window, one strip, several strips), grey / colour / 1-channel src in one batch, three passes;
    and again with every row forced through the list path (gf_exact_all_flagged)."""
    from tests import synth
    rf, co, torch = env
    for (h, w), n in (((70, 96), 3), ((130, 256), 3), ((97, 1040), 2)):
        guides = np.stack([synth.flat_guide_u8(h, w, seed=radius + i, cells=14) for i in range(n)])
        srcs = np.stack([synth.reflectance_like_u8(h, w, seed=100 + i) if i % 2 == 0 else
                         synth.scene_u8(h, w, seed=100 + i) for i in range(n)])
        g, s = torch.from_numpy(guides).cuda(), torch.from_numpy(srcs).cuda()
        want = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3)
        for i in range(min(n, 2)):
            ref = srcs[i]
            for _ in range(3):
                ref = co.guided_filter(guides[i], ref, radius, eps)
            assert np.array_equal(want[i].cpu().numpy(), ref), (h, w, i)
        for opts in ({"gf_exact": 1}, {"gf_exact": 1, "gf_exact_all_flagged": 1},
                     {"gf_exact": 1, "gf_one_stream": 1}, {"gf_exact": 1, "gf_no_compact": 1}):
            with rf._ffi.debug_options(**opts):
                got = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3)
            assert torch.equal(got, want), (h, w, opts)
        one = s[:, :, :, :1].contiguous()
        want1 = rf.ops.guided_filter_u8(g, one, radius, eps, iterations=2)
        with rf._ffi.debug_options(gf_exact=1):
            assert torch.equal(rf.ops.guided_filter_u8(g, one, radius, eps, iterations=2), want1)


def test_gf_exact_rows_adversarial(env):
    """Rows that FAIL the exactness test must come out of the sequential row walk: (i) a noise guide
    with a tiny eps (alpha / beta spanning far more than 2^22 within a row), (ii) an image whose
    upper half is that noise and whose lower half is a flat guide with a smooth src (rows that pass
    and rows that fail in one image, block boundaries in between), (iii) an all-zero src (alpha and
    beta exactly 0: the all-zero rows pass).  Every case against the oracle."""
    from tests import synth
    rf, co, torch = env
    h, w, radius = 200, 256, 45
    rng = np.random.default_rng(77)
    noise_g = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    noise_s = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    flat_g = synth.flat_guide_u8(h, w, seed=3, cells=9)
    smooth = synth.reflectance_like_u8(h, w, seed=4)
    mixed_g, mixed_s = flat_g.copy(), smooth.copy()
    mixed_g[:h // 2], mixed_s[:h // 2] = noise_g[:h // 2], noise_s[:h // 2]
    zeros = np.zeros_like(smooth)
    cases = ((noise_g, noise_s, 1e-3), (mixed_g, mixed_s, 1e-3), (mixed_g, mixed_s, 3.0),
             (flat_g, zeros, 3.0), (noise_g, smooth, 1e-6))
    for k, (gd, sr, eps) in enumerate(cases):
        g = torch.from_numpy(gd[None].copy()).cuda()
        s = torch.from_numpy(sr[None].copy()).cuda()
        with rf._ffi.debug_options(gf_exact=1):
            got = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=2)[0].cpu().numpy()
        want = co.guided_filter(gd, co.guided_filter(gd, sr, radius, eps), radius, eps)
        assert np.array_equal(got, want), k


def test_gf_radius_beyond_the_8bit_kernels(env):
    """int(sigma_spatial) is a free parameter of the reference's tool
    (/root/reference/filter_reflectance.py:67-70,118): radii above 128 run the float kernels inside
    rf_gf_u8 and round every pass to uint8 - the oracle's bytes, through apply_filter, the batch
    operator (two passes, grey and colour and 1-channel src) and with a workspace for one image."""
    from tests import synth
    rf, co, torch = env
    h, w = 200, 300
    joint = synth.flat_guide_u8(h, w, seed=5, cells=12)
    image = synth.reflectance_like_u8(h, w, seed=6)
    got = rf.filter_reflectance.apply_filter("guided", image, joint, sigma_color=3.0,
                                             sigma_spatial=150)
    assert got.dtype == np.uint8 and got.shape == image.shape
    assert np.array_equal(got, co.guided_filter(joint, image, 150, 3.0))
    colour = synth.scene_u8(h, w, seed=7)
    g = torch.from_numpy(np.stack([joint, joint])).cuda()
    s = torch.from_numpy(np.stack([colour, image])).cuda()
    for radius, eps in ((129, 7.0), (260, 3.0)):
        got2 = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=2).cpu().numpy()
        for i, src in enumerate((colour, image)):
            want = co.guided_filter(joint, co.guided_filter(joint, src, radius, eps), radius, eps)
            assert np.array_equal(got2[i], want), (radius, i)
    one = torch.from_numpy(image[None, :, :, :1].copy()).cuda()
    lib = rf._ffi.load_library()
    need1 = lib.rf_gf_workspace_bytes(1, h, w, 3, 1, 129)
    ws = torch.empty(need1, dtype=torch.uint8, device="cuda")
    three = one.expand(3, -1, -1, -1).contiguous()
    got3 = rf.ops.guided_filter_u8(g[:1].expand(3, -1, -1, -1).contiguous(), three, 129, 7.0,
                                   workspace=ws).cpu().numpy()
    want1 = co.guided_filter(joint, image[:, :, :1].copy(), 129, 7.0).reshape(got3[0].shape)
    for i in range(3):
        assert np.array_equal(got3[i], want1)
    # radius 128 (8-bit kernels: 257^2 x 255^2 is the last window sum below 2^32 - a white image makes
    # every sum that large) and 129 (float kernels) agree with the oracle on either side
    a = rf.ops.guided_filter_u8(g[:1], s[1:], 128, 3.0).cpu().numpy()[0]
    assert np.array_equal(a, co.guided_filter(joint, image, 128, 3.0))
    white = np.full((h, w, 3), 255, np.uint8)
    wg = torch.from_numpy(white[None]).cuda()
    assert np.array_equal(rf.ops.guided_filter_u8(wg, wg.clone(), 128, 3.0)[0].cpu().numpy(),
                          co.guided_filter(white, white, 128, 3.0))


@pytest.mark.parametrize("radius,eps", [(45, 3.0), (52, 7.0)])
def test_gf_fused_stage2_in_place_chain_and_oracle(env, radius, eps):
    """The fused stage 2 (radius 45 / 52): colour and grey images in one batch, three chained
    passes, dst aliasing src, a workspace for one image - against the oracle chained three times
    and against the two-kernel stage 2."""
    from tests import synth
    rf, co, torch = env
    h, w = 210, 333                      # h not a multiple of the sub-tile, w not of 16
    guide = synth.flat_guide_u8(h, w, seed=radius, cells=25)
    srcs = [synth.scene_u8(h, w, seed=radius + 1), synth.reflectance_like_u8(h, w, seed=radius + 2)]
    g = torch.from_numpy(np.stack([guide, guide])).cuda()
    s = torch.from_numpy(np.stack(srcs)).cuda()
    want = []
    for src in srcs:
        cur = src
        for _ in range(3):
            cur = co.guided_filter(guide, cur, radius, eps)
        want.append(cur)
    got = rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3)
    assert np.array_equal(got.cpu().numpy(), np.stack(want))
    with rf._ffi.debug_options(gf_two_kernel=1):
        assert torch.equal(rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3), got)
    # stage 1 cut into row segments of any height (each restarts its exact integer window sums)
    for seg_rows in (1, 17, 64, 1000):
        with rf._ffi.debug_options(gf_seg_rows=seg_rows):
            assert torch.equal(rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3), got)
    # everything on the caller's stream / the second image on a side stream (the default forks
    # from eight images on)
    with rf._ffi.debug_options(gf_one_stream=1):
        assert torch.equal(rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3), got)
    with rf._ffi.debug_options(gf_force_two_streams=1):
        assert torch.equal(rf.ops.guided_filter_u8(g, s, radius, eps, iterations=3), got)
    one = rf._ffi.load_library().rf_gf_workspace_bytes(1, h, w, 3, 3, radius)
    ws = torch.empty(one, dtype=torch.uint8, device="cuda")
    inplace = s.clone()
    rf.ops.guided_filter_u8(g, inplace, radius, eps, iterations=3, out=inplace, workspace=ws)
    assert torch.equal(inplace, got)


def test_cnn_register_kernel_equals_lds_column_kernel(env):
    """The register-resident CNN kernel (default: op_sel-routed inputs, no LDS) and the LDS-column
    kernel of round 1 run the same FMA chains: identical float outputs and bytes, odd pixel count
    (the second pixel of the last lane is clamped), custom weights."""
    from tests import synth
    rf, co, torch = env
    imgs = torch.from_numpy(np.stack([synth.scene_u8(77, 131, seed=s) for s in (1, 2, 3)])).cuda()
    wts = rf.weights.load_weights()
    for w in (None, (wts * np.float32(1.01)).astype(np.float32)):
        r, r8 = rf.ops.cnn_reflectance_u8(imgs, weights=w)
        with rf._ffi.debug_options(cnn_lds_columns=1):
            r_old, r8_old = rf.ops.cnn_reflectance_u8(imgs, weights=w)
        assert torch.equal(r, r_old) and torch.equal(r8, r8_old)
    ref_r, ref_r8 = co.cnn_reflectance(imgs[0].cpu().numpy(), wts)
    r, r8 = rf.ops.cnn_reflectance_u8(imgs)
    assert np.abs(r[0].cpu().numpy() - ref_r).max() <= 2e-7
