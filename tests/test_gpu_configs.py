"""GPU suite: the BASELINE.json configurations at their stated shapes.

  C3  batch 256 of IIW-size (333x500) images through CNN -> trunc(r*255) -> BF(CNN,CNN)
  C4  one rank's shard of "4096 x 1920x1080 over 8 GPUs": a 512-image joint-bilateral launch
  C5  3840x2160, piecewise-constant guide, 3x guided filter c=3.0 s=45.0 (radius 45, eps 3)

The guided filter is not local (its running sums start at the image border), so C5 is checked
against a full-size oracle run; the joint bilateral is local, so the big launches are checked
through batch independence plus oracle crops (crop + halo through the oracle == crop of the
full result).  Inputs are generated on the device with bench.py's seeded generators.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import torch
    import bench
    import reflectance_filtering_amd as rf
    from oracle import c_oracle as co
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    rf._ffi.load_library()
    return rf, co, torch, bench


@pytest.mark.parametrize("src_kind", ["grey", "colour"])
def test_c5_4k_three_guided_passes_against_oracle(env, src_kind):
    """/root/reference/filter_reflectance.py:67-70 applied three times (README.md:66 of the
    reference: `..._guided_c3.0s45.0` chained), one full 3840x2160 image, flat guide."""
    rf, co, torch, bench = env
    h, w = 2160, 3840
    dev = torch.device("cuda", 0)
    scene, grey = bench.synth_batch(torch, 1, h, w, 7000, dev)
    guide = bench.flat_guide(scene)
    src = grey if src_kind == "grey" else scene
    got = rf.ops.guided_filter_u8(guide, src, 45, 3.0, iterations=3)
    with rf._ffi.debug_options(gf_two_kernel=1):
        assert torch.equal(rf.ops.guided_filter_u8(guide, src, 45, 3.0, iterations=3), got)
    g_np = guide[0].cpu().numpy()
    cur = src[0].cpu().numpy()
    # the guide really is piecewise constant (seeded Voronoi cells of one colour +-1 dither): at most
    # 2,000 regions x 27 dither offsets, and nearly every pixel within the dither of its left neighbour
    assert len(np.unique(g_np.reshape(-1, 3), axis=0)) <= 2000 * 27
    step = np.abs(g_np[:, 1:].astype(np.int16) - g_np[:, :-1].astype(np.int16)).max(axis=2)
    assert (step <= 2).mean() > 0.97
    for _ in range(3):
        cur = co.guided_filter(g_np, cur, 45, 3.0)
    assert np.array_equal(got[0].cpu().numpy(), cur)


def test_c5_shard_batch_is_independent_of_its_neighbours(env):
    """Several 4K images in one call (grey and colour sources mixed, more images than fit the
    workspace at once) give the bytes of the one-image calls."""
    rf, co, torch, bench = env
    dev = torch.device("cuda", 0)
    scene, grey = bench.synth_batch(torch, 3, 2160, 3840, 7100, dev)
    guide = bench.flat_guide(scene)
    src = scene.clone()
    src[1] = grey[1]
    lib = rf._ffi.load_library()
    ws = torch.empty(lib.rf_gf_workspace_bytes(2, 2160, 3840, 3, 3, 45), dtype=torch.uint8,
                     device=dev)
    got = rf.ops.guided_filter_u8(guide, src, 45, 3.0, iterations=3, workspace=ws)
    for i in range(3):
        one = rf.ops.guided_filter_u8(guide[i:i + 1].contiguous(), src[i:i + 1].contiguous(), 45,
                                      3.0, iterations=3)
        assert torch.equal(one[0], got[i]), i


def test_c3_batch_256_chain(env):
    """256 IIW-size images through decompose_and_filter_batch: three of them against the oracle's
    CNN + BF(CNN,CNN) (the two-CLI chain), all of them against the same call on sub-batches."""
    rf, co, torch, bench = env
    dev = torch.device("cuda", 0)
    scene, _ = bench.synth_batch(torch, 256, 333, 500, 3000, dev)
    r8, filt = rf.decompose_and_filter_batch(scene)
    assert r8.shape == (256, 333, 500) and filt.shape == (256, 333, 500)
    wts = rf.weights.load_weights()
    for i in (0, 100, 255):
        img = scene[i].cpu().numpy()
        _, want_r8 = co.cnn_reflectance(img, wts)
        got_r8 = r8[i].cpu().numpy()
        # r = sigmoid(z) within 2e-7 of the oracle (observed equal); the byte trunc(r*255) may
        # differ only on an exact integer boundary
        assert np.abs(got_r8.astype(int) - want_r8.astype(int)).max() <= 1
        r3 = np.repeat(got_r8[:, :, None], 3, 2)
        want = co.joint_bilateral_filter(r3, r3.copy(), -1, 20, 22)[:, :, 0]
        assert np.array_equal(filt[i].cpu().numpy(), want), i
    for lo, hi in ((0, 7), (100, 101), (249, 256)):
        r8b, fb = rf.decompose_and_filter_batch(scene[lo:hi].contiguous())
        assert torch.equal(r8b, r8[lo:hi]) and torch.equal(fb, filt[lo:hi])


def test_c4_512_image_launch(env):
    """One rank's C4 shard: 512 x 1920x1080 in ONE rf_jbf_u8 launch.  Eight distinct images repeated
    64 times must give eight distinct results repeated 64 times (batch independence at the full
    launch size), and crops of the first and last image must equal the oracle on crop + halo."""
    rf, co, torch, bench = env
    dev = torch.device("cuda", 0)
    h, w, r = 1080, 1920, 33
    j8, s8 = bench.synth_batch(torch, 8, h, w, 4000, dev)
    joint = j8.repeat(64, 1, 1, 1)
    src = s8.repeat(64, 1, 1, 1)
    assert joint.shape == (512, h, w, 3)
    out = rf.ops.joint_bilateral_u8(joint, src, -1, 20.0, 22.0)
    first = out[:8]
    for k in range(1, 64):
        assert torch.equal(out[8 * k:8 * k + 8], first), k
    for img, (y0, x0) in ((0, (0, 0)), (511, (h - 70, w - 90)), (259, (500, 1000))):
        ya, yb = max(0, y0 - r), min(h, y0 + 70 + r)
        xa, xb = max(0, x0 - r), min(w, x0 + 90 + r)
        # the crop keeps the image border where it touches it, so REFLECT_101 agrees there
        jc = joint[img, ya:yb, xa:xb].cpu().numpy()
        sc = src[img, ya:yb, xa:xb].cpu().numpy()
        want = co.joint_bilateral_filter(np.ascontiguousarray(jc), np.ascontiguousarray(sc), -1,
                                         20.0, 22.0)
        got = out[img, y0:y0 + 70, x0:x0 + 90].cpu().numpy()
        assert np.array_equal(got, want[y0 - ya:y0 - ya + 70, x0 - xa:x0 - xa + 90]), img


def test_c5_full_shard_128_images(env):
    """One rank's C5 shard at its stated size: 128 x 3840x2160 through three guided passes in ONE
    call (the workspace holds a dozen images, so the call runs in chunks).  Four distinct images
    repeated 32 times must give four distinct results repeated 32 times, equal to the one-image
    calls (which test_c5_4k_three_guided_passes_against_oracle ties to the oracle)."""
    rf, co, torch, bench = env
    dev = torch.device("cuda", 0)
    scene4, grey4 = bench.synth_batch(torch, 4, 2160, 3840, 7200, dev)
    guide4 = bench.flat_guide(scene4)
    src4 = grey4.clone()
    src4[3] = scene4[3]                       # one colour source among the grey ones
    guide = guide4.repeat(32, 1, 1, 1)
    src = src4.repeat(32, 1, 1, 1)
    assert guide.shape == (128, 2160, 3840, 3)
    out = rf.ops.guided_filter_u8(guide, src, 45, 3.0, iterations=3)
    for i in range(4):
        one = rf.ops.guided_filter_u8(guide4[i:i + 1].contiguous(), src4[i:i + 1].contiguous(), 45,
                                      3.0, iterations=3)
        for k in range(32):
            assert torch.equal(out[4 * k + i], one[0]), (i, k)
    del out, guide, src
    rf.ops.release_workspaces()
    torch.cuda.empty_cache()
