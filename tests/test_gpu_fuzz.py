"""GPU suite: seeded random cases against the ORACLE (not against another kernel), sized so that
the C oracle answers in milliseconds - every radius of the fused guided filter (1..96) and beyond,
every border type, channel combination and flag of the joint bilateral, chained passes, batches
that mix grey and colour images, the CNN with the shipped and with random weights.  Bounded by
time: `RF_FUZZ_SECONDS` (default 40) per filter test; `RF_FUZZ_SEED` (default 0) offsets the seeds
for longer runs on other cases.
"""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SECONDS = float(os.environ.get("RF_FUZZ_SECONDS", "40"))
SEED = int(os.environ.get("RF_FUZZ_SEED", "0"))


@pytest.fixture(scope="module")
def env(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device visible")
    import reflectance_filtering_amd as rf
    from oracle import c_oracle as co
    torch.cuda.set_device(0)
    return rf, co, torch


def _image(rng, h, w, c, kind):
    """uint8 [h,w,c]: smooth field, posterised field, white noise or a constant."""
    if kind == 0:
        yy, xx = np.mgrid[0:h, 0:w]
        base = sum(np.sin(xx * rng.uniform(0.02, 0.4) + yy * rng.uniform(0.02, 0.4) + rng.uniform(0, 6))
                   for _ in range(3))
        img = 128 + 40 * base[..., None] + rng.normal(0, 6, (h, w, c))
    elif kind == 1:
        img = rng.integers(0, 5, (h // 7 + 1, w // 9 + 1, c)).repeat(7, 0).repeat(9, 1)[:h, :w] * 60 + 7
    elif kind == 2:
        img = rng.integers(0, 256, (h, w, c))
    else:
        img = np.full((h, w, c), int(rng.integers(0, 256)))
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


def test_guided_filter_random_cases_match_the_oracle(env):
    rf, co, torch = env
    rng = np.random.default_rng(2024 + SEED)
    t_end = time.time() + SECONDS
    cases = exact_cases = 0
    while time.time() < t_end or cases < 12:
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 220))
        radius = int(rng.integers(1, 101)) if rng.random() < 0.8 else int(rng.choice([45, 52, 104, 120, 128, 129]))
        # (round 6) a sixth of the cases through the exact-row stage 2: its radii, a width that is a
        # multiple of 16 - white-noise guides and tiny eps make rows fail the test, constants pass it
        exact = rng.random() < 1.0 / 6.0
        if exact:
            radius, w = int(rng.choice([45, 52])), 16 * int(rng.integers(1, 14))
        eps = float(rng.choice([3.0, 7.0, 0.5, 1e-3, 200.0]))
        iters = int(rng.choice([1, 1, 2, 3]))
        n = int(rng.integers(1, 4))
        scn = int(rng.choice([1, 3]))
        guides = [_image(rng, h, w, 3, int(rng.integers(0, 4))) for _ in range(n)]
        srcs = []
        for _ in range(n):
            s = _image(rng, h, w, scn, int(rng.integers(0, 3)))
            if scn == 3 and rng.random() < 0.4:          # a grey image among colour ones
                s = np.repeat(s[:, :, :1], 3, axis=2)
            srcs.append(s)
        with rf._ffi.debug_options(gf_exact=int(exact)):
            got = rf.ops.guided_filter_u8(torch.from_numpy(np.stack(guides)).cuda(),
                                          torch.from_numpy(np.stack(srcs)).cuda(), radius, eps,
                                          iterations=iters).cpu().numpy()
        exact_cases += int(exact)
        for i in range(n):
            cur = srcs[i]
            for _ in range(iters):
                cur = co.guided_filter(guides[i], cur, radius, eps).reshape(srcs[i].shape)
            assert np.array_equal(got[i], cur), (cases, h, w, radius, eps, iters, scn, i)
        cases += 1
    print("guided-filter fuzz: %d cases (%d through the exact-row stage 2)" % (cases, exact_cases))


def test_joint_bilateral_random_cases_match_the_oracle(env):
    rf, co, torch = env
    rng = np.random.default_rng(4048 + SEED)
    t_end = time.time() + SECONDS
    cases = 0
    while time.time() < t_end or cases < 12:
        h, w = int(rng.integers(1, 110)), int(rng.integers(1, 150))
        jcn, scn = int(rng.choice([1, 3])), int(rng.choice([1, 3]))
        joint = _image(rng, h, w, jcn, int(rng.integers(0, 4)))
        src = _image(rng, h, w, scn, int(rng.integers(0, 3)))
        # radius 33 42 8 18 51 2 38 46 | 54 60 64 70 76 99 130 140 (tap-row slabs)
        ss = float(rng.choice([22.0, 28.0, 5.0, 12.3, 34.0, 1.0, 25.0, 31.0, 36.0, 40.0, 42.9, 47.0, 50.5,
                               66.0, 86.7, 93.3]))
        sc = float(rng.choice([20.0, 15.0, 4.0, 60.0, 0.5]))
        d = int(rng.choice([-1, -1, 5, 9, 31]))
        border = int(rng.choice([0, 1, 2, 3, 4]))
        got = rf.ops.joint_bilateral_u8(torch.from_numpy(joint[None]).cuda(),
                                        torch.from_numpy(src[None]).cuda(), d, sc, ss,
                                        border=border).cpu().numpy()[0]
        want = co.joint_bilateral_filter(joint, src, d, sc, ss, border=border).reshape(src.shape)
        assert np.array_equal(got, want), (cases, h, w, jcn, scn, sc, ss, d, border)
        cases += 1
    print("joint-bilateral fuzz: %d cases" % cases)


def test_cnn_random_cases_match_the_oracle(env):
    """Random image sizes (odd pixel counts: the kernel pairs pixel i with pixel i + half), image
    statistics and - every other case - random weights of the magnitude of the shipped ones:
    float output within 2e-7 of the oracle (the contract of test_cnn_matches_oracle_and_golden),
    bytes within one count on < 1e-4 of the pixels."""
    rf, co, torch = env
    rng = np.random.default_rng(777 + SEED)
    shipped = rf.weights.load_weights()
    t_end = time.time() + SECONDS / 4
    cases = 0
    worst = 0.0
    while time.time() < t_end or cases < 6:
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 130))
        n = int(rng.integers(1, 4))
        imgs = np.stack([_image(rng, h, w, 3, int(rng.integers(0, 4))) for _ in range(n)])
        if cases % 2:
            wts = (shipped * rng.uniform(0.5, 1.5, shipped.shape)
                   + rng.normal(0, 0.02, shipped.shape)).astype(np.float32)
        else:
            wts = shipped
        r, r8 = rf.ops.cnn_reflectance_u8(torch.from_numpy(imgs).cuda(), weights=wts)
        r, r8 = r.cpu().numpy(), r8.cpu().numpy()
        for i in range(n):
            want_r, want_r8 = co.cnn_reflectance(imgs[i], wts)
            err = float(np.abs(r[i] - want_r).max())
            worst = max(worst, err)
            assert err <= 2e-7, (cases, h, w, i, err)
            d8 = np.abs(r8[i].astype(int) - want_r8.astype(int))
            assert d8.max() <= 1 and np.mean(d8 != 0) < 1e-4 + 1.0 / d8.size, (cases, h, w, i)
        cases += 1
    print("CNN fuzz: %d cases, largest |r - oracle| %.3g" % (cases, worst))
