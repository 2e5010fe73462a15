"""Fixtures F5-F7: frozen joint-bilateral / guided-filter / CNN vectors.

tests/golden/filter_vectors.{npz,json} hold seeded inputs, the bytes the C oracle produced when
the fixture was generated, and SHA-256 digests of every result (float planes included).  The CPU
half requires today's oracle to reproduce them (the oracle cannot drift with the kernels); the
GPU half requires the HIP path, through the C ABI, to produce the same bytes.
"""
import hashlib
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")

with open(os.path.join(G, "filter_vectors.json")) as _fh:
    MANIFEST = json.load(_fh)["cases"]
CASES = sorted(MANIFEST)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def vectors():
    return np.load(os.path.join(G, "filter_vectors.npz"))


def _inputs(vectors, name):
    e = MANIFEST[name]
    a = vectors[e["a"]]
    b = vectors[e["b"]] if "b" in e else None
    return e, a, b


def test_fixture_is_complete(vectors):
    kinds = {e["kind"] for e in MANIFEST.values()}
    assert kinds == {"jbf", "gf", "cnn"}
    assert len(CASES) >= 20
    for name in CASES:
        assert name + "/out" in vectors.files, name
        assert _sha(vectors[name + "/out"]) == MANIFEST[name]["sha256"]["out"], name


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_frozen_vectors(vectors, name):
    from oracle import c_oracle as co
    import reflectance_filtering_amd as rf
    e, a, b = _inputs(vectors, name)
    p = e["params"]
    if e["kind"] == "jbf":
        res = {"out": co.joint_bilateral_filter(a, b, p["d"], p["sc"], p["ss"])}
    elif e["kind"] == "gf":
        cur, qf = b, None
        for _ in range(p["iters"]):
            cur, qf = co.guided_filter(a, cur, p["radius"], p["eps"], return_float=True)
        res = {"out": cur, "qf": qf}
    else:
        r, r8 = co.cnn_reflectance(a, rf.weights.load_weights())
        res = {"out": r8, "r": r}
    assert np.array_equal(res["out"], vectors[name + "/out"]), name
    for key, digest in e["sha256"].items():
        assert _sha(res[key]) == digest, (name, key)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_reproduces_frozen_vectors(built, vectors, name):
    import torch
    import reflectance_filtering_amd as rf
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    e, a, b = _inputs(vectors, name)
    p = e["params"]
    want = vectors[name + "/out"]
    if e["kind"] == "jbf":
        got = rf.ximgproc.jointBilateralFilter(a, b, p["d"], p["sc"], p["ss"])
    elif e["kind"] == "gf":
        g = torch.from_numpy(np.ascontiguousarray(a)[None]).cuda()
        s = torch.from_numpy(np.ascontiguousarray(b if b.ndim == 3 else b[:, :, None])[None]).cuda()
        out = rf.ops.guided_filter_u8(g, s, p["radius"], p["eps"], iterations=p["iters"])
        got = out[0].cpu().numpy()
        got = got if b.ndim == 3 else got[:, :, 0]
    else:
        r, r8 = rf.get_reflectance_batch(torch.from_numpy(a[None]).cuda())
        got = r8[0].cpu().numpy()
        # CNN tolerance: 2e-7 on r (fp32 FMA chains, observed equal); the byte may differ only
        # where r*255 sits on an integer boundary
        assert np.abs(r[0].cpu().numpy() - vectors[name + "/r"]).max() <= 2e-7
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
        return
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(got, want), name


# ------------------------------------------------------------------ OpenCV's own bytes, when frozen
# `python tools/t2_report.py --freeze` on a machine with opencv-contrib writes
# tests/golden/opencv_vectors.npz: what cv2.ximgproc returned for the F5-F7 inputs.  From then on the
# oracle (CPU suite) and the HIP path (GPU suite) are held to OpenCV's bytes wherever the suite runs.
# No such machine was available to this build (profiles/r06_t2_probe.txt): until the file exists the
# two tests below skip, and parity with OpenCV stays unpinned.
OPENCV_VECTORS = os.path.join(G, "opencv_vectors.npz")


def _opencv_vectors():
    if not os.path.exists(OPENCV_VECTORS):
        pytest.skip("no OpenCV results frozen (tools/t2_report.py --freeze needs cv2.ximgproc)")
    data = np.load(OPENCV_VECTORS)
    return data, json.loads(str(data["meta"]))


FILTER_CASES = [n for n in CASES if MANIFEST[n]["kind"] in ("jbf", "gf")]


@pytest.mark.parametrize("name", FILTER_CASES)
def test_oracle_reproduces_opencv(vectors, name):
    from oracle import c_oracle as co
    data, meta = _opencv_vectors()
    if name + "/opencv" not in data.files:
        pytest.skip("this OpenCV build refused the case")
    e, a, b = _inputs(vectors, name)
    p = e["params"]
    assert meta["default_is_exact"].get(e["kind"], False), (
        "OpenCV %s is reproduced by the oracle variant(s) %s, not by its default: flip the default "
        "(and the kernels)" % (meta["opencv"], meta["identified"].get(e["kind"])))
    if e["kind"] == "jbf":
        got = co.joint_bilateral_filter(a, b, p["d"], p["sc"], p["ss"])
    else:
        got = b
        for _ in range(p["iters"]):
            got = co.guided_filter(a, got, p["radius"], p["eps"])
    want = data[name + "/opencv"]
    assert np.array_equal(np.asarray(got).reshape(want.shape), want), (name, meta["opencv"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILTER_CASES)
def test_hip_path_reproduces_opencv(built, vectors, name):
    import torch
    import reflectance_filtering_amd as rf
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    data, meta = _opencv_vectors()
    if name + "/opencv" not in data.files:
        pytest.skip("this OpenCV build refused the case")
    e, a, b = _inputs(vectors, name)
    p = e["params"]
    if e["kind"] == "jbf":
        got = rf.ximgproc.jointBilateralFilter(a, b, p["d"], p["sc"], p["ss"])
    else:
        got = b
        for _ in range(p["iters"]):
            got = rf.ximgproc.guidedFilter(a, got, p["radius"], p["eps"])
    want = data[name + "/opencv"]
    assert np.array_equal(np.asarray(got).reshape(want.shape), want), (name, meta["opencv"])
