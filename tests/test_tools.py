"""CPU suite: the one-command reports under tools/ (their logic, not their numbers)."""
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_t2_report_without_opencv(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "cv2", None)          # import cv2 -> ImportError
    out = tmp_path / "t2.json"
    assert _load("t2_report").main(["--out", str(out), "--no-gpu"]) == 3
    doc = json.loads(out.read_text())
    assert doc["opencv"] is None and doc["cases"] == [] and "unpinned" in doc["verdict"]


def test_t2_report_with_a_stand_in_opencv(tmp_path, monkeypatch):
    """A stand-in `cv2` (its ximgproc answers with the oracle, one case off by one grey level)
    drives the whole report path: every F5-F7 filter case compared, the odd case named."""
    from oracle import c_oracle as co
    calls = {"n": 0}

    def jbf(joint, src, d, sc, ss):
        calls["n"] += 1
        res = co.joint_bilateral_filter(joint, src, d, sc, ss)
        if calls["n"] == 1:
            res = res.copy()
            res.flat[0] = res.flat[0] + 1 if res.flat[0] < 255 else 254
        return res

    fake = types.ModuleType("cv2")
    fake.__version__ = "0.0-stand-in"
    fake.error = RuntimeError
    fake.getBuildInformation = lambda: "stand-in build"
    fake.getNumThreads = lambda: 1
    fake.ximgproc = types.SimpleNamespace(
        jointBilateralFilter=jbf,
        guidedFilter=lambda guide, src, radius, eps: co.guided_filter(guide, src, radius, eps))
    monkeypatch.setitem(sys.modules, "cv2", fake)
    out = tmp_path / "t2.json"
    assert _load("t2_report").main(["--out", str(out), "--no-gpu"]) == 0
    doc = json.loads(out.read_text())
    with open(os.path.join(ROOT, "tests", "golden", "filter_vectors.json")) as fh:
        want = sorted(k for k, e in json.load(fh)["cases"].items() if e["kind"] in ("jbf", "gf"))
    assert [c["case"] for c in doc["cases"]] == want
    assert doc["opencv"] == "0.0-stand-in" and doc["hip_path"] is False
    off = [c for c in doc["cases"] if c["oracle_vs_opencv"]["max_abs"]]
    assert len(off) == 1 and off[0]["oracle_vs_opencv"]["max_abs"] == 1
    assert doc["worst_max_abs"] == 1 and "differs" in doc["verdict"]
    assert all(c["frozen_vector_vs_opencv"]["max_abs"] <= 1 for c in doc["cases"])


def test_t2_report_freezes_what_opencv_returned(tmp_path, monkeypatch):
    """`--freeze`: a stand-in `cv2` that answers with the oracle; the file then holds one result per
    filter case, bit for bit what the stand-in returned, the version, the build digest and the verdict
    of the variant search - what tests/test_golden_filters.py compares oracle and HIP path with."""
    from oracle import c_oracle as co
    fake = types.ModuleType("cv2")
    fake.__version__ = "0.0-freeze"
    fake.error = RuntimeError
    fake.getBuildInformation = lambda: "stand-in build"
    fake.getNumThreads = lambda: 1
    fake.ximgproc = types.SimpleNamespace(
        jointBilateralFilter=lambda joint, src, d, sc, ss: co.joint_bilateral_filter(joint, src, d, sc, ss),
        guidedFilter=lambda guide, src, radius, eps: co.guided_filter(guide, src, radius, eps))
    monkeypatch.setitem(sys.modules, "cv2", fake)
    out, npz = tmp_path / "t2.json", tmp_path / "opencv_vectors.npz"
    assert _load("t2_report").main(["--out", str(out), "--no-gpu", "--freeze", str(npz)]) == 0
    data = np.load(npz)
    meta = json.loads(str(data["meta"]))
    assert meta["opencv"] == "0.0-freeze" and meta["default_is_exact"] == {"gf": True, "jbf": True}
    golden = np.load(os.path.join(ROOT, "tests", "golden", "filter_vectors.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "filter_vectors.json")) as fh:
        cases = sorted(k for k, e in json.load(fh)["cases"].items() if e["kind"] in ("jbf", "gf"))
    assert sorted(f[:-len("/opencv")] for f in data.files if f.endswith("/opencv")) == cases
    for name in cases:      # the stand-in IS the oracle, so its bytes are the frozen vectors
        assert np.array_equal(data[name + "/opencv"], golden[name + "/out"]), name
    assert json.loads(out.read_text())["frozen_to"]


def test_t2_report_identifies_a_non_default_variant(tmp_path, monkeypatch):
    """A stand-in `cv2` whose filters make two of the choices the oracle recalled differently (true
    division in the joint bilateral; FMA-contracted helpers and `+ eps` on the diagonal in the
    guided filter): the report's variant search names exactly those switches."""
    from oracle import c_oracle as co

    def jbf(joint, src, d, sc, ss):
        with co.variants("jbf_true_division"):
            return co.joint_bilateral_filter(joint, src, d, sc, ss)

    def gf(guide, src, radius, eps):
        with co.variants("gf_fma", "gf_diag_add_eps"):
            return co.guided_filter(guide, src, radius, eps)

    fake = types.ModuleType("cv2")
    fake.__version__ = "0.0-variant"
    fake.error = RuntimeError
    fake.getBuildInformation = lambda: "stand-in build with other choices"
    fake.getNumThreads = lambda: 1
    fake.ximgproc = types.SimpleNamespace(jointBilateralFilter=jbf, guidedFilter=gf)
    monkeypatch.setitem(sys.modules, "cv2", fake)
    out = tmp_path / "t2.json"
    assert _load("t2_report").main(["--out", str(out), "--no-gpu"]) == 0
    assert co.lib().rfo_get_variants() == 0                 # the search leaves the default behind
    vs = json.loads(out.read_text())["variant_search"]
    assert vs["gf"]["identified"] == [["gf_diag_add_eps", "gf_fma"]]
    assert not vs["gf"]["default_is_exact"] and "flip" in vs["gf"]["reading"]
    # the 8-bit joint bilateral hides most last-ulp choices behind its rounding: the true combination
    # is among the identified ones, the default is not, and every identified one divides truly
    assert ["jbf_true_division"] in vs["jbf"]["identified"]
    assert not vs["jbf"]["default_is_exact"]
    assert all("jbf_true_division" in c for c in vs["jbf"]["identified"])


def test_gf_overlap_table_from_a_kernel_trace(tmp_path, capsys):
    """tools/gf_overlap.py on a hand-made rocprofv3 kernel trace: stage 1 of queue 1 runs 10 us, a row
    walk of queue 2 covers its second half, a column walk of its OWN queue does not count."""
    hdr = ('"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name",'
           '"Correlation_Id","Start_Timestamp","End_Timestamp"\n')
    rows = [(1, "void rf::(anonymous namespace)::gf_stage1_kernel<1, 1, 0>(unsigned char const*)", 1000000, 11000000),
            (2, "void rf::gf_rowstate_kernel<45>(float const*)", 6000000, 13000000),
            (1, "void rf::gf_colwalk_kernel<45, false>(float const*)", 11000000, 15000000),
            (1, "void at::native::vectorized_elementwise_kernel<4>(int)", 0, 500)]
    d = tmp_path / "trace"
    d.mkdir()
    with open(d / "t_kernel_trace.csv", "w") as fh:
        fh.write(hdr)
        for k, (q, name, t0, t1) in enumerate(rows):
            fh.write('"KERNEL_DISPATCH","Agent 2",%d,%d,1,%d,1,"%s",%d,%d,%d\n' % (q, q - 1, k, name, k, t0, t1))
    mod = _load("gf_overlap")
    old = sys.argv
    sys.argv = ["gf_overlap.py", str(d), "--label", "test", "--out", str(tmp_path / "o.md")]
    try:
        mod.main()
    finally:
        sys.argv = old
    txt = capsys.readouterr().out
    assert "| stage1 | 1 | 10.00 | 10.00 |" in txt
    assert "| stage 1 and a walk kernel both running | 5.00 |" in txt
    assert "| 0 | 1/0 | 0.00 | 10.00 | 50 % |" in txt
    assert "**50 %**" in txt and (tmp_path / "o.md").read_text() == txt + "\n"
