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
    assert per_wave == 3664  # rows start at an even column (3,764 when they started at a multiple of 4)
    assert np.isclose(obj["taps_per_s"], taps / 0.265)


def test_lds_cobound_object_and_measurement_budget(monkeypatch):
    """Round 5: the headline line says what bounds it - `lds` beside `valu` (the tap loop's LDS
    pipeline priced per column step, plus the committed SQ pass's busy share) - and the rocprofv3
    child passes of one run share one wall-clock budget."""
    steps = bench.valu_roofline(256, 1080, 1920, 33, 265.0, 1.0)["column_steps_per_launch"]
    obj = bench.lds_cobound(steps, 257.0, 2319.0)
    assert obj["bound"] == "lds-issue" and obj["wave_instructions_per_column_step"] == 4.5
    valu = bench.valu_roofline(256, 1080, 1920, 33, 257.0, 1.0, 2319.0)
    # the LDS floor (4 gathers + half a texel-pair read per step) is 0.7-1.0 of the VALU floor
    assert 0.7 < obj["floor_ms"] / valu["floor_ms"] < 1.0 and 0.4 < obj["frac"] < 1.0
    assert obj["busy_measured"] is None or 0.3 < obj["busy_measured"] < 1.0
    assert bench.HBM_COPY_CEILING_GBS == 6290.0
    # budget: a pass may start while the previous one took at most a third of what is left
    monkeypatch.setattr(bench, "_child_clock", {"t_end": None, "last": 0.0})
    monkeypatch.setattr(bench, "CHILD_BUDGET_S", 60.0)
    ok, left = bench.child_pass_allowed()
    assert ok and 59.0 < left <= 60.0
    bench.child_pass_done(10.0)
    assert bench.child_pass_allowed()[0]
    bench.child_pass_done(25.0)
    assert not bench.child_pass_allowed()[0]


def _run_bench(extra_args, env_extra, timeout=300):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RF_BENCH_STUB="1", **env_extra)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        if key not in env_extra:
            env.pop(key, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()


def test_bench_launches_its_own_ranks_and_sums_their_pixels():
    """`python bench.py --gpus 2` starts two ranks itself (gloo here, kernel call stubbed): one
    JSON line from rank 0 with n_gpus = 2 and the pixels of both ranks in `value`."""
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                               "--height", "8", "--width", "16"], {})
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["batch_per_gpu"] == 4
    job_px = 2 * 4 * 8 * 16 * 3                       # ranks x batch x h x w x steps
    assert np.isclose(out["value"] * 1e6 * out["ms_per_step"] * 1e-3 * 3, job_px, rtol=1e-6)
    assert out["ms_per_step"] >= 10.0                 # the stub step sleeps 10 ms
    # (round 6) the N > 1 line is self-contained: every rank's own time (a slow GPU must not hide
    # behind the max), how many ranks reported, and the CPU baseline of the same run (rank 0, after
    # the final barrier)
    assert out["ranks_seen"] == 2 and len(out["per_rank_ms"]) == 2
    assert all(ms >= 10.0 for ms in out["per_rank_ms"])
    assert max(out["per_rank_ms"]) <= out["ms_per_step"] * 1.5
    base = out["cpu_baseline"]
    assert base["kind"] in ("port", "reference") and base["value"] > 0 and base["cores"] >= 1
    assert base["unit"] == "MP/s" and "sample" in base


def test_bench_has_the_colour_chain_configuration():
    """The reference's published 3x guided chain filters a colour reflectance
    (/root/reference/README.md:66): `c5_gf_colour` of the N = 1 line is this configuration."""
    assert bench.CONFIGS["c5c"] == ("gf3c", 64, 2160, 3840)
    t, why = bench.committed_config_traffic("no_such_line", 1)
    assert t is None and why


def test_bench_refuses_a_world_size_mismatch():
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"],
                              {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and out is None
    assert "WORLD_SIZE" in err


def test_bench_configs_name_the_baseline_shapes():
    assert bench.CONFIGS["c4"] == ("jbf", 512, 1080, 1920)       # 4096 images over 8 GPUs
    assert bench.CONFIGS["c5"] == ("gf3", 128, 2160, 3840)       # 1024 images over 8 GPUs
    assert bench.CONFIGS["c3"] == ("chain", 256, 333, 500)
    assert bench.CONFIGS["north_star"] == ("jbf", 256, 1080, 1920)


def test_bench_stops_all_ranks_when_one_dies():
    """Rank 1 exits before the rendezvous: bench.py ends the surviving rank and returns the
    failure instead of sitting in the barrier until the process-group timeout."""
    import time
    t0 = time.monotonic()
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
                               "--height", "8", "--width", "8"], {"RF_BENCH_STUB_FAIL_RANK": "1"},
                              timeout=120)
    assert rc == 3 and out is None
    assert "rank 1 exited with 3" in err
    assert time.monotonic() - t0 < 60


def test_scale_report_tabulates_the_curve(tmp_path, monkeypatch):
    """tools/scale_report.py: one command for the 1 -> N curve (here 1 and 2 stub ranks on gloo):
    absolute MP/s, ratio to one GPU, skipped counts named."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("scale_report",
                                                  os.path.join(root, "tools", "scale_report.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setenv("RF_BENCH_STUB", "1")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        monkeypatch.delenv(key, raising=False)
    out = tmp_path / "scale.json"
    doc = mod.main(["--gpus", "1,2", "--steps", "2", "--warmup", "0", "--batch", "2",
                    "--out", str(out)])
    assert [c["n_gpus"] for c in doc["curve"]] == [1, 2] and doc["skipped"] == []
    assert doc["curve"][0]["ratio_to_one_gpu"] == 1.0
    assert 1.5 < doc["curve"][1]["ratio_to_one_gpu"] < 2.2       # the stub step is a fixed sleep
    assert out.exists()


def test_live_traffic_helpers(tmp_path, monkeypatch):
    """The pieces of bench.py's live PMC measurement that run without a GPU: the counter CSV reader
    (the timed launch is the LAST dispatch of the kernel; other kernels and counters are ignored)
    and the guard that keeps a profiled run from starting profiler passes of its own."""
    csv_path = tmp_path / "1_counter_collection.csv"
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id",'
            '"Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size",'
            '"VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
            '"Start_Timestamp","End_Timestamp"\n')
    row = '%d,%d,"Agent 2",1,9,9,64,3,"%s",256,0,0,4,0,16,"%s",%f,1,2\n'
    body = (row % (1, 1, "void at::native::fill(float)", "FETCH_SIZE", 15.5)
            + row % (2, 2, "void rf::(anonymous namespace)::jbf_tile64_kernel<1, 8>(...)", "FETCH_SIZE", 100.0)
            + row % (3, 3, "void rf::(anonymous namespace)::jbf_tile64_kernel<1, 8>(...)", "FETCH_SIZE", 200.0)
            + row % (3, 3, "void rf::(anonymous namespace)::jbf_tile64_kernel<1, 8>(...)", "WRITE_SIZE", 7.0))
    csv_path.write_text(head + body)
    assert bench.parse_counter_csv(str(csv_path), "FETCH_SIZE") == [100.0, 200.0]
    assert bench.parse_counter_csv(str(csv_path), "WRITE_SIZE") == [7.0]
    assert bench.parse_counter_csv(str(csv_path), "FETCH_SIZE", match="nothing") == []
    for key in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH",
                "LD_PRELOAD"):
        monkeypatch.delenv(key, raising=False)
    assert not bench.being_profiled()
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.being_profiled()
    assert bench.parse_args([]).traffic == "auto"


def test_flat_guide_is_seeded_voronoi_cells():
    """C5's guidance: seeded Voronoi cells of flat colour +-1 dither, 200 - 2,000
    regions by image size, deterministic, generated with torch on whatever device the scene is on."""
    import numpy as np
    import torch
    bench = _load_bench() if "_load_bench" in globals() else __import__("bench")
    scene, _ = bench.synth_batch(torch, 2, 540, 960, 77, torch.device("cpu"))
    g = bench.flat_guide(scene)
    assert g.dtype == torch.uint8 and g.shape == scene.shape
    assert torch.equal(g, bench.flat_guide(scene))                     # seeded
    assert not torch.equal(g[0], g[1])
    a = g[0].numpy().astype(np.int16)
    flat = np.abs(a[:, 1:] - a[:, :-1]).max(axis=2) <= 2             # within the dither of the left neighbour
    assert flat.mean() > 0.95
    # count regions: connected runs of near-equal colour along rows are long (cells ~ 90 px wide)
    runs = (~flat).sum(axis=1).mean()                                  # cell borders crossed per row
    assert 2 <= runs <= 40
    coarse = len(np.unique((a // 8).reshape(-1, 3), axis=0))
    assert 50 <= coarse <= 2000 * 8                                    # hundreds of cells, not a posterised scene
