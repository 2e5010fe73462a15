"""GPU suite: the multi-rank paths as the driver launches them, on however many GPUs the box has.

`bench.py --gpus 2` (ranks started by bench.py itself; two ranks share the device over gloo when
there is only one) and the batch front-end under a 2-rank environment (contiguous file shards, no
collective) against the single-rank run, byte for byte.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def _clean_env(**extra):
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RF_BENCH_STUB"):
        env.pop(key, None)
    env.update(extra)
    return env


def test_bench_two_ranks(gpu):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps",
           "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extras"]
    p = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["batch_per_gpu"] == 8
    job_px = 2 * 8 * 1080 * 1920 * 2                     # ranks x batch x pixels x steps
    assert np.isclose(out["value"] * 1e6 * out["ms_per_step"] * 1e-3 * 2, job_px, rtol=1e-6)
    assert out["roofline"]["kernel_ms"] <= out["ms_per_step"] * 1.02


def test_batch_front_end_two_ranks_equal_one(gpu, tmp_path):
    from tests import synth
    from reflectance_filtering_amd import image_utils as iu
    src = tmp_path / "in"
    src.mkdir()
    names = []
    for i, (h, w) in enumerate([(60, 80), (60, 80), (48, 64), (60, 80), (48, 64)]):
        f = str(src / ("img%d.png" % i))
        iu.imwrite(f, synth.scene_u8(h, w, seed=40 + i))
        names.append(f)

    def run(tag, world):
        out = tmp_path / tag
        out.mkdir()
        procs = []
        for rank in range(world):
            env = _clean_env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
            procs.append(subprocess.Popen(
                [sys.executable, "-m", "reflectance_filtering_amd.batch", "decompose", "--inputs",
                 str(src / "*.png"), "--path_out", str(out)], cwd=ROOT, env=env,
                stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        logs = []
        for p in procs:
            o, e = p.communicate(timeout=600)
            assert p.returncode == 0, e.decode()[-3000:]
            logs.append(o.decode())
        return out, logs

    one, _ = run("one", 1)
    two, logs = run("two", 2)
    assert "wrote 3 file(s)" in logs[0] and "wrote 2 file(s)" in logs[1]   # 5 files: 3 + 2
    files = sorted(os.listdir(str(one)))
    assert files == sorted(os.listdir(str(two))) and len(files) == 15      # -r, -r_colorized, -s_colorized
    for f in files:
        with open(str(one / f), "rb") as a, open(str(two / f), "rb") as b:
            assert a.read() == b.read(), f


def test_sharding_helpers_over_rccl_on_one_device(gpu):
    """The `nccl` (= RCCL) branch of sharding.barrier / reduce_job - device-side barrier with
    device_ids, device tensors in the two scalar all-reduces - run through a real RCCL
    communicator.  A 1-GPU box admits a group of one rank only (RCCL refuses two ranks on one
    device), which still drives every call of that branch; on a box with >= 2 devices the same
    script runs two ranks, one per device."""
    script = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from reflectance_filtering_amd import sharding
rank, world, local = sharding.init_distributed(backend="nccl") if int(os.environ["WORLD_SIZE"]) > 1 else (0, 1, 0)
torch.cuda.set_device(local)
if world == 1:
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
sharding.barrier(2)                     # world_size argument only guards the call
units, secs = sharding.reduce_job(100.0 + rank, 0.5 + rank, 2, device=dev)
n = dist.get_world_size()
assert units == sum(100.0 + r for r in range(n)), units
assert secs == 0.5 + (n - 1), secs
sharding.barrier(2)
dist.destroy_process_group()
print("ok", n)
''' % ROOT
    world = 2 if gpu.cuda.device_count() >= 2 else 1
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(world):
        env = _clean_env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                         MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                         HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e.decode()[-3000:]
        assert "ok %d" % world in o.decode().splitlines()      # RCCL prints its banner to stdout too


def test_bench_measures_its_traffic_live(gpu):
    """`bench.py --traffic live`: the two rocprofv3 --pmc child passes run, the kernel is found in
    their counter files and `roofline.traffic` is of the order of the algorithmic bytes (at this
    small batch the counters read about 3x - 1.07x at the metric's 256 images, where the 128-byte
    requests FETCH_SIZE is corrected for dominate)."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this machine")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "1", "--warmup",
           "1", "--cpu-seconds", "0", "--no-extras", "--traffic", "live"]
    p = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    roof = out["roofline"]
    assert roof["traffic_source"].startswith("measured by this run"), (roof, p.stderr.decode()[-2000:])
    assert "jbf" in roof["kernel"]
    assert 0.8 < roof["traffic"] / roof["algorithmic_bytes_per_launch"] < 5.0
