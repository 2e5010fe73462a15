"""CPU suite, part 4: multi-GPU host logic (contiguous shards, no data-path collective),
exercised with world_size-2 gloo processes."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from reflectance_filtering_amd.sharding import shard_range, shard_sizes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 255, 4096):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
            assert sizes == sorted(sizes, reverse=True)
    assert shard_sizes(4096, 8) == [512] * 8      # BASELINE config C4
    assert shard_sizes(1024, 8) == [128] * 8      # BASELINE config C5
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


WORKER = textwrap.dedent("""
    import json, sys, time
    sys.path.insert(0, %(root)r)
    import numpy as np
    from reflectance_filtering_amd import sharding
    rank, world, local = sharding.init_distributed(backend="gloo")
    lo, hi = sharding.shard_range(37, world, rank)
    # each rank "processes" its own slice with no exchange; only (units, seconds) are reduced
    sharding.barrier(world)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))
    local_s = time.perf_counter() - t0
    units, secs = sharding.reduce_job(hi - lo, local_s, world)
    sharding.barrier(world)
    print(json.dumps({"rank": rank, "world": world, "lo": lo, "hi": hi, "units": units,
                      "secs": secs, "local_s": local_s}))
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_job(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    import json
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err.decode()[-2000:]
        outs.append(json.loads(out.decode().strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert (outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]) == (0, 19, 19, 37)
    for o in outs:
        assert o["units"] == 37.0                         # sum over ranks
        assert o["secs"] >= max(x["local_s"] for x in outs) - 1e-9   # max over ranks
    assert outs[0]["secs"] == outs[1]["secs"]
