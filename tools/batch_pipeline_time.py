#!/usr/bin/env python
"""End-to-end file -> file time of the batch front-end (filter, bilateral, grey predictions with
colour guidance) with and without the decode / device / encode pipeline.

    python tools/batch_pipeline_time.py [n_files] [h w]
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import reflectance_filtering_amd as rf
from reflectance_filtering_amd import batch, image_utils as iu
from tests import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
d = tempfile.mkdtemp()
for i in range(n):
    iu.imwrite(os.path.join(d, "p%03d.png" % i), synth.scene_u8(h, w, seed=i))
    g = synth.reflectance_like_u8(h, w, seed=100 + i)
    iu.imwrite(os.path.join(d, "p%03d-r.png" % i), np.repeat(g[..., :1], 3, axis=2))
files = batch.expand_inputs([os.path.join(d, "*-r.png")])
out = os.path.join(d, "out")
os.makedirs(out)
batch.filter_files("bilateral", files[:4], os.path.join(d, "{base}.png"), 20.0, 22.0, out)  # warm-up
for label, step in (("one step (no overlap)", 10 ** 9), ("pipeline, 16 files per step", 16),
                    ("pipeline, 64 files per step", 64)):
    batch.STEP_FILES = step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.filter_files("bilateral", files, os.path.join(d, "{base}.png"), 20.0, 22.0, out)
    dt = time.perf_counter() - t0
    print("%-32s %6.2f s  %.1f MP/s file to file (%d x %dx%d, %d I/O threads)"
          % (label, dt, n * h * w / 1e6 / dt, n, w, h, batch.IO_THREADS))
