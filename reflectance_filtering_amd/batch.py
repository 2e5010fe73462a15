#!/usr/bin/env python
"""Batch front-end: the reference's two tools over many files, one process per GPU.

The reference handles one image per invocation (/root/reference/filter_reflectance.py:76-96,
/root/reference/decompose_with_trained_CNN.py:98-130).  Every image is independent, so a list of
files shards embarrassingly: each rank (torchrun sets RANK/WORLD_SIZE/LOCAL_RANK) takes a
contiguous slice of the sorted file list and works through it in steps of 16 files as a pipeline
- a thread pool decodes the next step and encodes the previous one while the device filters the
current one - grouping the images of a step by size, pushing each group through the
device-resident batch operators and writing the same output files the single-image tools would
write.  No collective is involved.

    python -m reflectance_filtering_amd.batch filter --filter_type=bilateral --sigma_color=20 \
        --sigma_spatial=22 --inputs 'out/*-r.png' --guidance 'photos/{stem}.png' --path_out out
    python -m torch.distributed.run --nproc-per-node 8 -m reflectance_filtering_amd.batch \
        decompose --inputs 'photos/*.png' --path_out out

`--guidance` is a pattern evaluated per input: {path} {dir} {name} {stem} {ext}; `{stem}` of
`x-r.png` is `x-r`, `{base}` strips a trailing `-r` as well, so `photos/{base}.png` pairs a CNN
prediction with its photo.  Omit it to use each input as its own guidance (BF(CNN,CNN)).
"""
from __future__ import division, print_function

import argparse
import glob
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import filter_reflectance as fr
from . import ops
from . import image_utils as iu
from . import sharding

MAX_BATCH_BYTES = 2 << 30  # per group of equal-size images kept on the device at once
STEP_FILES = 16            # files per pipeline step (decode k+1 | device k | encode k-1)
IO_THREADS = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity")
                        else (os.cpu_count() or 1)))


def pipeline(items, load, compute, step=None):
    """Three-stage pipeline over `items` in steps of `step` files: while the device works on step
    k (in the calling thread: `compute(list of load() results)` returns a list of
    (file name, array) pairs to write), the thread pool decodes step k+1 and encodes what step k-1
    produced.  Image decode / encode spend their time in zlib with the GIL released and the device
    calls release it too, so the three stages overlap; at most two decoded steps and two steps of
    pending writes are alive at a time.  Returns the names written, in order."""
    step = step or STEP_FILES
    items = list(items)
    steps = [items[i:i + step] for i in range(0, len(items), step)]
    written = []
    if not steps:
        return written
    with ThreadPoolExecutor(max_workers=IO_THREADS) as pool:
        loads = [pool.submit(load, it) for it in steps[0]]
        pending = []                       # write futures of the previous steps, oldest first
        for k, _ in enumerate(steps):
            nxt = [pool.submit(load, it) for it in steps[k + 1]] if k + 1 < len(steps) else []
            loaded = [f.result() for f in loads]
            jobs = compute(loaded)
            while len(pending) > 1:        # writes of step k-2 must be done before step k's start
                for f in pending.pop(0):
                    f.result()
            pending.append([pool.submit(iu.imwrite, name, arr) for name, arr in jobs])
            written.extend(name for name, _ in jobs)
            loads = nxt
        for futs in pending:
            for f in futs:
                f.result()
    return written


def expand_inputs(patterns):
    """Sorted, de-duplicated list of files matched by the patterns (a literal path matches itself)."""
    files = []
    for pat in patterns:
        hits = sorted(glob.glob(pat))
        files.extend(hits if hits else ([pat] if os.path.exists(pat) else []))
    seen, out = set(), []
    for f in files:
        if f not in seen:
            seen.add(f)
            out.append(f)
    return out


def guidance_for(path, pattern):
    """File name of the guidance image of `path` under `pattern` (None -> the input itself)."""
    if not pattern:
        return path
    d, name = os.path.split(path)
    stem, ext = os.path.splitext(name)
    base = stem[:-2] if stem.endswith("-r") else stem
    return pattern.format(path=path, dir=d, name=name, stem=stem, ext=ext, base=base)


def my_slice(files, rank=None, world=None):
    """The contiguous part of `files` this rank owns."""
    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    if world is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    lo, hi = sharding.shard_range(len(files), world, rank)
    return files[lo:hi]


def group_by_shape(items, shape_of, max_bytes=MAX_BATCH_BYTES):
    """Consecutive runs of `items` with equal shape, cut so that a run stays under max_bytes."""
    groups, cur, cur_shape = [], [], None
    for it in items:
        shp = shape_of(it)
        per = int(np.prod(shp))
        if cur and (shp != cur_shape or (len(cur) + 1) * per > max_bytes):
            groups.append(cur)
            cur = []
        cur.append(it)
        cur_shape = shp
    if cur:
        groups.append(cur)
    return groups


def _is_grey(batch):
    """All images of a [N,H,W,3] uint8 batch have three equal channels."""
    return (batch.shape[-1] == 3 and np.array_equal(batch[..., 0], batch[..., 1])
            and np.array_equal(batch[..., 1], batch[..., 2]))


def filter_files(filter_type, inputs, guidance_pattern, sigma_color, sigma_spatial, path_out,
                 iterations=1, rank=None, world=None):
    """filter_reflectance.read_filter_write over this rank's share of `inputs`; returns the
    list of files written."""
    import torch
    fr._check_params(filter_type, sigma_color, sigma_spatial)
    mine = my_slice(inputs, rank, world)

    def load(f):
        img, gui = iu.imread(f), iu.imread(guidance_for(f, guidance_pattern))
        if img.shape[:2] != gui.shape[:2]:
            raise ValueError("input {} and its guidance differ in size".format(f))
        return f, img, gui

    def compute(loaded):
        jobs = []
        for group in group_by_shape(loaded, lambda t: t[1].shape):
            imgs = np.stack([t[1] for t in group])
            guis = np.stack([t[2] for t in group])
            # A grey PNG comes back from imread as three equal channels (the CNN's `-r.png` always
            # does).  The channels never mix in either filter, so such a group is filtered as one
            # channel - a third of the transfers and of the guided filter's scratch, all tile
            # shapes of the bilateral kernel - and replicated afterwards: identical bytes.
            grey_src = _is_grey(imgs)
            if grey_src:
                imgs = imgs[..., :1]
            images = torch.from_numpy(np.ascontiguousarray(imgs)).cuda()
            if filter_type == "bilateral" and grey_src and _is_grey(guis):
                joints = torch.from_numpy(np.ascontiguousarray(guis[..., :1])).cuda()
                out = images
                for _ in range(iterations):
                    out = ops.joint_bilateral_u8(joints, out, -1, sigma_color, sigma_spatial,
                                                 grey_as_bgr=True)
            else:
                joints = torch.from_numpy(guis).cuda()
                out = fr.apply_filter_batch(filter_type, images, joints, sigma_color,
                                            sigma_spatial, iterations=iterations)
            out = out.cpu().numpy()
            if grey_src:
                out = np.repeat(out, 3, axis=3)
            for (f, _, _), res in zip(group, out):
                name = f
                for _ in range(iterations):  # the chained CLI runs append the suffix once per pass
                    name = fr.output_filename(name, path_out, filter_type, sigma_color,
                                              sigma_spatial)
                jobs.append((name, res))
        return jobs

    return pipeline(mine, load, compute)


def decompose_files(inputs, path_out, rank=None, world=None):
    """decompose_image over this rank's share of `inputs`: CNN, colourisation, percentile
    normalisation and the sRGB byte conversion all run batched on the device; the host only
    decodes and encodes the image files."""
    import torch
    from . import decompose_with_trained_CNN as dc
    mine = my_slice(inputs, rank, world)
    firsts = []

    def compute(loaded):
        jobs = []
        for group in group_by_shape(loaded, lambda t: t[1].shape):
            images = torch.from_numpy(np.stack([t[1] for t in group])).cuda()
            _, r8, refl, shad = dc.decompose_batch(images)
            r8, refl, shad = r8.cpu().numpy(), refl.cpu().numpy(), shad.cpu().numpy()
            for i, (f, _) in enumerate(group):
                base = os.path.splitext(os.path.basename(f))[0]
                jobs.append((os.path.join(path_out, base + "-r.png"), r8[i]))
                jobs.append((os.path.join(path_out, base + "-r_colorized.png"), refl[i]))
                jobs.append((os.path.join(path_out, base + "-s_colorized.png"), shad[i]))
                firsts.append(os.path.join(path_out, base + "-r.png"))
        return jobs

    pipeline(mine, lambda f: (f, iu.imread(f)), compute)
    return firsts


def build_parser():
    parser = argparse.ArgumentParser(description="Batched, multi-GPU front-end of the two tools.")
    sub = parser.add_subparsers(dest="command")
    f = sub.add_parser("filter", help="filter_reflectance over many files")
    f.add_argument("--inputs", nargs="+", required=True, help="files or glob patterns")
    f.add_argument("--guidance", default=None, help="pattern for the guidance file (see module doc)")
    f.add_argument("--path_out", required=True)
    f.add_argument("--sigma_color", type=float, required=True)
    f.add_argument("--sigma_spatial", type=float, required=True)
    f.add_argument("--filter_type", required=True)
    f.add_argument("--iterations", type=int, default=1)
    d = sub.add_parser("decompose", help="decompose_with_trained_CNN over many files")
    d.add_argument("--inputs", nargs="+", required=True)
    d.add_argument("--path_out", required=True)
    return parser


def main(argv=None):
    args = build_parser().parse_args(sys.argv[1:] if argv is None else argv)
    if args.command is None:
        build_parser().print_help()
        return 0
    import torch
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    files = expand_inputs(args.inputs)
    if args.command == "filter":
        out = filter_files(args.filter_type, files, args.guidance, args.sigma_color,
                           args.sigma_spatial, args.path_out, iterations=args.iterations)
    else:
        out = decompose_files(files, args.path_out)
    print("rank {}: wrote {} file(s)".format(os.environ.get("RANK", "0"), len(out)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
