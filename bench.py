#!/usr/bin/env python
"""bench.py -- megapixels/s of the joint-bilateral hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of rf_jbf_u8 (sigma_color=20, sigma_spatial=22 -> radius 33, 3,409 taps
per pixel) over one batch of B synthetic 1920x1080 images per GPU, inputs already resident in
HBM.  Image batches shard across GPUs with no collective (weak scaling: B per GPU is fixed);
torch.distributed only carries the barrier and the max-over-ranks of the timed region.
Rank 0 prints ONE JSON line with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
JBF_BYTES_PER_PX = 9.0         # 3 B joint + 3 B src read, 3 B dst written
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--sigma-color", type=float, default=20.0)
    ap.add_argument("--sigma-spatial", type=float, default=22.0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target CPU time of the cpu_baseline sample (0 disables it)")
    ap.add_argument("--no-extras", "--no-colour-src", dest="no_extras", action="store_true",
                    help="skip the secondary colour-src and single-image launches (keeps a "
                         "profile of this command to the one timed kernel shape)")
    return ap.parse_args()


def synth_batch(torch, n, h, w, seed, device):
    """Seeded natural-image-like inputs generated on the device: `joint` = RGB scene with
    correlated channels, `src` = grey reflectance-like map replicated to 3 channels (the
    README example: CNN prediction filtered with the photo as guidance).  uint8 [n,h,w,3]."""
    import torch.nn.functional as F
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)

    def field(n_img, ch, octaves=5):
        acc = torch.zeros((n_img, ch, h, w), device=device)
        amp, tot = 1.0, 0.0
        for o in range(octaves):
            gh, gw = 2 + (h >> (octaves - o)), 2 + (w >> (octaves - o))
            g = torch.randn((n_img, ch, gh, gw), device=device, generator=gen)
            acc += amp * F.interpolate(g, size=(h, w), mode="bilinear", align_corners=True)
            tot += amp * amp
            amp *= 0.55
        return acc / tot ** 0.5

    joint = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    src = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    step = 16
    for i in range(0, n, step):
        m = min(step, n - i)
        base = field(m, 1)
        rgb = 0.9 * base + 0.44 * field(m, 3)
        rgb = 128 + 48 * rgb + 1.5 * torch.randn(rgb.shape, device=device, generator=gen)
        joint[i:i + m] = rgb.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
        r = 0.15 + 0.85 * torch.sigmoid(1.5 * field(m, 1, octaves=4))
        grey = (r.clamp(0, 0.9999) * 255).floor().to(torch.uint8).permute(0, 2, 3, 1)
        src[i:i + m] = grey.expand(-1, -1, -1, 3)
    return joint.contiguous(), src.contiguous()


def usable_cores():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, quota // int(fh2.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(joint0, src0, sigma_color, sigma_spatial, target_s):
    """Oracle (CPU restatement, row-parallel OpenMP) timed on a bounded strip of one image."""
    import numpy as np
    from oracle import c_oracle
    cores = usable_cores()
    h, w = joint0.shape[:2]
    r = c_oracle.jbf_radius(-1, sigma_spatial)

    def run(rows):
        rows = min(rows, h)
        hh = min(h, rows + 2 * r)
        t0 = time.perf_counter()
        c_oracle.joint_bilateral_filter(joint0[:hh], src0[:hh], -1, sigma_color, sigma_spatial,
                                        threads=cores)
        return time.perf_counter() - t0, hh

    t, hh = run(16)
    for _ in range(2):                            # grow the strip until it costs ~target_s
        if t >= 0.6 * target_s or hh >= h:
            break
        rows = int(max(16, min(h, target_s * (hh * w / t) / w)))
        t, hh = run(rows - 2 * r if rows > 4 * r else rows)
    reps = 1
    while t < 0.6 * target_s and reps < 64:       # whole image is cheap on this host: repeat it
        t2, _ = run(hh)
        t += t2
        reps += 1
    mp = reps * hh * w / 1e6
    return {"value": mp / t, "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": "%d pass(es) over a %d x %d strip of image 0 (%.2f MP) in %.1f s, OpenMP "
                      "threads=%d, oracle/rf_oracle.c (restatement of OpenCV's 8u joint "
                      "bilateral, not OpenCV itself)" % (reps, hh, w, mp, t, cores)}


def valu_roofline(n, h, w, radius, kernel_ms, taps_per_launch):
    """Issue-floor time of the launch = column steps x 26 instructions x 2 cycles / (1024 SIMDs
    x 2.4 GHz).  Column steps per wave follow the kernel's row walk: per tap row, groups of 4
    columns covering -hw4 .. hw4+3 (hw = half-width of the disk on that row, hw4 = hw rounded up
    to a multiple of 4)."""
    steps_per_wave = 0
    for i in range(-radius, radius + 1):
        hw = int((radius * radius - i * i) ** 0.5)
        hw4 = (hw + 3) & ~3
        steps_per_wave += 4 * (hw4 // 2 + 1)
    tiles = n * ((w + 63) // 64) * ((h + 63) // 64)
    wave_steps = tiles * 16 * steps_per_wave          # 16 waves per 64x64 tile
    floor_s = wave_steps * 26 * 2 / (1024 * 2.4e9)
    return {"bound": "valu-issue", "instructions_per_column_step": 26,
            "column_steps_per_launch": wave_steps, "floor_ms": floor_s * 1e3,
            "frac": floor_s / (kernel_ms * 1e-3), "taps_per_s": taps_per_launch / (kernel_ms * 1e-3),
            "lane_ops_per_tap": VALU_LANE_OPS_PER_S / (taps_per_launch / (kernel_ms * 1e-3))}


def main():
    args = parse_args()
    import torch
    import reflectance_filtering_amd as rf
    from reflectance_filtering_amd import sharding

    rank, world, local = sharding.init_distributed()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local % torch.cuda.device_count())
    device = torch.device("cuda", torch.cuda.current_device())
    rf._ffi.load_library()

    n, h, w = args.batch, args.height, args.width
    # weak scaling: every rank owns `batch` images of a global batch of world*batch
    lo, hi = sharding.shard_range(n * world, world, rank)
    assert hi - lo == n
    joint, src = synth_batch(torch, n, h, w, seed=1234 + 1000 * 2 + lo, device=device)
    dst = torch.empty_like(src)

    def step():
        rf.ops.joint_bilateral_u8(joint, src, -1, args.sigma_color, args.sigma_spatial, out=dst)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sharding.barrier(world)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    sharding.barrier(world)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    px_local = float(n) * h * w * args.steps
    px_total, t_max = sharding.reduce_job(px_local, elapsed, world, device=device)
    kernel_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, args.steps)

    # secondary figure (not `value`): the same launch with a 3-channel colour src, which takes
    # the 3-channel accumulation path (the headline src is the grey CNN-style map the reference
    # filters, for which the kernel accumulates one channel and replicates it: identical bits)
    rgb_ms = None
    if rank == 0 and not args.no_extras:
        src_rgb = joint.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
        rf.ops.joint_bilateral_u8(joint, src_rgb, -1, args.sigma_color, args.sigma_spatial, out=dst)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rf.ops.joint_bilateral_u8(joint, src_rgb, -1, args.sigma_color, args.sigma_spatial, out=dst)
        e1.record()
        torch.cuda.synchronize()
        rgb_ms = e0.elapsed_time(e1)
        del src_rgb

    # BF(CNN, CNN) (SURVEY.md 8d input A): src and joint are the same grey map, passed as one
    # channel with the joint counted as three equal channels - what the fused chain and the
    # file front-ends run; identical bytes to filtering the 3-channel copies
    grey_ms = None
    if rank == 0 and not args.no_extras:
        g1 = src[..., :1].contiguous()
        g1j = g1.clone()
        d1 = torch.empty_like(g1)
        rf.ops.joint_bilateral_u8(g1j, g1, -1, args.sigma_color, args.sigma_spatial, out=d1,
                                  grey_as_bgr=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rf.ops.joint_bilateral_u8(g1j, g1, -1, args.sigma_color, args.sigma_spatial, out=d1,
                                  grey_as_bgr=True)
        e1.record()
        torch.cuda.synchronize()
        grey_ms = e0.elapsed_time(e1)
        del g1, g1j, d1

    # BASELINE config C2: one 1080p image on one GPU (latency of a single launch)
    single_ms = None
    if rank == 0 and not args.no_extras:
        j1, s1, d1 = joint[:1].contiguous(), src[:1].contiguous(), dst[:1].contiguous()
        rf.ops.joint_bilateral_u8(j1, s1, -1, args.sigma_color, args.sigma_spatial, out=d1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rf.ops.joint_bilateral_u8(j1, s1, -1, args.sigma_color, args.sigma_spatial, out=d1)
        e1.record()
        torch.cuda.synchronize()
        single_ms = e0.elapsed_time(e1) / 10

    if rank != 0:
        return
    value = px_total / 1e6 / t_max
    radius = int(round(args.sigma_spatial * 1.5))
    taps = sum(1 for i in range(-radius, radius + 1) for j in range(-radius, radius + 1)
               if (i * i + j * j) ** 0.5 <= radius)
    launch_px = float(n) * h * w
    achieved = launch_px * JBF_BYTES_PER_PX / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "jbf_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            with open(pmc) as fh:
                rec = json.load(fh)
            if (rec.get("batch"), rec.get("height"), rec.get("width")) == (n, h, w):
                traffic = rec.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
    out = {
        "metric": "megapixels/sec joint-bilateral sigma_c=20 sigma_s=22 @1080p",
        "value": value, "unit": "MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": t_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "joint bilateral c=%g s=%g (radius %d, %d taps/px), batch %d x "
                               "%dx%d uint8 BGR per GPU, RGB scene as joint, grey map as src"
                               % (args.sigma_color, args.sigma_spatial, radius, taps, n, w, h),
                   "batch_per_gpu": n, "height": h, "width": w, "sharding": "image batch, "
                   "contiguous slices, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "jbf_tile64_kernel<3,32,16>", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": launch_px * JBF_BYTES_PER_PX},
        # the bound that actually limits an exact brute-force bilateral: VALU issue.  The grey
        # tap loop retires 26 VALU wave-instructions per 4-output column step; a gfx950 SIMD
        # issues at most one per 2 cycles (tools/microbench/valu_rates2.hip), 1024 SIMDs, 2.4 GHz.
        "valu": valu_roofline(n, h, w, radius, kernel_ms, launch_px * taps),
    }
    if single_ms:
        out["single_image"] = {"ms": single_ms, "value": h * w / 1e6 / (single_ms * 1e-3),
                               "unit": "MP/s", "note": "one %dx%d image per launch" % (w, h)}
    if grey_ms:
        out["grey_joint"] = {"value": launch_px / 1e6 / (grey_ms * 1e-3), "unit": "MP/s",
                             "kernel_ms": grey_ms,
                             "note": "BF(CNN,CNN): same launch with the grey map as joint and src, "
                                     "1-channel buffers, joint counted as 3 equal channels"}
    if rgb_ms:
        out["colour_src"] = {"value": launch_px / 1e6 / (rgb_ms * 1e-3), "unit": "MP/s",
                             "kernel_ms": rgb_ms, "note": "same launch, 3-channel colour src"}
    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(joint[0].cpu().numpy(), src[0].cpu().numpy(),
                                           args.sigma_color, args.sigma_spatial, args.cpu_seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
