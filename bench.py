#!/usr/bin/env python
"""bench.py -- megapixels/s of the joint-bilateral hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config north_star|c2|c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` without a torchrun environment launches the N ranks itself (one child process per
GPU, before anything touches the GPU); under torchrun it must equal WORLD_SIZE.

Default workload ("north_star"): one step = one pass of rf_jbf_u8 (sigma_color=20,
sigma_spatial=22 -> radius 33, 3,409 taps per pixel) over one batch of 256 synthetic 1920x1080
images per GPU, inputs already resident in HBM.  Image batches shard across GPUs with no
collective (weak scaling: the per-GPU batch is fixed); torch.distributed only carries the
barrier and the max-over-ranks of the timed region.  Rank 0 prints ONE JSON line with the
contract fields plus `roofline` and `cpu_baseline`, and (N = 1, default config) driver-timed lines
for the other BASELINE configurations: `c2_single_image`, `c3_chain` (CNN -> BF(CNN,CNN), 256 IIW
images) and `c5_gf` (3x guided filter r=45 eps=3 at 3840x2160).

Other workloads, same contract (`--config`):
    c2  one 1920x1080 image per step (BASELINE configs[1] as written)
    c3  256 x 500x333 per GPU: uint8 BGR -> 1x1 CNN -> trunc(r*255) -> BF(CNN,CNN)
    c4  512 x 1920x1080 joint bilateral per GPU (4096 images over 8 GPUs)
    c5  128 x 3840x2160 per GPU, 3x guided filter c=3.0 s=45.0, piecewise-constant guide
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md:36: the rate a plain device copy reaches
JBF_BYTES_PER_PX = 9.0         # 3 B joint + 3 B src read, 3 B dst written
GF_BYTES_PER_PX_X3 = 21.0      # 3x chain, shared guide, uint8 hand-offs: 9 + 6 + 6
C3_BYTES_PER_PX = 6.0          # fused chain: 3 B in, 3 B out at the cv2.imread layout
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz

CONFIGS = {
    # name: (kind, batch per GPU, height, width)
    "north_star": ("jbf", 256, 1080, 1920),
    "c2": ("jbf", 1, 1080, 1920),
    "c3": ("chain", 256, 333, 500),
    "c4": ("jbf", 512, 1080, 1920),
    "c5": ("gf3", 128, 2160, 3840),
    # the reference's published 3x guided chain filters a COLOUR reflectance
    # (/root/reference/README.md:66): the same guide, a 3-channel colour src, 64 images
    "c5c": ("gf3c", 64, 2160, 3840),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="north_star")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step "
                    "(default: the config's)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--sigma-color", type=float, default=20.0)
    ap.add_argument("--sigma-spatial", type=float, default=22.0)
    ap.add_argument("--src", choices=("grey", "colour"), default="grey",
                    help="joint-bilateral workloads: the src image - the grey CNN-style map the "
                         "reference filters (default, the metric's input) or a 3-channel colour "
                         "image (profiling the colour tap loop; named in config.workload)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target CPU time of the cpu_baseline sample (0 disables it)")
    ap.add_argument("--traffic", choices=("auto", "live", "profile", "off"), default="auto",
                    help="roofline.traffic of the metric launch: 'live' measures it now (two child "
                         "runs of this script under rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE "
                         "in separate passes), 'profile' takes the committed summary of the same "
                         "launch shape, 'auto' = live on one GPU when rocprofv3 is there and this "
                         "process is not itself being profiled, else profile; 'off' = null")
    ap.add_argument("--no-extras", "--no-colour-src", dest="no_extras", action="store_true",
                    help="skip the secondary launches and the C2/C3/C5 lines (keeps a profile of "
                         "this command to the one timed kernel shape)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- self-launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, deadline_s=3600.0):
    """Start one child per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* as torchrun would set them).
    The parent never initialises the GPU; rank 0's stdout is this process's stdout, the other
    ranks' stdout goes to this process's stderr (their diagnostics stay visible, the one JSON
    line stays alone on stdout).  All children are polled together: the first non-zero exit (or
    the deadline) ends the others, so a rank that dies during start-up cannot leave the rest
    waiting in a barrier."""
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=env, stdout=None if rank == 0 else sys.stderr))
    t_end = time.monotonic() + deadline_s
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = abs(code) or 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n"
                                 % (procs.index(p), code))
        if live and rc == 0:
            if time.monotonic() > t_end:
                rc = 124
                sys.stderr.write("bench.py: ranks still running after %.0f s; stopping them\n"
                                 % deadline_s)
            else:
                time.sleep(0.05)
    for p in live:                      # these are this process's own children, by handle
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


# --------------------------------------------------------------------------- inputs
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
    step = 16 if h * w <= 1080 * 1920 else 4
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


def flat_guide(scene, seed=9000):
    """Piecewise-constant ("L1-flattened") guidance of BASELINE config C5 as SURVEY.md 8(d)
    specifies it: seeded Voronoi cells (200 - 2,000 regions by image size: about one per 8,192
    pixels), each of one flat colour - the scene's colour at the cell's seed point, so the
    regions follow the scene the src map belongs to - plus a +-1 dither, generated on the device.
    The seed points are one per cell of a jittered grid and a pixel takes the nearest seed of the
    3 x 3 grid cells around its own (the exact Voronoi partition of those seeds but for rare
    slivers whose nearest seed lies two cells away)."""
    import torch
    n, h, w, _ = scene.shape
    dev = scene.device
    regions = int(min(2000, max(200, h * w / 8192.0)))
    regions = max(1, min(regions, (h * w) // 16))
    gw = max(1, int(round((regions * w / float(h)) ** 0.5)))
    gh = max(1, int(round(regions / float(gw))))
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    out = torch.empty_like(scene)
    py = torch.arange(h, device=dev, dtype=torch.float32)[None, :, None]
    px = torch.arange(w, device=dev, dtype=torch.float32)[None, None, :]
    cy = (torch.arange(h, device=dev) * gh) // h
    cx = (torch.arange(w, device=dev) * gw) // w
    step = max(1, min(n, (64 << 20) // max(1, h * w)))          # images per pass: ~64 Mpx of temporaries
    for i0 in range(0, n, step):
        m = min(step, n - i0)
        img_ix = torch.arange(m, device=dev)[:, None, None]
        sy = (torch.arange(gh, device=dev)[None, :, None]
              + torch.rand((m, gh, gw), device=dev, generator=gen)) * (h / float(gh))
        sx = (torch.arange(gw, device=dev)[None, None, :]
              + torch.rand((m, gh, gw), device=dev, generator=gen)) * (w / float(gw))
        colour = scene[i0:i0 + m][img_ix, sy.long().clamp_(0, h - 1), sx.long().clamp_(0, w - 1)]
        colour = colour.reshape(m, gh * gw, 3)
        best = torch.full((m, h, w), float("inf"), device=dev)
        lab = torch.zeros((m, h, w), dtype=torch.long, device=dev)
        for dy in (-1, 0, 1):
            ny = (cy + dy).clamp_(0, gh - 1)
            for dx in (-1, 0, 1):
                nx = (cx + dx).clamp_(0, gw - 1)
                cell = (ny[:, None] * gw + nx[None, :])[None].expand(m, -1, -1)     # [m,h,w]
                d = (py - sy.reshape(m, -1).gather(1, cell.reshape(m, -1)).reshape(m, h, w)) ** 2
                d += (px - sx.reshape(m, -1).gather(1, cell.reshape(m, -1)).reshape(m, h, w)) ** 2
                upd = d < best
                best = torch.where(upd, d, best)
                lab = torch.where(upd, cell, lab)
        img = colour.gather(1, lab.reshape(m, -1, 1).expand(-1, -1, 3)).reshape(m, h, w, 3).to(torch.int16)
        img += torch.randint(-1, 2, (m, h, w, 3), device=dev, generator=gen, dtype=torch.int16)
        out[i0:i0 + m] = img.clamp_(0, 255).to(torch.uint8)
    return out


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


# --------------------------------------------------------------------------- CPU baseline
def cpu_baseline_opencv(joint0, src0, sigma_color, sigma_spatial, target_s):
    """T2: the reference's own native call, cv2.ximgproc.jointBilateralFilter(joint, image, -1,
    sigma_color, sigma_spatial) (/root/reference/filter_reflectance.py:60-64), default OpenCV
    threading, on a bounded strip of image 0.  Returns None when OpenCV-contrib is not installed.
    Also reports how the oracle compares with it (what would pin the oracle)."""
    try:
        import cv2
        ximgproc = cv2.ximgproc
        ximgproc.jointBilateralFilter
    except (ImportError, AttributeError):
        return None
    import hashlib
    import numpy as np
    from oracle import c_oracle
    h, w = joint0.shape[:2]
    rows = min(h, 200)
    t0 = time.perf_counter()
    ximgproc.jointBilateralFilter(joint0[:rows], src0[:rows], -1, sigma_color, sigma_spatial)
    t = time.perf_counter() - t0
    rows = int(max(rows, min(h, rows * 0.8 * target_s / max(t, 1e-6))))
    reps, total, out = 0, 0.0, None
    while total < 0.6 * target_s and reps < 64:
        t0 = time.perf_counter()
        out = ximgproc.jointBilateralFilter(joint0[:rows], src0[:rows], -1, sigma_color,
                                            sigma_spatial)
        total += time.perf_counter() - t0
        reps += 1
    mp = reps * rows * w / 1e6
    crop = min(rows, 96)
    want = c_oracle.joint_bilateral_filter(joint0[:crop], src0[:crop], -1, sigma_color,
                                           sigma_spatial, threads=usable_cores())
    got = ximgproc.jointBilateralFilter(joint0[:crop], src0[:crop], -1, sigma_color, sigma_spatial)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    return {"value": mp / total, "unit": "MP/s", "cores": int(cv2.getNumThreads()),
            "kind": "reference",
            "sample": "%d pass(es) of cv2.ximgproc.jointBilateralFilter over a %d x %d strip of "
                      "image 0 (%.2f MP) in %.1f s, OpenCV %s, %d threads"
                      % (reps, rows, w, mp, total, cv2.__version__, cv2.getNumThreads()),
            "opencv_version": cv2.__version__,
            "opencv_build_sha256": hashlib.sha256(cv2.getBuildInformation().encode()).hexdigest(),
            "oracle_vs_opencv": {"rows": crop, "max_abs": int(diff.max()),
                                 "flip_rate": float((diff != 0).mean())}}


def cpu_baseline(joint0, src0, sigma_color, sigma_spatial, target_s):
    """Oracle (CPU restatement, row-parallel OpenMP) timed on a bounded strip of one image."""
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
                      "bilateral, not OpenCV itself: cv2.ximgproc is not installed here)"
                      % (reps, hh, w, mp, t, cores)}


def valu_roofline(n, h, w, radius, kernel_ms, taps_per_launch, clock_mhz=None):
    """Issue-floor time of the launch = column steps x 26 instructions x 2 cycles / (1024 SIMDs
    x shader clock).  Column steps per wave follow the kernel's row walk: per tap row, whole groups
    of 4 columns from the even column -hws on, covering -hw .. hw+3 (hw = half-width of the disk on
    that row, hws = hw rounded up to even; until round 5 rows started at a multiple of 4: 3,764
    instead of 3,664 steps per output quad at radius 33; the four outputs of a lane need 3,610).  The
    26 are the step's tap arithmetic; the loop spends 2 more per group of 4 steps on its two texel
    addresses.  The clock is the one measured under this launch (`measured_clock_mhz`:
    a one-wave probe on a second stream, s_memtime over s_memrealtime); `floor_ms_at_2400mhz` is
    the same floor at the 2.4 GHz peak clock of MI355X_MICROARCH.md."""
    steps_per_wave = 0
    for i in range(-radius, radius + 1):
        hw = int((radius * radius - i * i) ** 0.5)
        hws = (hw + 1) & ~1
        steps_per_wave += (hws + hw + 4 + 3) & ~3
    tiles = n * ((w + 63) // 64) * ((h + 63) // 64)
    wave_steps = tiles * 16 * steps_per_wave          # 16 waves per 64x64 tile
    cycles = wave_steps * 26 * 2 / 1024.0
    floor_peak_s = cycles / 2.4e9
    mhz = clock_mhz if clock_mhz else 2400.0
    floor_s = cycles / (mhz * 1e6)
    # (The step's 9 four-cycle instructions - 4 v_sad_u8, 4 v_lshl_add_u32, 1 v_cvt_f32_ubyte - are of the
    # kind that overlaps with a two-cycle neighbour when the stream interleaves them, as the asm loop
    # does: a v_sad_u8 + v_mul_f32 pair issues in 4.1 cycles, likewise v_lshl_add_u32, v_add3_u32 and
    # v_cvt_f32_ubyte*; tools/microbench/pipe_overlap.hip, profiles/r05_pipe_overlap.txt.  So 2 cycles
    # per instruction IS this loop's issue floor; what keeps it at ~0.68 of it is the LDS gather
    # pipeline working beside it - see lds_cobound.)
    return {"bound": "valu-issue", "instructions_per_column_step": 26,
            "column_steps_per_launch": wave_steps, "clock_mhz": mhz,
            "clock_source": ("s_memtime / s_memrealtime probe beside the launch" if clock_mhz
                             else "2.4 GHz peak (no probe)"),
            "floor_ms": floor_s * 1e3, "floor_ms_at_2400mhz": floor_peak_s * 1e3,
            "frac": floor_s / (kernel_ms * 1e-3),
            "frac_at_2400mhz": floor_peak_s / (kernel_ms * 1e-3),
            "taps_per_s": taps_per_launch / (kernel_ms * 1e-3),
            "lane_ops_per_tap": VALU_LANE_OPS_PER_S / (taps_per_launch / (kernel_ms * 1e-3))}


def lds_cobound(column_steps_per_launch, kernel_ms, clock_mhz=None):
    """The LDS side of the tap loop, priced like `valu_roofline` prices the VALU side.  Per 4-output
    column step a wave issues 4.5 LDS instructions: 4 colour-LUT gathers (ds_read_b32) and half a
    ds_read2_b32 (the texels of two columns per instruction); the weight window of a 4-column group,
    two broadcast ds_read_b128 until round 5, now comes through a scalar load.  A CU's four SIMDs
    share ONE LDS pipeline; its cost per wave-instruction, from tools/microbench/valu_rates.hip with all
    four SIMDs issuing (profiles/r01_valu_rates.txt: 8.98 cycles per SIMD = 2.25 per CU for
    conflict-free b32 / b64 reads, 4.2 for fully random gathers): 2.4 cycles for the 32x replicated LUT
    gather (7 % bank conflicts), 2.25 for the ds_read2_b32.  `frac` = that floor over the measured launch
    time; `busy_measured` is SQ_LDS_IDX_ACTIVE over the kernel's CU-cycles from the committed rocprofv3
    pass (not measured by this run).  VALU issue and the LDS pipeline work in the SAME cycles and are
    chained - every gather's result is the operand of the next multiply - so the loop sits between two
    floors, not 30 % below one."""
    per_wave_step = 4 * 2.4 + 0.5 * 2.25
    cycles = column_steps_per_launch * per_wave_step / 256.0      # one LDS pipeline per CU
    mhz = clock_mhz if clock_mhz else 2400.0
    floor_s = cycles / (mhz * 1e6)
    busy, src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "jbf_pmc_traffic.json")) as fh:
            rec = json.load(fh)
        busy, src = rec.get("lds_busy"), rec.get("lds_busy_source")
    except (OSError, ValueError):
        pass
    return {"bound": "lds-issue", "wave_instructions_per_column_step": 4.5,
            "lds_cycles_per_wave_step": per_wave_step, "clock_mhz": mhz,
            "floor_ms": floor_s * 1e3, "frac": floor_s / (kernel_ms * 1e-3),
            "busy_measured": busy, "busy_source": src}


def measured_clock_mhz(torch, rf, step, kernel_ms):
    """Shader clock while `step` runs: launch the step, then the one-wave probe
    (rf_debug_clock_probe) on a second stream for the middle half of the step's duration."""
    lib = rf._ffi.load_library()
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    micros = int(max(200, min(200000, kernel_ms * 1e3 * 0.5)))
    torch.cuda.synchronize()
    step()
    time.sleep(min(0.05, kernel_ms * 1e-3 * 0.2))   # let the launch spread over the chip first
    import ctypes
    rc = lib.rf_debug_clock_probe(out.data_ptr(), micros, ctypes.c_void_p(side.cuda_stream))
    rf._ffi.check(rc, "rf_debug_clock_probe")
    torch.cuda.synchronize()
    cyc, ticks = (int(v) for v in out.cpu())
    return 100.0 * cyc / ticks if ticks > 0 else None


# --------------------------------------------------------------------------- workloads
class Workload:
    """One BASELINE configuration: device-resident inputs + the step that is timed."""

    def __init__(self, kind, n, h, w, args, torch, rf, device, seed):
        self.kind, self.n, self.h, self.w = kind, n, h, w
        self.args, self.torch, self.rf = args, torch, rf
        sc, ss = args.sigma_color, args.sigma_spatial
        scene, grey = synth_batch(torch, n, h, w, seed, device)
        if kind == "jbf":
            self.joint, self.src = scene, grey
            if getattr(args, "src", "grey") == "colour":
                self.src = scene.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
            self.dst = torch.empty_like(grey)
            self.step = lambda: rf.ops.joint_bilateral_u8(self.joint, self.src, -1, sc, ss,
                                                          out=self.dst)
            self.bytes_per_px = JBF_BYTES_PER_PX
            radius = int(round(ss * 1.5))
            self.name = ("joint bilateral c=%g s=%g (radius %d), batch %d x %dx%d uint8 BGR per "
                         "GPU, RGB scene as joint, %s as src"
                         % (sc, ss, radius, n, w, h,
                            "3-channel colour image" if getattr(args, "src", "grey") == "colour"
                            else "grey map"))
        elif kind in ("gf3", "gf3c"):
            self.guide, self.src = flat_guide(scene), grey
            if kind == "gf3c":
                self.src = scene.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
                del grey
            del scene
            self.dst = torch.empty_like(self.src)
            self.ws = rf.ops.gf_workspace(n, h, w, 3, 45, device, torch)
            self.step = lambda: rf.ops.guided_filter_u8(self.guide, self.src, 45, 3.0,
                                                        iterations=3, out=self.dst,
                                                        workspace=self.ws)
            self.bytes_per_px = GF_BYTES_PER_PX_X3
            self.name = ("3x guided filter c=3.0 s=45.0 (radius 45, eps 3), batch %d x %dx%d per "
                         "GPU, piecewise-constant guide (seeded Voronoi cells of flat colour +-1), %s as src"
                         % (n, w, h, "grey map" if kind == "gf3" else "3-channel colour image"))
        else:  # chain
            self.scene = scene
            del grey
            self.step = lambda: rf.decompose_and_filter_batch(self.scene)
            self.bytes_per_px = C3_BYTES_PER_PX
            self.name = ("1x1 CNN -> trunc(r*255) -> BF(CNN,CNN) c=%g s=%g, batch %d x %dx%d per GPU"
                         % (sc, ss, n, w, h))
        self.pixels = float(n) * h * w

    def timed_steps(self, steps, world, sharding, stub=False):
        """barrier + sync | K steps with per-step events on the launch stream | sync + barrier."""
        if stub:
            sharding.barrier(world)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            sharding.barrier(world)
            return time.perf_counter() - t0, None
        torch = self.torch
        torch.cuda.synchronize()
        sharding.barrier(world)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(steps)]
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            self.step()
            e1.record()
        torch.cuda.synchronize()
        sharding.barrier(world)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        return elapsed, sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, steps)


class StubWorkload(Workload):
    """CPU stand-in for the kernel call (tests of the rank flow with gloo, no GPU): same pixel
    accounting, the step only sleeps."""

    def __init__(self, kind, n, h, w):
        self.kind, self.n, self.h, self.w = kind, n, h, w
        self.pixels = float(n) * h * w
        self.bytes_per_px = JBF_BYTES_PER_PX
        self.name = "stub (no kernel), batch %d x %dx%d per rank" % (n, w, h)
        self.step = lambda: time.sleep(0.01)


def one_shot(torch, fn, reps=1):
    """Median-free quick timing of a secondary launch: warm once, then `reps` timed calls."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def committed_traffic(n, h, w):
    """HBM-side bytes per launch of the metric kernel.  NOT measured by this run: PMC counters
    need rocprofv3 passes of their own; the value comes from the committed profile summary of the
    same launch shape (profiles/jbf_pmc_traffic.json, written by tools/make_profiles.py)."""
    pmc = os.path.join(ROOT, "profiles", "jbf_pmc_traffic.json")
    try:
        with open(pmc) as fh:
            rec = json.load(fh)
        if (rec.get("batch"), rec.get("height"), rec.get("width")) == (n, h, w):
            return rec.get("hbm_bytes_per_launch"), "profiles/jbf_pmc_traffic.json"
    except (OSError, ValueError):
        pass
    return None, None


def committed_config_traffic(key, batch):
    """HBM-side bytes per step of a secondary configuration from the newest committed bench line
    that measured it live at this batch (profiles/rNN_bench.json); (None, reason) without one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")), reverse=True):
        try:
            with open(path) as fh:
                rec = json.load(fh).get(key) or {}
        except (OSError, ValueError):
            continue
        roof = rec.get("roofline") or {}
        if rec.get("batch") == batch and roof.get("traffic") and "measured by this run" in str(
                roof.get("traffic_source", "")):
            return roof["traffic"], ("profiles/%s (rocprofv3 --pmc child passes of an earlier run of "
                                     "this configuration; not measured by this run)" % os.path.basename(path))
    return None, "no committed measurement of this configuration"


def being_profiled():
    """True when this process runs under a rocprofiler tool (its own PMC passes must not nest)."""
    env = os.environ
    return any(k in env for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD",
                                  "ROCPROF_OUTPUT_PATH")) or "rocprofiler" in env.get("LD_PRELOAD", "")


# All rocprofv3 child passes of one run share one wall-clock budget (the default run has to stay
# within a minute or so of driver time whatever the box does): a pass starts only while its
# predecessor's duration is at most a third of what is left; a skipped pass is said so on stderr
# and in `traffic_source`, never silently.
CHILD_BUDGET_S = float(os.environ.get("RF_BENCH_CHILD_BUDGET_S", "70"))
_child_clock = {"t_end": None, "last": 0.0}


def child_pass_allowed():
    now = time.time()
    if _child_clock["t_end"] is None:
        _child_clock["t_end"] = now + CHILD_BUDGET_S
    left = _child_clock["t_end"] - now
    return left > 0 and _child_clock["last"] <= left / 3.0, left


def child_pass_done(seconds):
    _child_clock["last"] = seconds


def parse_counter_csv(path, counter, match="jbf"):
    """Per-dispatch values of `counter` for the kernels whose name contains `match`, from a
    rocprofv3 *_counter_collection.csv."""
    import csv
    vals = {}
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") == counter and match in r.get("Kernel_Name", ""):
                key = int(r["Dispatch_Id"])
                vals[key] = vals.get(key, 0.0) + float(r["Counter_Value"])
                parse_counter_csv.last_kernel = r["Kernel_Name"]
    return [vals[k] for k in sorted(vals)]


def live_traffic(args, n, h, w, deadline_s=150.0):
    """HBM-side bytes of ONE launch of the metric kernel, measured now: this script is run twice
    as a child under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE - separate passes, as
    MI355X_MICROARCH.md's HBM section prescribes; no trace domain beside --kernel-trace), one
    warm-up and one timed launch of the same shape each.  Units and corrections of that guide:
    both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes, so it
    is doubled (tools/microbench/fetch_calib.hip measures 2.000 for this kernel's load widths).
    Returns (bytes per launch, source text) or (None, reason)."""
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="rf_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", RF_BENCH_CHILD="1")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    got = {}
    t_end = time.time() + deadline_s
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter.lower())
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir,
                   "--", sys.executable, os.path.abspath(__file__), "--config", args.config,
                   "--batch", str(n), "--height", str(h), "--width", str(w),
                   "--sigma-color", repr(args.sigma_color), "--sigma-spatial",
                   repr(args.sigma_spatial), "--src", args.src, "--steps", "1", "--warmup", "1",
                   "--cpu-seconds", "0", "--no-extras", "--traffic", "off"]
            ok, budget_left = child_pass_allowed()
            left = min(t_end - time.time(), budget_left)
            if not ok or left < 20:
                return None, "skipped: no time left in the run's measurement budget for the %s pass" % counter
            t_pass = time.time()
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=left)
            except subprocess.TimeoutExpired:
                child_pass_done(time.time() - t_pass)
                return None, "the %s pass exceeded its time" % counter
            child_pass_done(time.time() - t_pass)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, "the %s pass failed (rc %d)" % (counter, p.returncode)
            vals = parse_counter_csv(files[0], counter)
            if not vals:
                return None, "no %s row for the kernel" % counter
            got[counter] = vals[-1] * 1024.0          # the timed launch (the last one), KiB -> B
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]
    name = getattr(parse_counter_csv, "last_kernel", "")
    live_traffic.kernel = name[:name.index(">(") + 1] if ">(" in name else name.split("(")[0]
    return total, ("measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of "
                   "this launch shape (FETCH_SIZE %.0f B raw, doubled for gfx950; WRITE_SIZE %.0f B)"
                   % (got["FETCH_SIZE"], got["WRITE_SIZE"]))


def short_kernel_name(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rf::", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def live_traffic_config(cfg, batch, deadline_s=90.0):
    """HBM-side bytes of ONE step of another BASELINE configuration (c3 / c5), per kernel: this
    script run twice as a child under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE; separate
    passes, the program directly after `--`), one step each, no warm-up (counters do not need one).
    Every dispatch of a library kernel (`rf::`) of that step is added up per kernel name; units and
    the gfx950 correction as in `live_traffic` (KiB; FETCH_SIZE doubled - tools/microbench/
    fetch_calib.hip measures 2.00 for 1-, 4-, 8-, 12- and 16-byte loads).  Returns
    ({"traffic", "traffic_source", "kernels": {name: {"fetch_bytes", "write_bytes", "launches"}}},
    None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="rf_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", RF_BENCH_CHILD="1")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    kernels = {}
    t_end = time.time() + deadline_s
    try:
        for counter, field in (("FETCH_SIZE", "fetch_bytes"), ("WRITE_SIZE", "write_bytes")):
            out_dir = os.path.join(tmp, counter.lower())
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir,
                   "--", sys.executable, os.path.abspath(__file__), "--config", cfg,
                   "--batch", str(batch), "--steps", "1", "--warmup", "0", "--cpu-seconds", "0",
                   "--no-extras", "--traffic", "off"]
            ok, budget_left = child_pass_allowed()
            left = min(t_end - time.time(), budget_left)
            if not ok or left < 15:
                return None, "skipped: no time left in the run's measurement budget for the %s pass" % counter
            t_pass = time.time()
            try:
                # (run from the repository root.  History: rocprofv3's counter tool crashed - SIGSEGV in
                #  its own thread - while the C5 guide was generated with ~9,000 tiny torch launches; the
                #  batched generator of flat_guide() has not triggered it in 12 runs)
                p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=left)
            except subprocess.TimeoutExpired:
                child_pass_done(time.time() - t_pass)
                return None, "the %s pass exceeded its time" % counter
            child_pass_done(time.time() - t_pass)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, "the %s pass failed (rc %d)" % (counter, p.returncode)
            scale = 2.0 * 1024.0 if counter == "FETCH_SIZE" else 1024.0
            seen = set()
            with open(files[0], newline="") as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") != counter or "rf::" not in r.get("Kernel_Name", ""):
                        continue
                    k = kernels.setdefault(short_kernel_name(r["Kernel_Name"]),
                                           {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0})
                    k[field] += float(r["Counter_Value"]) * scale
                    if counter == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
                        seen.add(r["Dispatch_Id"])
                        k["launches"] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not kernels:
        return None, "no library kernel in the counter files"
    total = sum(k["fetch_bytes"] + k["write_bytes"] for k in kernels.values())
    return ({"traffic": total,
             "traffic_source": "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child "
                               "passes of one step of `bench.py --config %s --batch %d` (FETCH_SIZE "
                               "doubled for gfx950), every dispatch of a library kernel added up"
                               % (cfg, batch),
             "kernels": kernels}, None)


def run_rank(args):
    stub = os.environ.get("RF_BENCH_STUB") == "1"
    if stub and os.environ.get("RF_BENCH_STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit(3)          # test hook: a rank that dies before the rendezvous
    from reflectance_filtering_amd import sharding
    backend = "gloo" if stub else None
    if not stub and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        # more ranks than devices (e.g. a 2-rank flow test on a 1-GPU box): RCCL refuses two
        # ranks on one device, and the job needs no collective on the data path, so the host-side
        # barrier and the two scalar reductions go over gloo
        if torch.cuda.device_count() < int(os.environ["WORLD_SIZE"]):
            backend = "gloo"
    rank, world, local = sharding.init_distributed(backend=backend)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with matching values, or "
                         "omit the torchrun environment to let bench.py start the ranks)"
                         % (args.gpus, world))
    kind, n, h, w = CONFIGS[args.config]
    n = args.batch or n
    h = args.height or h
    w = args.width or w
    # weak scaling: every rank owns `n` images of a global batch of world*n (contiguous slice)
    lo, hi = sharding.shard_range(n * world, world, rank)
    assert hi - lo == n
    if stub:
        torch = rf = device = None
        wl = StubWorkload(kind, n, h, w)
    else:
        import torch
        import reflectance_filtering_amd as rf
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
        torch.cuda.set_device(local % torch.cuda.device_count())
        device = torch.device("cuda", torch.cuda.current_device())
        rf._ffi.load_library()
        wl = Workload(kind, n, h, w, args, torch, rf, device, seed=1234 + 1000 * 2 + lo)

    for _ in range(args.warmup):
        wl.step()
    elapsed, kernel_ms = wl.timed_steps(args.steps, world, sharding, stub=stub)
    px_total, t_max = sharding.reduce_job(wl.pixels * args.steps, elapsed, world, device=device)
    # every rank's own per-step time (events on its launch stream; wall clock in the stub): a slow GPU
    # must not hide behind the max
    per_rank_ms = sharding.gather_scalars(kernel_ms if kernel_ms is not None
                                          else elapsed / args.steps * 1e3, world, device=device)

    extras = {}
    clock_mhz = None
    if rank == 0 and not stub and kind == "jbf" and kernel_ms and kernel_ms > 2.0:
        try:
            clock_mhz = measured_clock_mhz(torch, rf, wl.step, kernel_ms)
        except Exception as exc:                  # noqa: BLE001 - a measurement aid, never fatal
            sys.stderr.write("bench.py: clock probe failed: %r\n" % (exc,))
    if rank == 0 and not stub and not args.no_extras and kind == "jbf" and n > 1:
        sc, ss = args.sigma_color, args.sigma_spatial
        joint, src, dst = wl.joint, wl.src, wl.dst
        # the same launch with a 3-channel colour src (3-channel accumulation; the headline src is
        # the grey CNN-style map the reference filters, detected per tile: identical bits)
        src_rgb = joint.roll(shifts=(37, 91), dims=(1, 2)).contiguous()
        ms = one_shot(torch, lambda: rf.ops.joint_bilateral_u8(joint, src_rgb, -1, sc, ss, out=dst))
        extras["colour_src"] = {"value": wl.pixels / 1e6 / (ms * 1e-3), "unit": "MP/s",
                                "kernel_ms": ms, "note": "same launch, 3-channel colour src"}
        del src_rgb
        # BF(CNN, CNN) (SURVEY.md 8d input A): src and joint are the same grey map, 1-channel
        # buffers, joint counted as three equal channels (what the fused chain runs)
        g1 = src[..., :1].contiguous()
        g1j, d1 = g1.clone(), torch.empty_like(g1)
        ms = one_shot(torch, lambda: rf.ops.joint_bilateral_u8(g1j, g1, -1, sc, ss, out=d1,
                                                               grey_as_bgr=True))
        extras["grey_joint"] = {"value": wl.pixels / 1e6 / (ms * 1e-3), "unit": "MP/s",
                                "kernel_ms": ms,
                                "note": "BF(CNN,CNN): grey map as joint and src, 1-channel buffers"}
        del g1, g1j, d1
        # BASELINE config C2: one 1080p image on one GPU (latency of a single launch)
        j1, s1, d1 = joint[:1].contiguous(), src[:1].contiguous(), dst[:1].contiguous()
        ms = one_shot(torch, lambda: rf.ops.joint_bilateral_u8(j1, s1, -1, sc, ss, out=d1), reps=10)
        extras["c2_single_image"] = {"ms": ms, "value": h * w / 1e6 / (ms * 1e-3), "unit": "MP/s",
                                     "note": "one %dx%d image per launch" % (w, h)}
        del j1, s1, d1
    # HBM-side bytes of the metric launch (the contract's `roofline.traffic`): first in line for the
    # run's measurement budget, before the other configurations' passes
    headline_traffic = (None, None)
    if rank == 0 and not stub:
        traffic, source = None, None
        mode = args.traffic
        if kind == "jbf" and mode != "off":
            if mode == "live" or (mode == "auto" and world == 1 and not being_profiled()
                                  and os.environ.get("RF_BENCH_CHILD") != "1"):
                try:
                    traffic, source = live_traffic(args, n, h, w)
                except Exception as exc:              # noqa: BLE001 - a measurement aid, never fatal
                    traffic, source = None, repr(exc)
                if traffic is None:
                    sys.stderr.write("bench.py: live traffic measurement unavailable: %s\n" % source)
            if traffic is None:
                traffic, source = committed_traffic(n, h, w)
                if source:
                    source += (" (rocprofv3 --pmc passes of this launch shape; not measured by "
                               "this run)")
        headline_traffic = (traffic, source)
    # the CPU baseline runs on rank 0 at ANY N ("next to the reference CPU filter ... in the same
    # run", BASELINE.json): after the final barrier, when the other ranks have left the node's cores
    image0 = None
    if rank == 0 and kind == "jbf" and args.cpu_seconds > 0:
        if stub:
            import numpy as np
            rng = np.random.default_rng(5)
            image0 = (rng.integers(0, 256, (h, w, 3), dtype=np.uint8),
                      rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        else:
            image0 = (wl.joint[0].cpu().numpy(), wl.src[0].cpu().numpy())
    if (rank == 0 and not stub and world == 1 and not args.no_extras
            and args.config == "north_star"):
        # driver-timed lines for the other BASELINE configurations (one GPU; not `value`)
        del wl.joint, wl.src, wl.dst
        torch.cuda.empty_cache()
        # C5 runs at its stated shard (128 x 4K per GPU: 10 GB of images + a workspace of up to
        # 32 GiB) when the device has the room, else at batch 16
        free_b, _ = torch.cuda.mem_get_info()
        c5_batch = CONFIGS["c5"][1] if free_b >= (64 << 30) else 16
        if c5_batch != CONFIGS["c5"][1]:
            sys.stderr.write("bench.py: WARNING: only %.1f GiB of device memory free - c5_gf runs at batch "
                             "%d instead of its %d-image shard (\"batch_reduced\": true in the line)\n"
                             % (free_b / 2.0 ** 30, c5_batch, CONFIGS["c5"][1]))
        c5c_batch = CONFIGS["c5c"][1] if free_b >= (64 << 30) else 8
        for key, cfg, nb in (("c3_chain", "c3", 256), ("c5_gf", "c5", c5_batch),
                             ("c5_gf_colour", "c5c", c5c_batch)):
            k2, _, h2, w2 = CONFIGS[cfg]
            w2l = Workload(k2, nb, h2, w2, args, torch, rf, device, seed=1234 + 1000 * int(cfg[1]))
            w2l.step()
            el, kms = w2l.timed_steps(3, 1, sharding)
            gbs = w2l.pixels * w2l.bytes_per_px / (kms * 1e-3) / 1e9
            extras[key] = {"workload": w2l.name, "batch": nb,
                           "value": w2l.pixels * 3 / 1e6 / el, "unit": "MP/s",
                           "ms_per_step": el / 3 * 1e3, "steps": 3,
                           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_px": w2l.bytes_per_px}}
            px_step = w2l.pixels
            if cfg in ("c5", "c5c"):
                extras[key]["batch_reduced"] = nb != CONFIGS[cfg][1]
                # what the step is made of: its VALU-bound stage 1 alone and its memory-bound row /
                # column walks alone (timing switches of the library, results discarded): the step is
                # their SUM minus what the second stream recovers from launch tails - the two classes
                # cannot share a CU (profiles/r05_c5_overlap.md)
                alone = {}
                for part, skip in (("stage1", 6), ("walks", 1)):
                    with rf._ffi.debug_options(gf_exp_skip=skip):
                        w2l.step()
                        _, alone[part] = w2l.timed_steps(2, 1, sharding)
                w2l.step()                     # leave the buffers as a full step leaves them
                extras[key]["alone_ms"] = {"stage1_valu_issue_bound": alone["stage1"],
                                           "row_and_column_walks_memory_bound": alone["walks"],
                                           "sum": alone["stage1"] + alone["walks"], "step": kms}
            del w2l
            rf.ops.release_workspaces()
            torch.cuda.empty_cache()
            # memory-side traffic of one step of this configuration, per kernel (two --pmc child
            # passes; never fatal, skipped when profiled / told not to / out of its time).  The C3
            # chain's passes went to the colour chain in round 6 (6.3 B/px moved against 6 algorithmic
            # for three rounds running): its line carries the committed number, labelled.
            if cfg == "c3":
                c3t, c3src = committed_config_traffic("c3_chain", nb)
                extras[key]["roofline"].update({"traffic": c3t, "traffic_source": c3src})
                if c3t:
                    extras[key]["roofline"]["traffic_bytes_per_px"] = c3t / px_step
            elif args.traffic in ("auto", "live") and (args.traffic == "live" or (
                    not being_profiled() and os.environ.get("RF_BENCH_CHILD") != "1")):
                try:
                    rec, why = live_traffic_config(cfg, nb)
                except Exception as exc:              # noqa: BLE001 - a measurement aid
                    rec, why = None, repr(exc)
                if rec is None:
                    sys.stderr.write("bench.py: %s traffic measurement unavailable: %s\n" % (key, why))
                    extras[key]["roofline"].update({"traffic": None, "traffic_source": why})
                else:
                    for k in rec["kernels"].values():
                        k["bytes_per_px"] = (k["fetch_bytes"] + k["write_bytes"]) / px_step
                    extras[key]["roofline"].update(
                        {"traffic": rec["traffic"], "traffic_source": rec["traffic_source"],
                         "traffic_bytes_per_px": rec["traffic"] / px_step,
                         "traffic_kernels": rec["kernels"]})
                    if k2 in ("gf3", "gf3c"):   # a step is three passes over every pixel
                        extras[key]["roofline"]["traffic_bytes_per_px_per_pass"] = (
                            rec["traffic"] / px_step / 3.0)
                    # the rate the memory side actually runs at (measured bytes over the event-timed
                    # step), against the rate a plain copy reaches on this chip; for c5 the walks move
                    # their bytes in `alone_ms.row_and_column_walks_memory_bound`, i.e. faster than
                    # this step-average says - stage 1's share of the step moves few bytes
                    ms_side = rec["traffic"] / (kms * 1e-3) / 1e9
                    extras[key]["roofline"]["memory_side_gbs"] = ms_side
                    extras[key]["roofline"]["memory_side_frac_of_copy_ceiling"] = (
                        ms_side / HBM_COPY_CEILING_GBS)
                    if "alone_ms" in extras[key]:
                        walks = [v["fetch_bytes"] + v["write_bytes"] for n_, v in rec["kernels"].items()
                                 if "rowstate" in n_ or "colwalk" in n_]
                        if walks:
                            wg = sum(walks) / (extras[key]["alone_ms"]["row_and_column_walks_memory_bound"]
                                               * 1e-3) / 1e9
                            extras[key]["roofline"]["walks_alone_gbs"] = wg
                            extras[key]["roofline"]["walks_alone_frac_of_copy_ceiling"] = (
                                wg / HBM_COPY_CEILING_GBS)

    if world > 1:
        import torch.distributed as dist
        sharding.barrier(world)
        dist.destroy_process_group()
    if rank != 0:
        return
    value = px_total / 1e6 / t_max
    launch_px = wl.pixels
    out = {
        "metric": {"jbf": "megapixels/sec joint-bilateral sigma_c=20 sigma_s=22 @1080p",
                   "gf3": "megapixels/sec 3x guided filter c=3.0 s=45.0 @3840x2160 (BASELINE C5)",
                   "gf3c": "megapixels/sec 3x guided filter c=3.0 s=45.0 @3840x2160, colour src",
                   "chain": "megapixels/sec 1x1 CNN + BF(CNN,CNN) @IIW size (BASELINE C3)"}[kind],
        "value": value, "unit": "MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": t_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if kind not in ("gf3", "gf3c") else "f64", "data": "synthetic",
        "config": {"workload": wl.name, "name": args.config, "batch_per_gpu": n, "height": h,
                   "width": w, "sharding": "image batch, contiguous slices, no collective"},
        "per_rank_ms": per_rank_ms, "ranks_seen": len(per_rank_ms),
    }
    if not stub:
        achieved = launch_px * wl.bytes_per_px / (kernel_ms * 1e-3) / 1e9
        traffic, source = headline_traffic
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "traffic_source": source,
                           "kernel": getattr(live_traffic, "kernel", None),
                           "kernel_note": "the launch is timed through HIP events on its stream; "
                                          "'kernel' is the name rocprofv3 gave the launch in the "
                                          "live --pmc pass (null without one); profiles/ lists "
                                          "the kernels of the same command under rocprofv3",
                           "kernel_ms": kernel_ms,
                           "algorithmic_bytes_per_launch": launch_px * wl.bytes_per_px}
    if kind == "jbf" and not stub:
        radius = int(round(args.sigma_spatial * 1.5))
        taps = sum(1 for i in range(-radius, radius + 1) for j in range(-radius, radius + 1)
                   if (i * i + j * j) ** 0.5 <= radius)
        # the bound that actually limits an exact brute-force bilateral: VALU issue.  The grey
        # tap loop retires 26 VALU wave-instructions per 4-output column step; a gfx950 SIMD
        # issues at most one per 2 cycles (tools/microbench/valu_rates2.hip), 1024 SIMDs, 2.4 GHz.
        out["valu"] = valu_roofline(n, h, w, radius, kernel_ms, launch_px * taps, clock_mhz)
        # ... and the LDS gather pipeline, co-saturated with it (see lds_cobound)
        out["lds"] = lds_cobound(out["valu"]["column_steps_per_launch"], kernel_ms, clock_mhz)
        out["config"]["taps_per_px"] = taps
    out.update(extras)
    if image0 is not None:
        cpu_s = min(args.cpu_seconds, 0.3) if stub else args.cpu_seconds   # (the stub is a flow test)
        base = cpu_baseline_opencv(image0[0], image0[1], args.sigma_color, args.sigma_spatial, cpu_s)
        if base is None:
            base = cpu_baseline(image0[0], image0[1], args.sigma_color, args.sigma_spatial, cpu_s)
        out["cpu_baseline"] = base
    print(json.dumps(out))
    sys.stdout.flush()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
